#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
for shp in "20 128 128 128 128" "4 128 128 128 128" "20 256 256 64 64"; do
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_h4_stamp.so python practical-collab-perception_amd/tools/stamp_h4.py $shp 2>&1 | grep -v amdgpu | tee -a $O/h4_stamps.txt
done
