#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ab; mkdir -p $O
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_h4_stamp.so python practical-collab-perception_amd/tools/stamp_h4.py 20 128 128 128 128 2>&1 | grep -v amdgpu | tee -a $O/h4_slice_stamps.txt
