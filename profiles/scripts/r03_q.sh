#!/bin/bash
# round 3, run Q: in-kernel stamps of the blocked k_wino4f step (diagnostic build, workgroup 100)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_f4_stamp.so python practical-collab-perception_amd/tools/stamp_f4.py 20 128 128 128 128 2>&1 | grep -v amdgpu | tee $O/stamps_128.txt
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_f4_stamp.so python practical-collab-perception_amd/tools/stamp_f4.py 20 256 256 64 64 2>&1 | grep -v amdgpu | tee $O/stamps_64.txt
