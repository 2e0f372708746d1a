#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_f4_stampnoswap.so python practical-collab-perception_amd/tools/stamp_f4.py 20 128 128 128 128 2>&1 | grep -v amdgpu | tail -9 | tee $O/stamps_noswap.txt
PCP_DIAG_VARIANTS=f4_noswap PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 2>&1 | grep -v amdgpu.ids | tee -a $O/f4_noswap_ab.txt
