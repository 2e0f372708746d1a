#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_h4_eb4.so timeout 600 python practical-collab-perception_amd/tools/bench_w4h.py 3 2>&1 | grep -v amdgpu.ids | tee $O/w4h_eb4_check.txt
for B in 20 4; do
PCP_DIAG_VARIANTS=h4_ PCP_DIAG_ENTRY=pcp_conv3x3_winograd4h timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/h4_eb_ab.txt
done
