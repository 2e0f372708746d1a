#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ag; mkdir -p $O
for B in 20 4; do PCP_DIAG_SHAPES="128,128,128,128;64,64,256,256;128,384,128,128" PCP_DIAG_VARIANTS=h4_ PCP_DIAG_ENTRY=pcp_conv3x3_winograd4h timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/h4_raw.txt; done
