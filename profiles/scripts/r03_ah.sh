#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ah; mkdir -p $O
PCP_DIAG_SHAPES="128,128,128,128;64,64,256,256" PCP_DIAG_VARIANTS=h4d_ PCP_DIAG_ENTRY=pcp_conv3x3_winograd4h timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 2>&1 | grep -v amdgpu.ids | tee -a $O/h4_raw2.txt
