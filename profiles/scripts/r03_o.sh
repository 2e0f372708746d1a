#!/bin/bash
# round 3, run O: k_wino4f first-round start stagger (half of the CUs start ~4 / ~8 us late) -- does de-synchronising the epilogue store bursts pay?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 2>&1 | grep -v amdgpu.ids | tee -a $O/f4_stagger_ab.txt
