#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ao; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3 and not winograd" 2>&1 | tail -3 | tee $O/pytest.log
PCP_DIAG_VARIANTS=conv_old PCP_DIAG_ENTRY=pcp_conv3x3 timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 2>&1 | grep -v amdgpu.ids | tee $O/direct_s2_ab.txt
PCP_DIAG_VARIANTS=conv_old PCP_DIAG_ENTRY=pcp_conv3x3 timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/direct_s2_ab.txt
