#!/bin/bash
# round 3, run P: direct conv with the fragment reads one step ahead (pinned) vs hipcc's order; parity of every conv op test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv" 2>&1 | tail -3 | tee $O/pytest.log
for B in 20 4; do
  PCP_DIAG_ENTRY=pcp_conv3x3 timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_s2_ab.txt
done
