#!/bin/bash
# round 3, run B: k_wino4f epilogue with half 1's dump + reads under half 0's store tail -- op tests, then interleaved A/B against the serial form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "winograd4f or auto_dispatch" 2>&1 | tail -5 > $O/pytest_f4.log; cat $O/pytest_f4.log
for B in 20 4; do
  PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/epilogue_ab.txt
done
