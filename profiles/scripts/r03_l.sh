#!/bin/bash
# round 3, run L: k_wino4f main loop as nine fenced blocks (variant f4_blocks) -- parity of the variant, interleaved A/B vs the shipped step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_f4_blocks.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "winograd4f" 2>&1 | tail -4 > $O/pytest_variant.log; cat $O/pytest_variant.log
for B in 20 4; do
  PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/f4_blocks_ab.txt
done
