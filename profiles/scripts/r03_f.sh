#!/bin/bash
# round 3, run F: k_pfn with prefetched pillar ids, run-folded running-max updates and LDS-staged 16-byte row stores -- parity tests, then
# interleaved A/B against the round-2 epilogue (variant pfn_old)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "pfn or voxelize or single_agent or full_size or degenerate" 2>&1 | tail -5 > $O/pytest_pfn.log; cat $O/pytest_pfn.log
for A in 6 1; do
  timeout 600 python practical-collab-perception_amd/tools/bench_pfn_ab.py 4 $A 2>&1 | grep -v amdgpu.ids | tee -a $O/pfn_ab.txt
done
timeout 600 python practical-collab-perception_amd/tools/bench_pfn_ab.py 20 1 2>&1 | grep -v amdgpu.ids | tee -a $O/pfn_ab.txt
