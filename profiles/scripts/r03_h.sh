#!/bin/bash
# round 3, run H: sparse first layer (batched accumulator reads) + k_pfn mean sweep from registers: parity tests, A/B vs the previous forms
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "sparse or pipeline or full_size or first_layer or pfn or voxelize or degenerate" 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
for B in 4 20; do
  timeout 600 python practical-collab-perception_amd/tools/bench_pfn_ab.py sparse $B 2>&1 | grep -v amdgpu.ids | tee -a $O/sparse_ab.txt
done
timeout 600 python practical-collab-perception_amd/tools/bench_pfn_ab.py 4 6 2>&1 | grep -v amdgpu.ids | tee -a $O/pfn_ab.txt
timeout 600 python practical-collab-perception_amd/tools/bench_pfn_ab.py 20 1 2>&1 | grep -v amdgpu.ids | tee -a $O/pfn_ab.txt
