#!/bin/bash
# round 3, run G: sparse first layer with channel-sliced waves (two taps per barrier) -- parity tests, interleaved A/B against the row-split form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "sparse or pipeline or full_size or first_layer" 2>&1 | tail -5 > $O/pytest_sparse.log; cat $O/pytest_sparse.log
for B in 4 20; do
  timeout 600 python practical-collab-perception_amd/tools/bench_pfn_ab.py sparse $B 2>&1 | grep -v amdgpu.ids | tee -a $O/sparse_ab.txt
done
