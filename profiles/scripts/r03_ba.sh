#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ba; mkdir -p $O
for i in $(seq 1 10); do echo "run $i: $(timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "stress_many" 2>&1 | tail -1)"; done | tee $O/flaky4.txt
