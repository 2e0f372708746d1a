#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3av; mkdir -p $O
for i in 1 2 3; do timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "pipelined_detector" 2>&1 | tail -3; done | tee $O/pytest.log
