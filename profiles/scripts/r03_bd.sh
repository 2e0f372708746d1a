#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bd; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "bench" 2>&1 | tail -40 | tee $O/pytest.log
