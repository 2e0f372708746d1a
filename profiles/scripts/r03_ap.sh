#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ap; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dist.py -x -q -m gpu -k "tools_test_py or test_py or pipelined" 2>&1 | tail -5 | tee $O/pytest.log
