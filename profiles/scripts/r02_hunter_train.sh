#!/bin/bash
# HunterJr training tests (round 2)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_e2e.py -q -x -k "hunter or segment_max or hard_mining or bev_correction or filter_gt" 2>&1 | tail -60 > gpurun_out/hunter_train_tests.log
cat gpurun_out/hunter_train_tests.log
