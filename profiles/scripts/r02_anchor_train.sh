#!/bin/bash
# anchor-head training tests (round 2)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_e2e.py -q -x -k "anchor" 2>&1 | tail -40 > gpurun_out/anchor_train_tests.log
cat gpurun_out/anchor_train_tests.log
