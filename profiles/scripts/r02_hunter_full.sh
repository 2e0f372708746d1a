#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train_e2e.py -q -x -k "full_size" 2>&1 | tail -25
