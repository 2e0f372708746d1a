#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train_e2e.py -q -x -k "train_py and (anchor or basic_car)" 2>&1 | tail -30 > gpurun_out/train_cli.log
python bench.py --config car --train --steps 5 --warmup 2 > gpurun_out/bench_car_train.json 2> gpurun_out/bench_car_train.err
tail -5 gpurun_out/bench_car_train.err >> gpurun_out/train_cli.log
cat gpurun_out/bench_car_train.json >> gpurun_out/train_cli.log
cat gpurun_out/train_cli.log
