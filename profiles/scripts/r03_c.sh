#!/bin/bash
# round 3, run C: new tests (g13 exact final sets, dead-maker elision, bench --gpus 2 through the real kernels), then the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "well_conditioned or eliding" 2>&1 | tail -25 > $O/pytest_g13.log; cat $O/pytest_g13.log
timeout 1500 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -k "bench_two_ranks" 2>&1 | tail -25 > $O/pytest_dist.log; cat $O/pytest_dist.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
