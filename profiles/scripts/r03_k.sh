#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "lately" 2>&1 | tail -30 > $O/pytest_chain.log; cat $O/pytest_chain.log
