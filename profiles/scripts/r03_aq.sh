#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3aq; mkdir -p $O
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee -a $O/pytest_twice.log; done
