#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ax; mkdir -p $O
for em in 1 0; do for i in 1 2 3 4 5 6; do echo "early_makers=$em run $i: $(PCP_PIPELINE_EARLY_MAKERS=$em timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "pipelined_detector and not 2]" 2>&1 | tail -1)"; done; done | tee $O/flaky2.txt
