#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ae; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py -x -q -m gpu -k "wgrad or conv_bn_act" 2>&1 | tail -4 | tee $O/pytest.log
python practical-collab-perception_amd/tools/bench_wgrad.py 4 2>&1 | grep -v amdgpu.ids | grep wgrad | tee $O/wgrad_b4.txt
