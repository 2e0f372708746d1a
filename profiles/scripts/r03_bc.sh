#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bc; mkdir -p $O
for i in 1 2 3; do echo "run $i: $(timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "pipelined_detector" 2>&1 | tail -1)"; done | tee $O/pytest.txt
for rep in 1 2; do timeout 900 python practical-collab-perception_amd/tools/stress_pipelined.py 300 $rep 2>&1 | grep -v amdgpu | tail -1; done | tee -a $O/pytest.txt
for rep in 1 2; do timeout 900 python profiles/scripts/debug/stress_fixture.py disco_full 300 $rep 2>&1 | grep -v amdgpu | tail -1; done | tee -a $O/pytest.txt
