#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bk; mkdir -p $O
for rep in 2 1; do timeout 1200 python practical-collab-perception_amd/tools/stress_pipelined.py 2000 $rep 2>&1 | grep -v amdgpu | tail -1; done | tee $O/stress_long.txt
for rep in 2 1; do timeout 900 python profiles/scripts/debug/stress_fixture.py disco_full 1500 $rep 2>&1 | grep -v amdgpu | tail -1; done | tee -a $O/stress_long.txt
timeout 600 python profiles/scripts/debug/repro_seq.py 2>&1 | grep -v amdgpu | tail -3 | tee -a $O/stress_long.txt
