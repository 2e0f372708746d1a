#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ay; mkdir -p $O
for rep in 1 2; do for em in 1 0; do PCP_PIPELINE_EARLY_MAKERS=$em timeout 900 python practical-collab-perception_amd/tools/stress_pipelined.py 300 $rep 2>&1 | grep -v amdgpu | tail -1; done; done | tee $O/stress.txt
