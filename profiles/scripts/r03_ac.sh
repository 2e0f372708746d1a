#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ac; mkdir -p $O
for B in 4 20 1; do timeout 900 python practical-collab-perception_amd/tools/bench_conv.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_conv.txt; done
