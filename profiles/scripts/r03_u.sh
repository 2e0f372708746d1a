#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u; mkdir -p $O
timeout 600 python practical-collab-perception_amd/tools/bench_w4h.py 20 2>&1 | grep -v amdgpu.ids | tee $O/w4h_ab.txt
