#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ai; mkdir -p $O
timeout 600 python practical-collab-perception_amd/tools/trace_glue.py disco 2>&1 | grep -v amdgpu.ids | tee $O/glue_disco.txt | head -60
