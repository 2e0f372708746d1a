#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bb; mkdir -p $O
(echo "== default"; python profiles/scripts/debug/repro_seq.py 2>&1 | grep -v amdgpu | tail -3; echo "== early makers off"; PCP_PIPELINE_EARLY_MAKERS=0 python profiles/scripts/debug/repro_seq.py 2>&1 | grep -v amdgpu | tail -3; echo "== share_voxelization off in the last"; REPRO_SHARE=0 python profiles/scripts/debug/repro_seq.py 2>&1 | grep -v amdgpu | tail -3; echo "== overlap_makers off"; REPRO_OVERLAP=0 python profiles/scripts/debug/repro_seq.py 2>&1 | grep -v amdgpu | tail -3) | tee $O/repro.txt
