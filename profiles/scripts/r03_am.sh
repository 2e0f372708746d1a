#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3am; mkdir -p $O
echo "== shipped =="; python practical-collab-perception_amd/tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids | grep "B=16"
echo "== no stores =="; PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_pw_nostore.so python practical-collab-perception_amd/tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids | grep "B=16"
