#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3as; mkdir -p $O
for r in 1 2; do
echo "== shipped =="; python practical-collab-perception_amd/tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids | grep "B=16"
echo "== nontemporal dword stores =="; PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_pw_nt.so python practical-collab-perception_amd/tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids | grep "B=16"
done | tee $O/pw_nt.txt
