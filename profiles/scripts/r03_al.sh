#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3al; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pointwise" 2>&1 | tail -3 | tee $O/pytest.log
for r in 1 2; do
echo "== 128x128 tiles (shipped) =="; python practical-collab-perception_amd/tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids
echo "== 128x64 tiles only =="; PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_pw_narrow.so python practical-collab-perception_amd/tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids
done | tee $O/pointwise_ab.txt
