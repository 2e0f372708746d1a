#!/bin/bash
# round 3, run N: k_wino4f block-schedule variants (shipped = nine fenced blocks; compiler = hipcc's own order; ring4; other work first; local interleave)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "winograd4f" 2>&1 | tail -2 | tee -a $O/pytest.log
for V in f4_ring4 f4_otherfirst f4_interleave; do
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_$V.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "winograd4f_matches" 2>&1 | tail -2 | tee -a $O/pytest.log
done
for B in 20 4; do
  PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/f4_ab.txt
done
