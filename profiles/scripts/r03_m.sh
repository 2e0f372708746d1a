#!/bin/bash
# round 3, run M: k_wino4f fenced blocks with / without the SIMD partners de-phased inside every block
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
for V in f4_blocks f4_dephase; do
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_$V.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "winograd4f" 2>&1 | tail -2 | tee -a $O/pytest_variant.log
done
for B in 20 4; do
  PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py $B 2>&1 | grep -v amdgpu.ids | tee -a $O/f4_ab.txt
done
