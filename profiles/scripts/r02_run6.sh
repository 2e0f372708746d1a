cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd4f" 2>&1 | tail -5
PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f PCP_DIAG_VARIANTS=f4_ timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 > $O/f4_diag_b20_v3.txt 2>&1; cat $O/f4_diag_b20_v3.txt
