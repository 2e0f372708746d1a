cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd4f" 2>&1 | tail -3
PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f PCP_DIAG_VARIANTS=zzz timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py 20
PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f PCP_DIAG_VARIANTS=zzz timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py 4
