cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd4f" 2>&1 | tail -3
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_f4_stamp.so python practical-collab-perception_amd/tools/stamp_f4.py 20 128 128 128 128 | tail -10
PCP_DIAG_ENTRY=pcp_conv3x3_winograd4f PCP_DIAG_VARIANTS=f4_nospec timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py 20
