set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd_ws" 2>&1 | tail -5
timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 > $O/ws_diag_b20.txt 2>&1; cat $O/ws_diag_b20.txt
timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py 4 > $O/ws_diag_b4.txt 2>&1; cat $O/ws_diag_b4.txt
