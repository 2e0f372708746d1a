set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 600 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 > $O/ws_diag_b20.txt 2>&1; cat $O/ws_diag_b20.txt
timeout 300 python practical-collab-perception_amd/tools/bench_conv.py 20 > $O/bench_conv_b20.txt 2>&1; cat $O/bench_conv_b20.txt
