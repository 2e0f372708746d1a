set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd4f" 2>&1 | tail -25 > $O/pytest_4f.log; cat $O/pytest_4f.log
timeout 300 python practical-collab-perception_amd/tools/bench_conv.py 4 > $O/bench_conv_b4.txt 2>&1; cat $O/bench_conv_b4.txt
timeout 300 python practical-collab-perception_amd/tools/bench_conv.py 20 > $O/bench_conv_b20.txt 2>&1; cat $O/bench_conv_b20.txt
