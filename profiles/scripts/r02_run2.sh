set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd_ws" 2>&1 | tail -15 > $O/pytest_ws.log; cat $O/pytest_ws.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "switches" 2>&1 | tail -15 > $O/pytest_switch.log; cat $O/pytest_switch.log
timeout 300 python practical-collab-perception_amd/tools/bench_conv.py 4 > $O/bench_conv_b4.txt 2>&1; cat $O/bench_conv_b4.txt
timeout 300 python practical-collab-perception_amd/tools/bench_conv.py 20 > $O/bench_conv_b20.txt 2>&1; cat $O/bench_conv_b20.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_disco -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/prof_disco.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_write.log 2>&1
cd $R
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -size +30M -delete
find $O -type f | head -30; du -sh $O
