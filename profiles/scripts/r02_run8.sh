cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco.json 2> $O/bench_disco.err; cat $O/bench_disco.json
python bench.py --config car --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_car.json 2>&1; tail -c 1500 $O/bench_car.json
python bench.py --config early --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_early.json 2>&1; tail -c 900 $O/bench_early.json
python bench.py --config ego --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ego.json 2>&1; tail -c 900 $O/bench_ego.json
