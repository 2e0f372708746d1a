set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_disco.json 2> gpurun_out/r2a/bench_disco.err
tail -c 3000 gpurun_out/r2a/bench_disco.json
python bench.py --config car --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_car.json 2> gpurun_out/r2a/bench_car.err
python bench.py --config early --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_early.json 2>&1
python bench.py --config ego --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_ego.json 2>&1
PCP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_disco_2ranks_gloo.json 2> gpurun_out/r2a/bench_2r.err
python practical-collab-perception_amd/tools/bench_conv.py 4 > gpurun_out/r2a/bench_conv_b4.txt 2>&1
python practical-collab-perception_amd/tools/bench_conv.py 20 > gpurun_out/r2a/bench_conv_b20.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof_disco -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/prof_disco.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r2a -name "*.db" -delete
find gpurun_out/r2a -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/r2a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest_gpu.log
cat gpurun_out/r2a/pytest_gpu.log
