#!/bin/bash
# round 3, final measurements part A: the whole GPU suite + every bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_disco.json 2> $O/bench_disco.err
python bench.py --steps 20 --warmup 5 --no-overlap --no-cpu-baseline > $O/bench_disco_no_overlap.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline > $O/bench_disco_batch_by_batch.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --elide-dead-makers --no-cpu-baseline > $O/bench_disco_elided_dead_makers.json 2>/dev/null
python bench.py --config car --steps 20 --warmup 5 > $O/bench_car.json 2>/dev/null
python bench.py --config ego --steps 20 --warmup 5 > $O/bench_ego.json 2>/dev/null
python bench.py --config early --steps 20 --warmup 5 > $O/bench_early.json 2>/dev/null
python bench.py --config lately6 --steps 20 --warmup 5 > $O/bench_lately6.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --plugin-default --no-cpu-baseline > $O/bench_disco_plugin_default.json 2>/dev/null
python bench.py --dist ring --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco_ring.json 2>/dev/null
python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_disco_train.json 2>/dev/null
python bench.py --train --conv-algo bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_disco_train_bf16.json 2>/dev/null
python bench.py --config car --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_car_train.json 2>/dev/null
python bench.py --config ego --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ego_train.json 2>/dev/null
python bench.py --config early --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_early_train.json 2>/dev/null
PCP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_disco_2ranks_gloo_functional.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3final/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print('%-55s %9.2f %s  %8.3f ms  n_gpus %d' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['n_gpus']))
    except Exception as e:
        print(f, 'FAILED', e)
PY
