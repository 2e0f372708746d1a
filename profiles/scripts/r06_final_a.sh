#!/bin/bash
# round 6 final measurements, part A: bench lines of every config (inference, training fp32 / bf16), B = 1 latency lines (eager and hipGraph)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6final; mkdir -p $O; cd $R
run() { name=$1; shift; sec="--no-secondary --no-configs"; [ "$name" = r06_bench_disco ] && sec=; timeout 500 python bench.py "$@" $sec < /dev/null 2>/dev/null | grep '^{' > $O/$name.json; python - $O/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], d.get('value'), d.get('unit'), d.get('ms_per_step', d.get('p50_ms')))
PY
}
run r06_bench_disco --steps 20 --warmup 5
run r06_bench_disco_pipeline_graph --steps 20 --warmup 5 --no-cpu-baseline --no-configs --pipeline-graph
run r06_bench_disco_batch_by_batch --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline
run r06_bench_disco_optin --steps 20 --warmup 5 --no-cpu-baseline --optin
run r06_bench_disco_ring --steps 20 --warmup 5 --dist ring --no-cpu-baseline
run r06_bench_car --config car --steps 20 --warmup 5 --no-cpu-baseline
run r06_bench_ego --config ego --steps 20 --warmup 5 --no-cpu-baseline
run r06_bench_early --config early --steps 20 --warmup 5 --no-cpu-baseline
run r06_bench_lately6 --config lately6 --steps 20 --warmup 5 --no-cpu-baseline
run r06_bench_disco_train --train --steps 10 --warmup 3 --no-cpu-baseline
run r06_bench_disco_train_bf16 --train --conv-algo bf16 --steps 10 --warmup 3 --no-cpu-baseline
run r06_bench_disco_elided_dead_makers --steps 20 --warmup 5 --elide-dead-makers --no-cpu-baseline
for c in car ego early disco lately6; do run r06_latency_$c --config $c --batch 1 --latency 200; done
for c in car ego early disco; do run r06_latency_${c}_graph --config $c --batch 1 --latency 200 --graph; done
