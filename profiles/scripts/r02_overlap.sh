#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
python bench.py --train --steps 10 --warmup 3 2>/dev/null > gpurun_out/bench_disco_train_overlap_$i.json
python bench.py --train --steps 10 --warmup 3 --no-overlap 2>/dev/null > gpurun_out/bench_disco_train_nooverlap_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_disco_train_*overlap_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['config'].get('loss_last_step'))
PY
timeout 1200 python -m pytest tests/test_gpu_train_e2e.py -q -x -k "disco" 2>&1 | tail -5
