#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -x -k "overlap or (disco_full_size)" 2>&1 | tail -15 > gpurun_out/overlap_tests.log
cat gpurun_out/overlap_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 2>/dev/null > gpurun_out/bench_disco_overlap_$i.json
python bench.py --steps 20 --warmup 5 --no-overlap 2>/dev/null > gpurun_out/bench_disco_nooverlap_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_disco_*overlap_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'])
PY
