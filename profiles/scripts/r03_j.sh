#!/bin/bash
# round 3, run J: pillariser rank-from-histogram + weightor load batching: whole suite, headline bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco.json 2> $O/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap > $O/bench_disco_no_overlap.json 2>> $O/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3j/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print('%-40s %9.2f %s  %8.3f ms frac %s' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac']))
        print('   ', {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
