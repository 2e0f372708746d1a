#!/bin/bash
# round 3, run A: new compaction tests, whole GPU suite, headline bench with and without the compacted BEV-maker clouds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "compact or cells_ready or column_id or auto_dispatch" 2>&1 | tail -15 > $O/pytest_new.log; cat $O/pytest_new.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
PCP_BEVMAKER_COMPACT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco_masked.json 2> $O/bench_disco_masked.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco_compact.json 2> $O/bench_disco_compact.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3a/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print('%-40s %9.2f %s  %8.3f ms' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step']))
        print('   ', {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
