#!/bin/bash
# round 3, run D: fused weightor -- op + e2e tests, A/B of the headline with and without it, elided-makers line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "weight_fuse or fused_weightor or disco or decode_bbox" 2>&1 | tail -8 > $O/pytest_wf.log; cat $O/pytest_wf.log
PCP_DISCO_FUSED_WEIGHTOR=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco_unfused.json 2> $O/err1.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco_fused.json 2> $O/err2.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --elide-dead-makers > $O/bench_disco_elided.json 2> $O/err3.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3d/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print('%-40s %9.2f %s  %8.3f ms' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step']))
        print('   ', {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
tail -3 $O/err*.log
