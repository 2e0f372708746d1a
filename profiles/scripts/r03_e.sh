#!/bin/bash
# round 3, run E: whole GPU suite; training evidence (fp32 + bf16 lines, rocprof kernel stats of the DiscoNet training step)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3e; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_disco_train.json 2> $O/err_train.log
python bench.py --train --conv-algo bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_disco_train_bf16.json 2> $O/err_train_bf16.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train -- python3 $R/bench.py --train --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train_bf16 -- python3 $R/bench.py --train --conv-algo bf16 --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_train_bf16.log 2>&1
cd $R
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete
S=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); cp $S $O/train_disco_b4_kernel_stats.csv; head -14 $O/train_disco_b4_kernel_stats.csv | cut -c1-150
S=$(find $O/prof_train_bf16 -name "*kernel_stats.csv" | head -1); cp $S $O/train_disco_b4_bf16_kernel_stats.csv; head -8 $O/train_disco_b4_bf16_kernel_stats.csv | cut -c1-150
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3e/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print('%-40s %9.2f %s  %8.3f ms  dtype %s' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['dtype'][:40]))
        print('   ', {k: round(v, 3) for k, v in list(d['kernel_ms_per_step'].items())[:8]})
    except Exception as e:
        print(f, 'FAILED', e)
PY
du -sh $O
