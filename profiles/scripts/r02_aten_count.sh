#!/bin/bash
mkdir -p gpurun_out/aten
cd /tmp && export TMPDIR=/tmp
for n in 10 30; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/aten/s$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps $n --warmup 2 --no-overlap > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
def load(n):
    f = glob.glob('gpurun_out/aten/s%d/**/*kernel_stats.csv' % n, recursive=True)[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a, b = load(10), load(30)
rows = []
for k in b:
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b[k]
    rows.append(((tb - ta) / 20.0 / 1e3, (cb - ca) / 20.0, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('per step: %.1f us total' % tot)
for us, calls, k in rows[:60]:
    print('%9.1f us  %6.1f calls  %s' % (us, calls, k[:110]))
PY
rm -rf gpurun_out/aten
