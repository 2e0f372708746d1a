#!/bin/bash
# per-step launch census of the headline command: two rocprofv3 kernel-trace runs (10 and 30 timed steps), counts differenced so that
# one-time work (weight packing, BatchNorm folding, allocations' fills) drops out; $1 = extra bench.py flags (e.g. "--no-pipeline")
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${2:-default}
cd /tmp; export TMPDIR=/tmp
for n in 10 30; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/census_${TAG}_$n -- python3 $R/bench.py --steps $n --warmup 3 --no-cpu-baseline $1 > $O/census_${TAG}_$n.log 2>&1 < /dev/null
done
python3 - $O $TAG <<'PY'
import csv, glob, sys, re
O, TAG = sys.argv[1], sys.argv[2]
def load(n):
    f = glob.glob('%s/census_%s_%d/*/*kernel_stats.csv' % (O, TAG, n))[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a, b = load(10), load(30)
own = other = 0.0
lines = []
for k in sorted(set(a) | set(b)):
    dc = (b.get(k, (0, 0))[0] - a.get(k, (0, 0))[0]) / 20.0
    dt = (b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]) / 20.0 / 1e3
    if abs(dc) < 1e-9:
        continue
    mine = ('anonymous namespace' in k or k.startswith('pcp_') or '_GLOBAL__N_' in k)
    if mine: own += dc
    else: other += dc
    lines.append((mine, dc, dt, re.sub(r'\s+', ' ', k)[:130]))
print('launches per step: %.1f own (libpcp_hip.so) + %.1f other (ATen / runtime copies)' % (own, other))
for mine, dc, dt, k in sorted(lines, key=lambda t: (t[0], -t[1])):
    if not mine:
        print('  other %6.2f /step %8.1f us/step  %s' % (dc, dt, k))
PY
