#!/bin/bash
# rocprofv3 kernel trace of the single-stream headline run with each pointwise kernel: per-launch durations of k_pointwise / k_pw_stream in situ
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pw_prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for a in tile stream; do
  export PCP_PW_ALGO=$a
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/p_$a -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap --no-pipeline > $O/$a.log 2>&1 < /dev/null
  f=$(ls $O/p_$a/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 - "$f" $a <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'k_pointwise' in n or 'k_pw_stream' in n:
        import re; key = re.search(r'(k_pointwise<\d|k_pw_stream<\d, \d+, \d)', n).group(1)
        d[(key, r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', '?'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
for k in sorted(d):
    v = d[k]
    print(tag, k, 'n=%d' % len(v), 'avg %.1f us  min %.1f  max %.1f' % (sum(v) / len(v), min(v), max(v)))
PY
  rm -rf $O/p_$a
done
