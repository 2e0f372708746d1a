#!/bin/bash
# round 4 check B: headline bench line with the new keys, B = 1 latency lines (eager + hipGraph where supported), differenced launch census
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/r04_bench_disco_b.json 2> $O/r04_bench_disco_b.err
python - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
l=[x for x in open(R+'/gpurun_out/r04_bench_disco_b.json') if x.startswith('{')]
d=json.loads(l[0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm'], d['decode_nms_us'], d['pipelined_replicas'])
PY
tail -3 $O/r04_bench_disco_b.err
for c in car ego early disco lately6; do
  timeout 300 python bench.py --config $c --batch 1 --latency 200 < /dev/null 2>/dev/null | grep '^{' > $O/r04_latency_$c.json; python -c "
import json;d=json.loads(open('$O/r04_latency_$c.json').read());print('$c', d['p50_ms'], d['p99_ms'], d['mean_ms'])"
done
for c in car ego early; do
  timeout 300 python bench.py --config $c --batch 1 --latency 200 --graph < /dev/null 2>/dev/null | grep '^{' > $O/r04_latency_${c}_graph.json; python -c "
import json;d=json.loads(open('$O/r04_latency_${c}_graph.json').read());print('$c graph', d['p50_ms'], d['p99_ms'], d['mean_ms'])"
done
