#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bi; mkdir -p $O
for c in ego early car; do python bench.py --config $c --no-cpu-baseline --graph > $O/bench_${c}_graph.json 2>/dev/null; python bench.py --config $c --no-cpu-baseline > $O/bench_${c}_pipe.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3bi/bench_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    else: print(f,"NO LINE")
PY
