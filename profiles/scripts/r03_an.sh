#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3an; mkdir -p $O
for r in 1 2; do
for m in 256 768; do
PCP_WINO4H_MIN_WGS=$m python bench.py --no-cpu-baseline > $O/bench_disco_min${m}_$r.json 2>/dev/null
PCP_WINO4H_MIN_WGS=$m python bench.py --config ego --no-cpu-baseline > $O/bench_ego_min${m}_$r.json 2>/dev/null
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3an/bench_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    d=json.loads(l[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
