#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bh; mkdir -p $O
for r in 1 2; do
python bench.py --no-cpu-baseline > $O/bench_auto_$r.json 2>/dev/null
PCP_WINO4H_MIN_WGS=128 python bench.py --no-cpu-baseline > $O/bench_min128_$r.json 2>/dev/null
PCP_WINO4H_MIN_WGS=64 python bench.py --no-cpu-baseline > $O/bench_min064_$r.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3bh/bench_*.json")):
    d=json.loads([x for x in open(f) if x.startswith("{")][-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
