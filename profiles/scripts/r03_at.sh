#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3at; mkdir -p $O
for r in 1 2 3; do
python bench.py --no-cpu-baseline > $O/bench_ship_$r.json 2>/dev/null
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_pw_nt.so python bench.py --no-cpu-baseline > $O/bench_nt_$r.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3at/bench_*.json")):
    d=json.loads([x for x in open(f) if x.startswith("{")][-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
