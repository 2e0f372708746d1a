#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bl; mkdir -p $O
PCP_DIAG_VARIANTS=conv_nt PCP_DIAG_ENTRY=pcp_conv3x3 timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py 20 2>&1 | grep -v amdgpu.ids | tee $O/direct_nt.txt
for r in 1 2 3; do
python bench.py --no-cpu-baseline > $O/bench_ship_$r.json 2>/dev/null
PCP_HIP_LIB=$PWD/practical-collab-perception_amd/lib/variants/libpcp_hip_conv_nt.so python bench.py --no-cpu-baseline > $O/bench_nt_$r.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3bl/bench_*.json")):
    d=json.loads([x for x in open(f) if x.startswith("{")][-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
