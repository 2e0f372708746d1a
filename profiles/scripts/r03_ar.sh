#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ar; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_dist.py -x -q -m gpu -k "warp or disco or fusion or pipelined or bench_pipelined" 2>&1 | tail -4 | tee $O/pytest.log
for r in 1 2; do python bench.py --no-cpu-baseline > $O/bench_disco_$r.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3ar/bench_*.json")):
    d=json.loads([x for x in open(f) if x.startswith("{")][-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms_per_step"].get("pcp_warp_nearest_batch"), d["kernel_ms_per_step"].get("pcp_warp_nearest"))
PY
