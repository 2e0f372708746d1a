#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3au; mkdir -p $O
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "pipelined_detector" 2>&1 | tail -2; done | tee $O/pytest.log
for r in 1 2 3; do
python bench.py --no-cpu-baseline > $O/bench_early_$r.json 2>/dev/null
PCP_PIPELINE_EARLY_MAKERS=0 python bench.py --no-cpu-baseline > $O/bench_wait_$r.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3au/bench_*.json")):
    d=json.loads([x for x in open(f) if x.startswith("{")][-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["final_boxes_last_step"])
PY
