#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bf; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "lately" 2>&1 | tail -6 | tee $O/pytest.log
for n in 1 2; do python bench.py --config lately6 --no-cpu-baseline --pipeline-replicas $n > $O/bench_lately_rep$n.json 2>$O/err_$n.txt; done
python bench.py --config lately6 --no-cpu-baseline --no-pipeline > $O/bench_lately_seq.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3bf/bench_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["final_boxes_last_step"])
    else: print(f, "NO LINE")
PY
tail -4 $O/err_2.txt
