#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3be; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "pipelined_detector" 2>&1 | tail -4 | tee $O/pytest.log
for n in 1 2 3; do python bench.py --config car --no-cpu-baseline --pipeline-replicas $n > $O/bench_car_rep$n.json 2>$O/err_car_$n.txt; done
python bench.py --config car --no-cpu-baseline --no-pipeline > $O/bench_car_seq.json 2>/dev/null
python bench.py --no-cpu-baseline --pipeline-replicas 3 > $O/bench_disco_rep3.json 2>/dev/null
python bench.py --no-cpu-baseline --pipeline-replicas 2 > $O/bench_disco_rep2.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3be/bench_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["final_boxes_last_step"])
    else: print(f, "NO LINE")
PY
tail -3 $O/err_car_2.txt
