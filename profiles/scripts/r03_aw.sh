#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3aw; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "pipelined_detector" 2>&1 | tail -12 | tee $O/pytest.log
for c in disco ego early; do for n in 1 2; do
python bench.py --config $c --no-cpu-baseline --pipeline-replicas $n > $O/bench_${c}_rep$n.json 2> $O/err_${c}_$n.txt
done; done
python bench.py --no-cpu-baseline --pipeline-replicas 2 > $O/bench_disco_rep2_b.json 2>/dev/null
python bench.py --no-cpu-baseline --pipeline-replicas 1 > $O/bench_disco_rep1_b.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3aw/bench_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["final_boxes_last_step"], d["config"]["peak_device_memory_mb"])
    else: print(f, "NO LINE")
PY
tail -5 $O/err_disco_2.txt
