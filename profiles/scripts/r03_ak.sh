#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ak; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "pipelined_detector or overlapped_makers" 2>&1 | tail -5 | tee $O/pytest.log
for c in disco ego early; do
python bench.py --config $c --no-cpu-baseline > $O/bench_${c}_pipe.json 2> $O/err_$c.txt
python bench.py --config $c --no-cpu-baseline --no-pipeline > $O/bench_${c}_seq.json 2>> $O/err_$c.txt
done
python bench.py --no-cpu-baseline > $O/bench_disco_pipe_b.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pipeline > $O/bench_disco_seq_b.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3ak/bench_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["final_boxes_last_step"])
    else: print(f, "NO LINE")
PY
tail -3 $O/err_disco.txt
