#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bj; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_e2e.py -x -q -m gpu -k "prefetched or disco_train or train_py" 2>&1 | tail -12 | tee $O/pytest.log
for r in 1 2; do python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_prefetch_$r.json 2>$O/err_$r.txt; python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline > $O/bench_train_plain_$r.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3bj/bench_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("loss_last_step"))
    else: print(f,"NO LINE")
PY
tail -5 $O/err_1.txt
