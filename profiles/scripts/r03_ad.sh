#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ad; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_e2e.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_train.log
python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --layer-table $O/layers_train.txt > $O/bench_disco_train.json 2>$O/err.txt
PCP_WINO4H=0 python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_disco_train_b.json 2>$O/err2.txt
python bench.py --config ego --train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ego_train.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3ad/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
head -30 $O/layers_train.txt
