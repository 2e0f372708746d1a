#!/bin/bash
# round 4 check A: new parity tests (late-fusion golden g14, full-size config-3 chain g13cf), bf16 loop tests, bf16 training bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "late_fusion or chain_full_size" < /dev/null > $O/r04_check_a_parity.log 2>&1; tail -5 $O/r04_check_a_parity.log
timeout 900 python -m pytest tests/test_gpu_train_e2e.py -x -q -k "bf16" -s < /dev/null > $O/r04_check_a_bf16.log 2>&1; grep -E "loop:|deviation|passed|failed" $O/r04_check_a_bf16.log | tail -8
timeout 600 python bench.py --train --conv-algo bf16 --steps 10 --warmup 3 --no-cpu-baseline < /dev/null > $O/r04_train_bf16_b.json 2> $O/r04_train_bf16_b.err; python - <<'PY'
import json,os
l=[x for x in open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out','r04_train_bf16_b.json')) if x.startswith('{')]
d=json.loads(l[0]); print(d['ms_per_step'], json.dumps(d['roofline'])[:900]); print(d['kernel_ms_per_step'])
PY
