#!/bin/bash
# round 4 check C: mp kernel tests, bf16 e2e tests, train tests (fp32 unchanged), bf16 training bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_mp.py -x -q < /dev/null 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_train_e2e.py -x -q < /dev/null 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_train_ops.py -x -q < /dev/null 2>&1 | tail -3
timeout 600 python bench.py --train --conv-algo bf16 --steps 10 --warmup 3 --no-cpu-baseline < /dev/null > $O/r04_train_bf16_d.json 2> $O/r04_train_bf16_d.err; python - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
l=[x for x in open(R+'/gpurun_out/r04_train_bf16_d.json') if x.startswith('{')]
d=json.loads(l[0]); print(d['ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['frac']); print(d['kernel_ms_per_step'])
PY
timeout 600 python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline < /dev/null 2>/dev/null | grep '^{' > $O/r04_train_fp32_d.json; python -c "
import json;d=json.loads(open('$O/r04_train_fp32_d.json').read());print('fp32 train', d['ms_per_step'])"
