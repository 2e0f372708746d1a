#!/bin/bash
# A/B of the pointwise kernels inside the headline bench (PCP_PW_ALGO=tile | stream), two runs each, interleaved
mkdir -p gpurun_out/ab_pw
for i in 1 2; do
  for a in tile stream; do
    PCP_PW_ALGO=$a python bench.py --steps 30 --warmup 8 > gpurun_out/ab_pw/${a}_$i.json 2> gpurun_out/ab_pw/${a}_$i.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/ab_pw/${a}_$i.json').read().strip().splitlines()[-1])
print('$a', $i, d['value'], d['ms_per_step'], {k:v for k,v in d.get('kernels',{}).items() if 'pointwise' in k or 'pw' in k} if isinstance(d.get('kernels'),dict) else '')
PY
  done
done
