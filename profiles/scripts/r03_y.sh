#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3y; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -m gpu -k "winograd4 or fused_f4 or fused_winograd4" 2>&1 | tail -5 | tee $O/pytest_w4h.log
python bench.py --no-cpu-baseline --layer-table $O/layers_disco_4h.txt > $O/bench_disco_4h.json 2> $O/err1.txt
PCP_WINO4H=0 python bench.py --no-cpu-baseline > $O/bench_disco_4f.json 2> $O/err2.txt
python bench.py --no-cpu-baseline > $O/bench_disco_4h_b.json 2> $O/err3.txt
PCP_WINO4H=0 python bench.py --no-cpu-baseline > $O/bench_disco_4f_b.json 2> $O/err4.txt
python bench.py --no-cpu-baseline --config car > $O/bench_car_4h.json 2> $O/err5.txt
PCP_WINO4H=0 python bench.py --no-cpu-baseline --config car > $O/bench_car_4f.json 2> $O/err6.txt
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3y/bench_*.json')):
    l=[x for x in open(f) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel'][:10], d['roofline']['frac'], d['kernel_ms_per_step'])
P
head -30 $O/layers_disco_4h.txt
PCP_DIAG_SHAPES="128,128,64,64;256,256,64,64;256,128,128,128;128,256,128,128;64,64,128,128" PCP_DIAG_VARIANTS=h4_u4 PCP_DIAG_ENTRY=pcp_conv3x3_winograd4h timeout 900 python practical-collab-perception_amd/tools/bench_ws_diag.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/h4_more_shapes.txt
