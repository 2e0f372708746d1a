#!/bin/bash
# PMC pass over the bf16 kernels in isolation (tools/bench_mp.py, 4 frames): LDS bank conflicts and wave stall split of k_mp_wgrad3x3 / k_mp_conv3x3_s1
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_mp -- python3 $R/practical-collab-perception_amd/tools/bench_mp.py 4 > $O/pmc_mp.log 2>&1 < /dev/null
f=$(ls $O/pmc_mp/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no counter file"; tail -5 $O/pmc_mp.log; exit 0; fi
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    if 'mp_' not in k:
        continue
    k = k.split('(')[0][-60:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    n = max(cnt[(k, 'SQ_WAVE_CYCLES')], 1)
    wc = d['SQ_WAVE_CYCLES'] or 1
    print('%-62s launches %3d  LDS conflict/active %.3f  wait_any %.3f  wait_inst %.3f (lds %.3f)  active %.3f  mfma_busy/wave_cyc %.3f' % (
        k, n, d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1), d['SQ_WAIT_ANY'] / wc, d['SQ_WAIT_INST_ANY'] / wc, d['SQ_WAIT_INST_LDS'] / wc,
        d['SQ_ACTIVE_INST_ANY'] / wc, d['SQ_VALU_MFMA_BUSY_CYCLES'] / wc))
PY
