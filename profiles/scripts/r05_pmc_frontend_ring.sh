#!/bin/bash
# SQ / LDS counters of the front end on the LiDAR-like cloud (tools/bench_frontend.py 4 6 1 ring) next to the uniform one
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/pmc_ring; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
for dist in ring uniform; do
  pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/${dist}_$name -- python3 $R/practical-collab-perception_amd/tools/bench_frontend.py 4 6 1 $dist > $R/$O/${dist}_$name.log 2>&1 < /dev/null; }
  pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
  pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
  Q=$(find $R/$O/${dist}_sq -name "*counter_collection.csv" | head -1); L=$(find $R/$O/${dist}_lds -name "*counter_collection.csv" | head -1)
  python3 $R/practical-collab-perception_amd/tools/pmc_sq_summary.py $R/$O/r05_pmc_frontend_$dist.json $Q $L > /dev/null
  python3 - $R/$O/r05_pmc_frontend_$dist.json $dist <<'PY'
import json, sys
q = json.load(open(sys.argv[1]))
for k in ('k_pfn_rows', 'k_pfn_crowd', 'k_point_cells', 'k_point_place'):
    if k in q:
        print(sys.argv[2], k, {a.split(' ')[0]: b for a, b in q[k]['derived'].items()}, 'launches', q[k]['launches'])
PY
  find $R/$O -name "*.db" -delete; find $R/$O -name "*.csv" -delete
done
