#!/bin/bash
# round 5: PMC passes over the VFE-stage micro-benchmark (tools/bench_frontend.py): wave stall split, LDS conflicts, instruction mix
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ARGS=${1:-"4 6 1"}
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- python3 $R/practical-collab-perception_amd/tools/bench_frontend.py $ARGS > $O/p$i.log 2>&1 < /dev/null
  echo "pass $i rc=$?"
done
python3 - $O <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for key in ('k_pfn_rows', 'k_pfn<', 'k_point_cells', 'k_cell_scan', 'k_point_place<8', 'k_point_place<0', 'k_canvas_clear'):
            if key in k:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    wc = e.get('SQ_WAVE_CYCLES', 0) or 1
    print('==', k, ' launches', max(len(v) for v in d.values()))
    print('   ' + '  '.join('%s %.3g' % (c, v) for c, v in sorted(e.items())))
    print('   parked %.3f  issue-stalled %.3f (lds %.3f)  issuing %.3f | LDS conflict/active %.3f' % (
        e.get('SQ_WAIT_ANY', 0) / wc, e.get('SQ_WAIT_INST_ANY', 0) / wc, e.get('SQ_WAIT_INST_LDS', 0) / wc, e.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        e.get('SQ_LDS_BANK_CONFLICT', 0) / max(e.get('SQ_LDS_IDX_ACTIVE', 0), 1)))
    g = e.get('GRBM_GUI_ACTIVE')
    if g:
        print('   GRBM cycles per XCD %.0f; matrix pipe busy %.3f; LDS busy %.3f' % (g / 8, e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * 256 * g / 8), e.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * g / 8)))
    w = e.get('SQ_WAVES')
    if w:
        print('   per wave: VALU %.0f MFMA %.0f LDS %.0f SALU %.0f VMEM_RD %.0f VMEM_WR %.0f SMEM %.0f' % tuple(e.get(c, 0) / w for c in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM')))
PY
rm -rf $O/p1 $O/p2 $O/p3
