#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3aa; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TA_[A-Z_0-9]+)\b" | sort -u > $R/$O/counters.txt
for k in 4h 4f; do
 for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $grp | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/${k}_$tag -- python3 $R/practical-collab-perception_amd/tools/run_conv_once.py $k 20 128 128 128 128 6 > $R/$O/${k}_$tag.log 2>&1
 done
done
cd $R
python3 - <<'P'
import csv,glob,collections
for k in ('4h','4f'):
    acc=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/r3aa/%s_*/**/*counter_collection.csv'%k, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_wino4' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    e={c:sum(v)/len(v) for c,v in acc.items()}
    print(k, {c:round(v) for c,v in sorted(e.items())})
    g=e.get('GRBM_GUI_ACTIVE'); wc=e.get('SQ_WAVE_CYCLES')
    if g and wc:
        print('   mfma busy %.3f  lds busy %.3f  bank-conflict/lds %.3f  parked %.3f  stalled %.3f  issuing %.3f' % (e['SQ_VALU_MFMA_BUSY_CYCLES']/(4*256*g/8), e['SQ_LDS_IDX_ACTIVE']/(256*g/8), e['SQ_LDS_BANK_CONFLICT']/e['SQ_LDS_IDX_ACTIVE'], e['SQ_WAIT_ANY']/wc, e['SQ_WAIT_INST_ANY']/wc, e['SQ_ACTIVE_INST_ANY']/wc))
P
grep -E "SQ_WAIT|SQ_INST_LEVEL|SQ_LDS|SQ_ACTIVE|MFMA" $O/counters.txt | tr '\n' ' '
