#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
python practical-collab-perception_amd/tools/bench_wgrad.py 4 2>&1 | grep -v amdgpu.ids | grep wgrad | tee $O/wgrad_b4.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o wg -- python3 $GRAFT_REPO_ROOT/practical-collab-perception_amd/tools/bench_wgrad.py 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r3s/prof/**/*kernel_trace.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
import collections
seq=[(r['Kernel_Name'][:60],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows if 'wgrad' in r['Kernel_Name']]
# group consecutive identical (13 launches per layer)
i=0
while i<len(seq):
    j=i
    while j<len(seq) and j-i<26: j+=1
    mains=[t for n,t in seq[i:j] if 'reduce' not in n]; reds=[t for n,t in seq[i:j] if 'reduce' in n]
    print('main %8.1f us   reduce %6.1f us'%(sorted(mains)[len(mains)//2], sorted(reds)[len(reds)//2]))
    i=j
P
