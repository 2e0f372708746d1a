#!/bin/bash
# round 5: HBM-side bytes of the front-end kernels (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5traffic; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ARGS=${1:-"4 6 1"}
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
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
        for key in ('k_pfn_rows', 'k_pfn<', 'k_point_cells', 'k_cell_scan', 'k_point_place<8', 'k_point_place<0', 'k_canvas_clear', 'pcp_k_zero'):
            if key in k:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    # FETCH_SIZE / WRITE_SIZE are in kilobytes; FETCH_SIZE counts 128-B requests as 64 B on gfx950: doubled
    print('%-18s fetch %.1f MB (x2 correction applied)  write %.1f MB' % (k, 2 * e.get('FETCH_SIZE', 0) * 1024 / 1e6, e.get('WRITE_SIZE', 0) * 1024 / 1e6))
PY
rm -rf $O/p1 $O/p2
