#!/bin/bash
# round 2, final measurements part B: rocprofv3 kernel stats of the default command, PMC traffic passes (single stream)
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2final; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_disco -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_disco.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $R/$O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $R/$O/pmc_write.log 2>&1
cd $R
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -delete
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
cp profiles/r02_pmc_traffic.json $O/pmc_traffic.json
python3 practical-collab-perception_amd/tools/pmc_summary.py disco $F $W $O/pmc_traffic.json
find $O -name "*counter_collection.csv" -delete
S=$(find $O/prof_disco -name "*kernel_stats.csv" | head -1); cp $S $O/bench_disco_b4_kernel_stats.csv; head -12 $O/bench_disco_b4_kernel_stats.csv | cut -c1-160
python3 -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print(json.dumps(d.get('disco', {}), indent=0)[:1500])"
du -sh $O
