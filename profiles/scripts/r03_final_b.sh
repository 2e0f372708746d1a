#!/bin/bash
# round 3, final measurements part B: rocprofv3 kernel stats of the default command (overlapped + single stream), PMC traffic passes and
# SQ / LDS counter passes of the single-stream command
R=$GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_disco -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_disco.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_disco_ss -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap > $R/$O/prof_disco_ss.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $R/$O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $R/$O/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $R/$O/pmc_sq.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_lds -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $R/$O/pmc_lds.log 2>&1
cd $R
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -delete
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 practical-collab-perception_amd/tools/pmc_summary.py disco $F $W $O/r03_pmc_traffic.json | head -40
Q=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); L=$(find $O/pmc_lds -name "*counter_collection.csv" | head -1)
python3 practical-collab-perception_amd/tools/pmc_sq_summary.py $O/r03_pmc_sq_counters.json $Q $L
find $O -name "*counter_collection.csv" -delete
S=$(find $O/prof_disco -name "*kernel_stats.csv" | head -1); cp $S $O/bench_disco_b4_overlapped_kernel_stats.csv
S=$(find $O/prof_disco_ss -name "*kernel_stats.csv" | head -1); cp $S $O/bench_disco_b4_kernel_stats.csv; head -14 $O/bench_disco_b4_kernel_stats.csv | cut -c1-160
tail -2 $O/pmc_sq.log $O/pmc_lds.log
du -sh $O
