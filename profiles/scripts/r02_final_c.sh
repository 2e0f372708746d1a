#!/bin/bash
# round 2, final measurements part C: rocprofv3 kernel stats of the single-stream command (the mode bench.py's instrumented pass uses)
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2final; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_disco_ss -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap > $R/$O/prof_disco_ss.log 2>&1
cd $R
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -delete
S=$(find $O/prof_disco_ss -name "*kernel_stats.csv" | head -1); cp $S $O/bench_disco_b4_single_stream_kernel_stats.csv; head -6 $O/bench_disco_b4_single_stream_kernel_stats.csv | cut -c1-160
