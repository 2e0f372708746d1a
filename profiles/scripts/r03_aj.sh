#!/bin/bash
R=$GRAFT_REPO_ROOT
O=gpurun_out/r3aj; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
for mode in seq pipe; do
  extra=""; [ $mode = seq ] && extra="--no-pipeline"
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$mode -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline $extra > $R/$O/bench_$mode.log 2>&1
  T=$(find $R/$O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 $R/practical-collab-perception_amd/tools/gpu_idle.py $T 290 100 > $R/$O/gpu_idle_$mode.txt
  rm -rf $R/$O/trace_$mode
  head -3 $R/$O/gpu_idle_$mode.txt
done
