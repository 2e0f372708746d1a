#!/bin/bash
# round 4 final measurements, part B: rocprofv3 kernel stats of the headline command and of the bf16 training step, launch census
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_disco -- python3 $R/bench.py --steps 15 --warmup 3 --no-cpu-baseline > $O/prof_disco.log 2>&1 < /dev/null
f=$(ls $O/prof_disco/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r04_bench_disco_b4_kernel_stats.csv && head -12 "$f" | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tb16 -- python3 $R/bench.py --train --conv-algo bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_tb16.log 2>&1 < /dev/null
f=$(ls $O/prof_tb16/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r04_train_disco_b4_bf16_kernel_stats.csv && head -8 "$f" | cut -c1-160
bash $R/profiles/scripts/r04_launch_census.sh "" default > $O/r04_launch_census_default.txt 2>&1; cat $O/r04_launch_census_default.txt
bash $R/profiles/scripts/r04_launch_census.sh "--no-pipeline" nopipe > $O/r04_launch_census_batch_by_batch.txt 2>&1; head -3 $O/r04_launch_census_batch_by_batch.txt
