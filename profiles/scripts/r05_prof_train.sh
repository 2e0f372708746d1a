#!/bin/bash
# rocprofv3 kernel stats of the bf16 training loop (config 5, 4 frames), single process
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/train_prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/bench.py --train --conv-algo bf16 --steps 12 --warmup 4 --no-cpu-baseline --no-secondary > $O/train_bf16.log 2>&1 < /dev/null
f=$(ls $O/p/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r05_train_disco_b4_bf16_kernel_stats.csv
rm -rf $O/p
grep '^{' $O/train_bf16.log | tail -1 | cut -c1-400
head -40 $O/r05_train_disco_b4_bf16_kernel_stats.csv | cut -c1-160
