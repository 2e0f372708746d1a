#!/bin/bash
# rocprofv3 kernel stats of the bf16 training step (config 5, 4 frames): bench.py --train --conv-algo bf16
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_tb16 -- python3 $R/bench.py --train --conv-algo bf16 --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/prof_tb16.log 2>&1 < /dev/null
f=$(ls $R/$O/prof_tb16/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $R/$O/r04_train_disco_b4_bf16_kernel_stats.csv; head -70 "$f" | cut -c1-200; else echo "no stats file"; tail -5 $R/$O/prof_tb16.log; fi
