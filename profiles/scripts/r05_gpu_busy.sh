#!/bin/bash
# is a step kernel-bound or launch-bound?  union of the kernel intervals of a rocprofv3 kernel trace against the wall span
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/busy; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/p_$name -- python3 $R/bench.py "$@" --no-cpu-baseline --no-secondary > $O/$name.log 2>&1 < /dev/null
  f=$(ls $O/p_$name/*/*kernel_trace.csv 2>/dev/null | head -1)
  echo "== $name: $(grep '^{' $O/$name.log | tail -1 | cut -c1-160)"
  python3 $R/practical-collab-perception_amd/tools/gpu_busy.py "$f" 0.4
  rm -rf $O/p_$name; }
run train_bf16 --train --conv-algo bf16 --steps 30 --warmup 5
run train_f32 --train --steps 20 --warmup 5
run infer --steps 40 --warmup 5
