#!/bin/bash
# round 5: rocprofv3 kernel stats of the VFE-stage micro-benchmark (tools/bench_frontend.py), old vs new front end
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for a in "4 6 1" "20 1 0" "4 1 0"; do
  tag=$(echo $a | tr " " _)
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/practical-collab-perception_amd/tools/bench_frontend.py $a > $O/prof_$tag.log 2>&1 < /dev/null
  echo "== $a rc=$?"
  f=$(ls $O/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $O/frontend_${tag}_kernel_stats.csv && head -14 "$f" | cut -d, -f1-4 | cut -c1-140
  rm -rf $O/prof_$tag
done
