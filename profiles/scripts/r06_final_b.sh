#!/bin/bash
# round 6 final measurements, part B: rocprofv3 kernel stats of the headline command -- the overlapped default run AND the single-stream run
# the roofline's avg_launch_us corresponds to -- and of the VFE-stage micro-benchmark
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
prof() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py "$@" > $O/prof_$name.log 2>&1 < /dev/null
  f=$(ls $O/prof_$name/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r06_bench_disco_b4_${name}_kernel_stats.csv && head -6 "$f" | cut -c1-150; rm -rf $O/prof_$name; }
prof overlapped --steps 15 --warmup 3 --no-cpu-baseline --no-secondary --no-configs
prof single_stream --steps 15 --warmup 3 --no-cpu-baseline --no-secondary --no-configs --no-overlap --no-pipeline
grep '^{' $O/prof_single_stream.log | tail -1 > $O/r06_bench_disco_single_stream.json
for a in "4 6 1" "20 1 0" "4 1 0"; do
  tag=$(echo $a | tr " " _)
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf_$tag -- python3 $R/practical-collab-perception_amd/tools/bench_frontend.py $a > $O/r06_frontend_$tag.txt 2>&1 < /dev/null
  f=$(ls $O/pf_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r06_frontend_${tag}_kernel_stats.csv; rm -rf $O/pf_$tag
done
cd $R; for a in "4 6 1" "20 1 0" "4 1 0"; do timeout 120 python practical-collab-perception_amd/tools/bench_frontend.py $a 2>&1 | grep -v amdgpu.ids; done > $O/r06_frontend_old_vs_new.txt; cat $O/r06_frontend_old_vs_new.txt
