#!/bin/bash
# round 6 final measurements, part D: the LiDAR-like cloud (--dist ring) under rocprofv3 (single stream, the run the per-kernel table is
# read from) and the PCIe-inclusive line (--host-input)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ring -- python3 $R/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-secondary --no-configs --no-overlap --no-pipeline --dist ring > $O/prof_ring.log 2>&1 < /dev/null
f=$(ls $O/prof_ring/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r06_bench_disco_ring_b4_single_stream_kernel_stats.csv && head -12 "$f" | cut -c1-150; rm -rf $O/prof_ring
cd $R
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-configs --host-input 2>/dev/null | grep '^{' > $O/r06_bench_disco_host_input.json
timeout 300 python bench.py --config ego --steps 20 --warmup 5 --no-cpu-baseline --host-input 2>/dev/null | grep '^{' > $O/r06_bench_ego_host_input.json
cut -c1-200 $O/r06_bench_disco_host_input.json $O/r06_bench_ego_host_input.json
