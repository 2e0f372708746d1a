#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_disco.json 2> $O/bench_disco.err
tail -1 $O/bench_disco.json | cut -c1-1500
