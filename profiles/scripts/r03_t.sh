#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
python bench.py --no-cpu-baseline --layer-table $O/layers_disco.txt > $O/bench_disco.json 2> $O/err1.txt
python bench.py --no-cpu-baseline --train --layer-table $O/layers_disco_train.txt > $O/bench_disco_train.json 2> $O/err2.txt
tail -3 $O/err1.txt $O/err2.txt
head -40 $O/layers_disco.txt
