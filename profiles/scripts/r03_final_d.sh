#!/bin/bash
# round 3, final measurements part D: rocprofv3 kernel stats of the DiscoNet TRAINING step (fp32 and bf16 products) and a PMC pass
# (SQ counters) of the fp32 training step, with the final kernels (fused F(4x4) forward / data gradient, eight-wave weight gradient)
R=$GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train -- python3 $R/bench.py --train --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train_bf16 -- python3 $R/bench.py --train --conv-algo bf16 --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_train_bf16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc_train_sq -- python3 $R/bench.py --train --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_train_sq.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_train_lds -- python3 $R/bench.py --train --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_train_lds.log 2>&1
cd $R
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete
S=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); cp $S $O/train_disco_b4_kernel_stats.csv; head -12 $O/train_disco_b4_kernel_stats.csv | cut -c1-150
S=$(find $O/prof_train_bf16 -name "*kernel_stats.csv" | head -1); cp $S $O/train_disco_b4_bf16_kernel_stats.csv; head -6 $O/train_disco_b4_bf16_kernel_stats.csv | cut -c1-150
Q=$(find $O/pmc_train_sq -name "*counter_collection.csv" | head -1); L=$(find $O/pmc_train_lds -name "*counter_collection.csv" | head -1)
python3 practical-collab-perception_amd/tools/pmc_sq_summary.py $O/r03_pmc_sq_counters_train.json $Q $L | head -60
find $O -name "*counter_collection.csv" -delete
