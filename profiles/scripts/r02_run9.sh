cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_disco -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/prof_disco.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/$O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_lds -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc_lds.log 2>&1
cd $R
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -size +30M -delete
find $O -type f | head -40; du -sh $O
tail -3 $O/pmc_sq.log
