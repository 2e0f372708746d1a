#!/bin/bash
# round 6 final measurements, part C: PMC passes of the headline command (single stream): FETCH_SIZE / WRITE_SIZE (bench.py's roofline.traffic)
# and the SQ / LDS counters of the dominant kernels; every --pmc pass on its own with --kernel-trace only, the program directly after `--`
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r6pmcfinal; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-configs --no-overlap --no-pipeline > $R/$O/$name.log 2>&1 < /dev/null; }
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass pmc_lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
cd $R
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python3 practical-collab-perception_amd/tools/pmc_summary.py disco $F $W gpurun_out/r06_pmc_traffic.json | head -30; else echo "no traffic counters"; tail -3 $O/pmc_fetch.log; fi
Q=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); L=$(find $O/pmc_lds -name "*counter_collection.csv" | head -1)
if [ -n "$Q" ] && [ -n "$L" ]; then python3 practical-collab-perception_amd/tools/pmc_sq_summary.py gpurun_out/r06_pmc_sq_counters.json $Q $L; else echo "no SQ counters"; tail -3 $O/pmc_sq.log; fi
find $O -name "*counter_collection.csv" -delete
