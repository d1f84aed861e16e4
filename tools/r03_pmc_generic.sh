#!/bin/bash
# separate rocprofv3 --pmc passes (MI355X_MICROARCH.md) of one bench.py command line:
#   bash tools/r03_pmc_generic.sh TAG KERNEL_GREP <bench args ...>      -> gpurun_out/r03pmc_TAG/pmc_summary.txt
TAG=$1; KG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03pmc_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py $* --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-target --no-strong --no-traffic --no-side-legs --no-split"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
cd $R
python tools/pmc_summary.py $O/sq1 $O/sq2 $O/fetch $O/write > $O/pmc_summary.txt 2>&1
python tools/summarize_rocprof.py $O/stats $O/kernel_stats.csv "bench.py $* --steps 5 --warmup 2" > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep -A12 "$KG" $O/pmc_summary.txt | head -80
head -6 $O/kernel_stats.csv | cut -c1-140
