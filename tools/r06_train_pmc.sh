#!/bin/bash
# PMC passes of the decoder training step (tools/train_time.py --only-ours): where do bwd_layer_kernel / plane_gemm_kernel wait?
#   gpurun -- 'bash tools/r06_train_pmc.sh'   -> gpurun_out/r06_train_pmc/pmc_summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_train_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/train_time.py --only-ours"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
cd $R
python tools/pmc_summary.py $O/sq1 $O/sq2 $O/fetch $O/write > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep -A10 "bwd_layer\|plane_gemm\|decode_kernel<2, true, true\|rowdot\|cell_sum" $O/pmc_summary.txt | head -150
