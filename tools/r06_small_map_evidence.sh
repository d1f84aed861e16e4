#!/bin/bash
# r06_small_map_evidence.sh (on the GPU box): rocprofv3 kernel stats and PMC passes of the RDN trunk at 48 x 48 (the reference's
# timing protocol) with the small-map kernels and with them switched off -> gpurun_out/small/ (copy the summaries to profiles/).
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/small
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
T="python3 $R/tools/enc_trunk_time.py 48 --only-hip"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_t16 -- $T > $O/stats_t16.log 2>&1
DIINN_ENC_NO_T16=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ks -- $T > $O/stats_ks.log 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_sq1 -- $T > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_sq2 -- $T > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d $O/pmc_sq3 -- $T > $O/pmc_sq3.log 2>&1
cd $R
python tools/summarize_rocprof.py $O/stats_t16 $O/r06_enc_trunk_48_kernel_stats.csv "tools/enc_trunk_time.py 48 --only-hip (13 forwards)" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/stats_ks $O/r06_enc_trunk_48_no_t16_kernel_stats.csv "DIINN_ENC_NO_T16=1 tools/enc_trunk_time.py 48 --only-hip (13 forwards)" > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 2>&1 | grep -A10 "conv_t16\|conv1x1_t16\|conv_ksplit" > $O/r06_enc_trunk_48_pmc.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
du -sh $O
