#!/bin/bash
# one SQ counter pass per split-bf16 decode kernel (DIINN_X3_KERNEL = 1 one block, 2 persistent, 3 shared weight stream)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for K in ${KERNELS:-1 2 3}; do
  export DIINN_X3_KERNEL=$K
  O=$R/gpurun_out/r03x3pmc_k$K; rm -rf $O; mkdir -p $O
  B="python3 $R/bench.py --compute bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-target --no-strong"
  SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
  SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
  (cd $R && python tools/pmc_summary.py $O/sq1 $O/sq2 > $O/pmc_summary.txt 2>&1)
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
  echo "== DIINN_X3_KERNEL=$K"; grep -A9 "decode_bf16x3" $O/pmc_summary.txt | grep -v "^--"
done
