#!/bin/bash
# extra PMC passes on the c5 bf16_full step for precompute_P_bf16_wide_kernel: LDS conflicts, VMEM instruction cycles, occupancy
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03pbf; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload c5 --compute bf16_full --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-target"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/a -- $B > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_IFETCH --output-format csv -d $O/b -- $B > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $O/c -- $B > $O/c.log 2>&1
cd $R
python tools/pmc_summary.py $O/a $O/b $O/c 2>&1 | sed "s#$R/gpurun_out/##" | grep -A10 "precompute_P_bf16\|decode_bf16" 
