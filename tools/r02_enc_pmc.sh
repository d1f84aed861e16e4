# PMC passes over the encoder trunk (separate rocprofv3 runs): r02_enc_pmc.sh TAG [SIZE]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/encpmc_${1:-enc}; mkdir -p $O
B="python3 $R/tools/enc_trunk_time.py ${2:-256} --only-hip"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/tcp -- $B > $O/tcp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --output-format csv -d $O/ta -- $B > $O/ta.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/mem -- $B > $O/mem.log 2>&1
cd $R
python tools/pmc_summary.py $O/sq1 $O/sq2 $O/tcp $O/ta $O/mem > gpurun_out/encpmc_${1:-enc}_summary.txt 2>&1
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
cat gpurun_out/encpmc_${1:-enc}_summary.txt
