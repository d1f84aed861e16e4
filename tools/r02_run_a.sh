#!/bin/bash
# round 2, GPU run A: full gpu test suite, bench lines for every BASELINE workload, rocprof stats + PMC of the bf16 path
set -u
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.err
for wl in c1 c3 tgt c4 c5; do
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
done
python bench.py --workload c5 --compute bf16 --no-cpu-baseline > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
python bench.py --workload c5 --compute bf16_full --no-cpu-baseline > $O/bench_c5_bf16_full.json 2> $O/bench_c5_bf16_full.err
python bench.py --workload c2 --compute bf16_full --no-cpu-baseline > $O/bench_c2_bf16_full.json 2> $O/bench_c2_bf16_full.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c3 --scaling strong --gather --no-cpu-baseline > $O/bench_c3_strong_torchrun.json 2> $O/bench_c3_strong_torchrun.err
cat $O/bench_*.json
rocprofv3 -L > $O/counters_list.txt 2>&1
# kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check > $O/stats_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5bf -- python3 bench.py --workload c5 --compute bf16_full --steps 10 --warmup 3 --no-cpu-baseline --no-check > $O/stats_c5bf.log 2>&1
# PMC passes (separate), bf16_full at c5
B="python3 bench.py --workload c5 --compute bf16_full --steps 5 --warmup 2 --no-cpu-baseline --no-check"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_bf_sq -- $B > $O/pmc_bf_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_bf_sq2 -- $B > $O/pmc_bf_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/pmc_bf_tcp -- $B > $O/pmc_bf_tcp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_bf_fetch -- $B > $O/pmc_bf_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_bf_write -- $B > $O/pmc_bf_write.log 2>&1
python tools/pmc_summary.py $O/pmc_bf_sq $O/pmc_bf_sq2 $O/pmc_bf_tcp $O/pmc_bf_fetch $O/pmc_bf_write > $O/pmc_bf_summary.txt 2>&1
python tools/summarize_rocprof.py $O/stats_c2 $O/stats_c2_summary.csv "bench.py c2 f32" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/stats_c5bf $O/stats_c5bf_summary.csv "bench.py c5 bf16_full" > /dev/null 2>&1
# keep the merge small: drop raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
