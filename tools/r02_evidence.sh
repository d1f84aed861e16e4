#!/bin/bash
# round 2 evidence run (one MI355X box): gpu tests, bench lines of every BASELINE workload, rocprofv3 kernel stats and
# PMC passes (separate, as the guide prescribes) of the c2 fp32 step and the c5 bf16_full step, coop-kernel stamps.
#   gpurun --timeout 2400 -- 'bash tools/r02_evidence.sh TAG'
set -u
TAG=${1:-r02}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for wl in c1 c3 tgt c4 c5; do
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
done
python bench.py --workload c5 --compute bf16 --no-cpu-baseline > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
python bench.py --workload c5 --compute bf16_full --no-cpu-baseline > $O/bench_c5_bf16_full.json 2> $O/bench_c5_bf16_full.err
python bench.py --workload c2 --compute bf16_full --no-cpu-baseline > $O/bench_c2_bf16_full.json 2> $O/bench_c2_bf16_full.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c3 --scaling strong --gather --no-cpu-baseline > $O/bench_c3_strong_torchrun.json 2> $O/bench_c3_strong_torchrun.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        r = json.load(open(f))
    except Exception as e:
        print(f, "UNREADABLE", e); continue
    print("%-34s %9.2f Mpix/s  step %8.3f ms  decode %8.3f ms frac %.3f  P %7.3f ms (%.3f)  err %.2e ok=%s" % (
        f.split("/")[-1], r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"],
        r["roofline"]["p_kernel"]["ms"], r["roofline"]["p_kernel"]["frac"], r["checked"]["max_err"], r["checked"]["ok"]))
PY
# kernel stats
C2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check"
C5="python3 bench.py --workload c5 --compute bf16_full --steps 10 --warmup 3 --no-cpu-baseline --no-check"
C1="python3 bench.py --workload c1 --steps 50 --warmup 5 --no-cpu-baseline --no-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- $C2 > $O/stats_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5bf -- $C5 > $O/stats_c5bf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c1 -- $C1 > $O/stats_c1.log 2>&1
python tools/summarize_rocprof.py $O/stats_c2 $O/stats_c2_summary.csv "bench.py (c2 f32) --steps 10 --warmup 3" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/stats_c5bf $O/stats_c5bf_summary.csv "bench.py --workload c5 --compute bf16_full --steps 10 --warmup 3" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/stats_c1 $O/stats_c1_summary.csv "bench.py --workload c1 --steps 50 --warmup 5" > /dev/null 2>&1
# PMC: separate passes
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
for cfg in c2 c5; do
  if [ $cfg = c2 ]; then B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check"; else B="python3 bench.py --workload c5 --compute bf16_full --steps 5 --warmup 2 --no-cpu-baseline --no-check"; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_${cfg}_sq1 -- $B > $O/pmc_${cfg}_sq1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_${cfg}_sq2 -- $B > $O/pmc_${cfg}_sq2.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/pmc_${cfg}_tcp -- $B > $O/pmc_${cfg}_tcp.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${cfg}_fetch -- $B > $O/pmc_${cfg}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${cfg}_write -- $B > $O/pmc_${cfg}_write.log 2>&1
  python tools/pmc_summary.py $O/pmc_${cfg}_sq1 $O/pmc_${cfg}_sq2 $O/pmc_${cfg}_tcp $O/pmc_${cfg}_fetch $O/pmc_${cfg}_write > $O/pmc_${cfg}_summary.txt 2>&1
done
# stamps of the cooperative bf16 kernel (variant library built beforehand)
if [ -f variants/libdiinn_stamps.so ]; then
  DIINN_HIP_LIB=variants/libdiinn_stamps.so python tools/stamp_report_coop.py c5 > $O/bf16_coop8_stamps.txt 2>&1; DIINN_BF16_KERNEL=4 DIINN_HIP_LIB=variants/libdiinn_stamps.so python tools/stamp_report_coop.py c5 > $O/bf16_coop4_stamps.txt 2>&1
fi
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
