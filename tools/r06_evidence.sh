#!/bin/bash
# round 6 evidence run (one MI355X box): gpu tests with durations, the default bench line (driver's command), bench lines of
# the other BASELINE workloads, rocprofv3 --kernel-trace --stats of the c2 fp32 / c5 bf16_full / c1 steps, PMC passes of c2.
#   gpurun --timeout 2400 -- 'bash tools/r06_evidence.sh r06_ev'
set -u
TAG=${1:-r06_ev}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
X="--no-cpu-baseline --no-target --no-traffic --no-side-legs --no-split"
for wl in c3 tgt c4 c5; do
  python bench.py --workload $wl --steps 12 --warmup 3 $X > $O/bench_$wl.json 2> $O/bench_$wl.err
done
python bench.py --workload c1 --steps 200 --warmup 20 $X > $O/bench_c1_200.json 2> $O/bench_c1_200.err
python bench.py --workload c5 --compute bf16 $X > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
python bench.py --workload c5 --compute bf16_full $X > $O/bench_c5_bf16_full.json 2> $O/bench_c5_bf16_full.err
python bench.py --workload c2 --compute bf16_full $X > $O/bench_c2_bf16_full.json 2> $O/bench_c2_bf16_full.err
python bench.py --workload c2 --compute bf16x3 $X > $O/bench_c2_bf16x3.json 2> $O/bench_c2_bf16x3.err
DIINN_BENCH_ONE_DEVICE=1 DIINN_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 8 --steps 3 --warmup 1 > $O/bench_8ranks_one_device_gloo.json 2> $O/bench_8ranks_one_device_gloo.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        r = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
    except Exception as e:
        print(f, "UNREADABLE", e); continue
    print("%-38s %9.2f Mpix/s  step %8.4f ms  decode %8.4f ms frac %.3f  P %7.4f ms (%.3f)  err %.2e ok=%s" % (
        f.split("/")[-1], r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"],
        r["roofline"]["p_kernel"]["ms"], r["roofline"]["p_kernel"]["frac"], r["checked"]["max_err"], r["checked"]["ok"]))
    for leg in r.get("side_legs", []):
        print("    side leg %-4s %-10s step %8.4f ms decode %8.4f ms frac %.3f ok=%s" % (leg["name"], leg["compute"], leg["ms_per_step"], leg["kernel_ms"], leg["frac"], leg["checked"]["ok"]))
    for leg in r.get("strong", []):
        print("    strong   %-4s N=%d  1 GPU %8.3f ms  N GPUs %8.3f ms  checked_ok=%s" % (leg["workload"], leg["n_gpus"], leg["ms_1gpu"], leg["ms_Ngpu"], leg.get("checked_ok")))
    if "traffic_source" in r["roofline"] and r["roofline"]["traffic"]:
        print("    traffic %d B: %s" % (r["roofline"]["traffic"], r["roofline"]["traffic_source"][:60]))
PY
cd /tmp
C2="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-target --no-traffic --no-side-legs --no-split"
C5="python3 $R/bench.py --workload c5 --compute bf16_full --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-target --no-traffic --no-side-legs"
C1="python3 $R/bench.py --workload c1 --steps 50 --warmup 5 --no-cpu-baseline --no-check --no-target --no-traffic --no-side-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- $C2 > $O/stats_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5bf -- $C5 > $O/stats_c5bf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c1 -- $C1 > $O/stats_c1.log 2>&1
cd $R
python tools/summarize_rocprof.py $O/stats_c2 $O/r06_c2_kernel_stats.csv "bench.py (c2 f32) --steps 10 --warmup 3" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/stats_c5bf $O/r06_c5_bf16_full_kernel_stats.csv "bench.py --workload c5 --compute bf16_full --steps 10 --warmup 3" > /dev/null 2>&1
python tools/summarize_rocprof.py $O/stats_c1 $O/r06_c1_kernel_stats.csv "bench.py --workload c1 --steps 50 --warmup 5" > /dev/null 2>&1
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-target --no-traffic --no-side-legs --no-split"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_c2_sq1 -- $B > $O/pmc_c2_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_c2_sq2 -- $B > $O/pmc_c2_sq2.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_c2_sq1 $O/pmc_c2_sq2 > $O/r06_c2_pmc_summary.txt 2>&1
# encoder: trunk times over map sizes (split by the cost model), kernel stats of the trunk at 192 / 384 (the split kernel's launches), e2e times
python tools/enc_trunk_time.py 48 96 128 144 160 192 224 256 320 384 448 512 2>&1 | grep -v amdgpu.ids > $O/enc_trunk_times.txt
python tools/e2e_time.py > $O/e2e_times.txt 2>&1
cd /tmp
for s in 192 384; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_enc$s -- python3 $R/tools/enc_trunk_time.py $s --only-hip > $O/stats_enc$s.log 2>&1
done
cd $R
for s in 192 384; do python tools/summarize_rocprof.py $O/stats_enc$s $O/r06_enc_trunk_${s}_kernel_stats.csv "tools/enc_trunk_time.py $s --only-hip" > /dev/null 2>&1; done
# round 6: the decoder's training step (kernel stats), PMC of conv_wino4_kernel whole vs split at 192 / 384
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 $R/tools/train_time.py --only-ours > $O/stats_train.log 2>&1
cd $R
python tools/summarize_rocprof.py $O/stats_train $O/r06_train_kernel_stats.csv "tools/train_time.py --only-ours (decoder training step, B = 16, 48x48 -> 192x192)" > /dev/null 2>&1
python tools/train_time.py > $O/train_time.txt 2>&1
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
for s in 192 384; do for split in 1 0; do
  T="python3 $R/tools/enc_trunk_time.py $s --only-hip"
  DIINN_ENC_WINO4_SPLIT=$split timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_enc${s}_split${split}_sq1 -- $T > /dev/null 2>&1
  DIINN_ENC_WINO4_SPLIT=$split timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_enc${s}_split${split}_sq2 -- $T > /dev/null 2>&1
done; done
cd $R
for s in 192 384; do for split in 1 0; do
  echo "==== enc_trunk_time.py $s, DIINN_ENC_WINO4_SPLIT=$split (1 = the shipped cost model, 0 = whole items only)"
  python tools/pmc_summary.py $O/pmc_enc${s}_split${split}_sq1 $O/pmc_enc${s}_split${split}_sq2 | grep -A9 "conv_wino4\|conv1x1_stream"
done; done > $O/r06_enc_wino4_pmc.txt 2>&1
python tools/clock_trace.py 2.0 > $O/r06_clock_trace_all.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
du -sh $O
