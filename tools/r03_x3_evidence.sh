#!/bin/bash
# evidence for DESIGN 3.5: PMC passes + kernel stats of the split-bf16 c2 step for both decode forms, the issue
# microbenchmark, and the default bench line (fp32 headline + split_bf16 side leg).  Output: gpurun_out/r03x3/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03x3; rm -rf $O; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_issue tools/ubench/mfma_issue.hip 2> $O/ubench_build.log
timeout 120 /tmp/mfma_issue > $O/ubench_mfma_issue.txt 2>&1
for K in 2 1; do
  DIINN_X3_KERNEL=$K bash tools/r03_pmc_generic.sh x3k$K decode_bf16x3 --compute bf16x3 > $O/pmc_k$K.log 2>&1
  cp gpurun_out/r03pmc_x3k$K/pmc_summary.txt $O/pmc_summary_k$K.txt
  cp gpurun_out/r03pmc_x3k$K/kernel_stats.csv $O/kernel_stats_k$K.csv
done
python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python bench.py --compute bf16x3 --no-cpu-baseline 2> $O/bench_x3.err | tail -1 > $O/bench_x3.json
DIINN_X3_KERNEL=1 python bench.py --compute bf16x3 --no-cpu-baseline 2> /dev/null | tail -1 > $O/bench_x3_oneblock.json
python - <<PY
import json
for f in ("bench_default","bench_x3","bench_x3_oneblock"):
    r=json.load(open("$O/%s.json"%f)); print(f, r["dtype"], r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["checked"]["ok"], r.get("split_bf16",{}).get("mpix_s"))
PY
