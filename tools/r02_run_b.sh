#!/bin/bash
# bf16 decode kernel variants: parity tests with each forced, then timing at c5 / c2
set -u
O=gpurun_out/r02b; mkdir -p $O
for k in 1 2 4 8; do
  DIINN_BF16_KERNEL=$k python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "bf16" > $O/pytest_k$k.log 2>&1
  echo "variant $k: $(tail -1 $O/pytest_k$k.log)"
done
for k in 2 4 8; do
  for wl in c5 c2; do
    DIINN_BF16_KERNEL=$k python bench.py --workload $wl --compute bf16_full --no-cpu-baseline > $O/bench_${wl}_k$k.json 2>$O/bench_${wl}_k$k.err
    python - <<PY
import json
r=json.load(open("$O/bench_${wl}_k$k.json"))
print("variant $k $wl: step %.3f ms  decode %.3f ms (min %.3f)  P %.3f ms  frac %.3f  err %.2e ok=%s" % (r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["kernel_ms_min"], r["roofline"]["p_kernel"]["ms"], r["roofline"]["frac"], r["checked"]["max_err"], r["checked"]["ok"]))
PY
  done
done
