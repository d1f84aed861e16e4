#!/bin/bash
# whole-step A/B (bench.py) of the bf16 decode kernels: DIINN_BF16_KERNEL=8 (one block per workgroup) vs default (persistent)
for wl in c5 c2; do for comp in bf16_full bf16; do for k in 8 0 8 0; do
  DIINN_BF16_KERNEL=$k python bench.py --workload $wl --compute $comp --no-cpu-baseline --no-target --no-traffic --no-side-legs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$wl $comp k=$k  %.1f Mpix/s  step %.3f ms  decode %.3f  P %.3f  err %.2e ok=%s' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['p_kernel']['ms'], r['checked']['max_err'], r['checked']['ok']))"
done; done; done
