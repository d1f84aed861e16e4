#!/bin/bash
# r04: the 16-pixel fp32 latency kernel at c1 (48x48 -> 96x96): workgroups per CU (LDS padding) and kernel choice
for rep in 1 2; do
for v in c16p4 c16p3 c16p2; do
for k in 3 2; do
  DIINN_F32_KERNEL=$k DIINN_HIP_LIB=variants/libdiinn_r4_$v.so python bench.py --workload c1 --steps 200 --warmup 20 --no-cpu-baseline --no-target --no-split --no-side-legs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('$v k=$k step %.4f ms decode %.4f (min %.4f) P %.4f ok=%s' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['kernel_ms_min'], r['roofline']['p_kernel']['ms'], r['checked']['ok']))"
done; done; done
