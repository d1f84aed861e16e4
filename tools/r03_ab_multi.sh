#!/bin/bash
# A/B of several library variants against the shipped build inside one box: bash tools/r03_ab_multi.sh "v1 v2 .." [bench args]
VARS=$1; shift
for i in 1 2; do for v in shipped $VARS; do
  if [ $v = shipped ]; then L=""; else L=variants/libdiinn_$v.so; fi
  DIINN_HIP_LIB=$L python bench.py "$@" --no-cpu-baseline --no-target --no-traffic --no-side-legs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('%-10s %.2f Mpix/s  step %.4f ms  decode %.4f  P %.4f ok=%s' % ('$v', r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['p_kernel']['ms'], r['checked']['ok']))"
done; done
