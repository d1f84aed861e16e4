#!/bin/bash
# order-balanced A/B of library variants inside one box (a run's position in the sequence biases it by ~0.4 %):
#   bash tools/r03_ab_abba.sh "v1 v2 v3" [bench args]     -> v1 v2 v3 v3 v2 v1 v1 v2 v3 ...
VARS=$1; shift
REV=$(echo $VARS | tr ' ' '\n' | tac | tr '\n' ' ')
for v in $VARS $REV $VARS $REV; do
  DIINN_HIP_LIB=variants/libdiinn_$v.so python bench.py "$@" --no-cpu-baseline --no-target --no-traffic --no-side-legs --no-split 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('%-8s step %.4f ms  decode %.4f  P %.4f ok=%s' % ('$v', r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['p_kernel']['ms'], r['checked']['ok']))"
done
