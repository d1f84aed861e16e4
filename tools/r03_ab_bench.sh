for i in 1 2 3; do for v in shipped ${VARIANT:-nowarm}; do
  if [ $v = shipped ]; then L=""; else L=variants/libdiinn_$v.so; fi
  DIINN_HIP_LIB=$L python bench.py $BARGS --no-cpu-baseline --no-target --no-traffic --no-side-legs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$v  %.2f Mpix/s  step %.4f ms  decode %.4f  P %.4f (min %.4f) ok=%s' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['p_kernel']['ms'], r['roofline']['p_kernel']['ms_min'], r['checked']['ok']))"
done; done
