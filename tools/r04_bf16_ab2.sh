#!/bin/bash
# r04: compiler-flag A/B of the bf16 decode at c5 and c2 (order-balanced)
for rep in 1 2; do
for v in base noslp vgprf noslpvg noslpvg vgprf noslp base; do
  DIINN_HIP_LIB=variants/libdiinn_r4_$v.so python tools/bf16_time.py c5 10 2>/dev/null
done
done
for v in base noslp vgprf noslpvg; do
  DIINN_HIP_LIB=variants/libdiinn_r4_$v.so python tools/bf16_time.py c2 20 2>/dev/null
  COMPUTE=4 DIINN_HIP_LIB=variants/libdiinn_r4_$v.so python tools/bf16_time.py c2 20 2>/dev/null | sed 's/$/ (x3)/'
done
