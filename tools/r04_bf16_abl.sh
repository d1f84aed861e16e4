#!/bin/bash
# r04: where decode_bf16_coop8p_kernel's time goes at c5 (timing-only ablation builds; results are wrong by construction)
#   bash tools/r04_bf16_abl.sh > gpurun_out/r04_b/abl.txt
for rep in 1 2; do
for v in base nol0 norefill noseed noepi nolds nobar mfmaonly; do
  DIINN_HIP_LIB=variants/libdiinn_r4_$v.so python tools/bf16_time.py c5 10 2>/dev/null
done
done
SIN=1 DIINN_HIP_LIB=variants/libdiinn_r4_base.so python tools/bf16_time.py c5 10 2>/dev/null | sed 's/$/  (sin hw)/'
DIINN_BF16_KERNEL=8 DIINN_HIP_LIB=variants/libdiinn_r4_base.so python tools/bf16_time.py c5 10 2>/dev/null
DIINN_BF16_KERNEL=4 DIINN_HIP_LIB=variants/libdiinn_r4_base.so python tools/bf16_time.py c5 10 2>/dev/null
DIINN_HIP_LIB=variants/libdiinn_r4_base.so python tools/bf16_time.py c2 20 2>/dev/null
DIINN_HIP_LIB=variants/libdiinn_r4_mfmaonly.so python tools/bf16_time.py c2 20 2>/dev/null
