#!/bin/bash
# r04: what is left in decode_bf16_coop8p_kernel at c5 once layer 0, refill, seeds, epilogue, B reads and layer barriers are
# compiled out (m0 = the "all removed" build of r04_bf16_abl.sh): head, last-layer parking, per-block coordinates, block barriers
for rep in 1 2; do
for v in m0 m_nohead m_nopark m_nocoord m_nosync0 m_all; do
  DIINN_HIP_LIB=variants/libdiinn_r4_$v.so python tools/bf16_time.py c5 10 2>/dev/null
done
done
