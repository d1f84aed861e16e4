#!/bin/bash
# A/B of the bf16 decode at c5 / c2: variants/libdiinn_r5base.so (before) against the tree's library, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for lib in variants/libdiinn_r5base.so ""; do
    for wl in c5 c2; do DIINN_HIP_LIB=$lib python tools/bf16_time.py $wl 10 2>/dev/null; done
  done
done
