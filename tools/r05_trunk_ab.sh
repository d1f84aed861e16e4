#!/bin/bash
# RDN trunk A/B: variants/libdiinn_r5base.so (before) against the tree's library, interleaved;  bash tools/r05_trunk_ab.sh SIZE...
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for lib in variants/libdiinn_r5base.so ""; do
    echo "# lib=[$lib]"; DIINN_HIP_LIB=$lib python tools/enc_trunk_time.py --only-hip "$@" 2>&1 | grep "HIP trunk"
  done
done
