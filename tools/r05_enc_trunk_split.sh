#!/bin/bash
# RDN trunk per map size with the F(4x4,3x3) kernel's last round whole (DIINN_ENC_WINO4_SPLIT=0), split by the cost model (1), always split (2)
cd "$(dirname "$0")/.."
for m in 0 1 2; do
  echo "# DIINN_ENC_WINO4_SPLIT=$m"
  DIINN_ENC_WINO4_SPLIT=$m python tools/enc_trunk_time.py --only-hip "$@"
done
