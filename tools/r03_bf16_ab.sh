#!/bin/bash
# A/B of the bf16 decode kernels on one box: DIINN_BF16_KERNEL = 8 (one block per workgroup) vs 9 (persistent), c5 and c2
for wl in c5 c2; do for k in 8 9 8 9; do DIINN_BF16_KERNEL=$k timeout 300 python tools/bf16_time.py $wl 10 2>&1 | grep -v amdgpu.ids; done; done
