#!/bin/bash
# kernel_resources.sh > profiles/<round>_kernel_resources.txt : registers / spills / LDS of every kernel
# (hipcc -Rpass-analysis=kernel-resource-usage over the gfx950 translation units)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC="$ROOT/dual-interactive-implicit-neural-network_amd/csrc"
echo "# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage csrc/diinn_*.hip"
for f in "$CSRC"/diinn_*.hip; do
  echo "## $(basename "$f")"
  EXTRA=""                                                     # the per-file flags of build.py (PER_FILE_FLAGS)
  case "$(basename "$f")" in diinn_decode.hip|diinn_bf16x3.hip) EXTRA="-mllvm -amdgpu-mfma-vgpr-form";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc $EXTRA -Rpass-analysis=kernel-resource-usage \
      -c "$f" -o /dev/null 2>&1 |
    grep -oE "(Function Name: [^ ]+|VGPRs: [0-9]+|AGPRs: [0-9]+|ScratchSize \[bytes/lane\]: [0-9]+|VGPRs Spill: [0-9]+|Occupancy \[waves/SIMD\]: [0-9]+|LDS Size \[bytes/block\]: [0-9]+)" |
    awk '/Function Name/{if(l)print l; l=$0; next}{l=l" "$0}END{if(l)print l}'
done
