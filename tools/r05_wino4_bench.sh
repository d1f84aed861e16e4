#!/bin/bash
# conv_wino4_kernel alone: whole rounds vs the split of the last round over the input channels (tools/ubench/wino4_bench.hip).
#   bash tools/r05_wino4_bench.sh build ; gpurun -- 'bash tools/r05_wino4_bench.sh run 192 256 320 384'
cd "$(dirname "$0")/.."
C=dual-interactive-implicit-neural-network_amd/csrc
if [ "$1" = build ]; then
  g++ -O2 -std=c++17 -fPIC -ffp-contract=off -c $C/diinn_host.cpp -o /tmp/diinn_host_ub.o || exit 1
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I include ${W4_FLAGS} -c tools/ubench/wino4_bench.hip -o /tmp/wino4_bench_ub.o || exit 1
  hipcc --offload-arch=gfx950 /tmp/wino4_bench_ub.o /tmp/diinn_host_ub.o -o tools/ubench/wino4_bench${W4_SUFFIX} || exit 1
  ls -la tools/ubench/wino4_bench${W4_SUFFIX}
else
  shift
  ./tools/ubench/wino4_bench${W4_SUFFIX} "$@"
fi
