#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  -> variants/libdiinn_NAME.so (kernel A/B experiments)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC="$ROOT/dual-interactive-implicit-neural-network_amd/csrc"
NAME=$1; shift
mkdir -p "$ROOT/variants/obj"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc "$@" -c "$CSRC/diinn_kernels.hip" -o "$ROOT/variants/obj/k_$NAME.o"
hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -x c++ -c "$CSRC/diinn_host.cpp" -o "$ROOT/variants/obj/h_$NAME.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/variants/libdiinn_$NAME.so" "$ROOT/variants/obj/k_$NAME.o" "$ROOT/variants/obj/h_$NAME.o"
echo "$ROOT/variants/libdiinn_$NAME.so"
