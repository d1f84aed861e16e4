#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  -> variants/libdiinn_NAME.so (kernel A/B experiments;
# select it with DIINN_HIP_LIB=variants/libdiinn_NAME.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/variants"
cd "$ROOT"
python -c 'import sys, diinn_amd.build as b; print(b.build(force=True, verbose=False, extra_flags=tuple(sys.argv[2:]), out=sys.argv[1]))' "$ROOT/variants/libdiinn_$NAME.so" "$@"
