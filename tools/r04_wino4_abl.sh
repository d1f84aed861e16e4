#!/bin/bash
# timing ablations of conv_wino4_kernel (tools/ubench/wino4_bench.hip); built here, run on the GPU box:
#   bash tools/r04_wino4_abl.sh build ; gpurun -- 'bash tools/r04_wino4_abl.sh run [SIZE]'
cd "$(dirname "$0")/.."
VARS="base SYM SYM+NOTR SYM+FENCE EMAJOR STAMPS MFMAFETCH MIXED NOTRANSFORM NOPATCH NOMFMA NOW NOTRANSFORM+NOPATCH NOW+NOTRANSFORM+NOPATCH NOMFMA+NOW"
if [ "$1" = build ]; then
  for v in $VARS; do
    fl=""; for a in ${v//+/ }; do case $a in base) ;; STAMPS) fl="$fl -DW4_STAMPS" ;; MFMAFETCH) fl="$fl -DW4_MFMA_FETCH" ;; EMAJOR) fl="$fl -DW4_EMAJOR" ;; SYM) fl="$fl -DW4_SYM_KERNEL" ;; NOTR) fl="$fl -DSYM_NOTRANSFORM" ;; FENCE) fl="$fl -DSYM_FENCE" ;; *) fl="$fl -DW4_ABL_$a" ;; esac; done
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I include $fl tools/ubench/wino4_bench.hip -o tools/ubench/wino4_bench_$v &
  done; wait; ls tools/ubench/wino4_bench_* | wc -l
else
  for v in $VARS; do printf "%-28s" $v; ./tools/ubench/wino4_bench_$v ${2:-256}; done
fi
