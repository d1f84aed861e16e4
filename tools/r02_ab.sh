# A/B of library variants inside one box: r02_ab.sh "base ns4 ..." [size]
for v in $1; do
  if [ "$v" != base ]; then export DIINN_HIP_LIB=variants/libdiinn_$v.so; else unset DIINN_HIP_LIB; fi
  echo "== $v"; python tools/enc_trunk_time.py ${2:-256} --only-hip | tail -1
done
