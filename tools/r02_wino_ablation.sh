for v in ${VARIANTS:-"" wn_NOW wn_NOROWS wn_NOVALU wn_NOMEM wn_NONE}; do
  if [ "$v" != base ]; then export DIINN_HIP_LIB=variants/libdiinn_$v.so; fi
  echo "== ${v:-base}"; python tools/enc_layer_time.py 256 --relu --wino 2>&1 | grep "Cin   512 taps 9\|Cin    64 taps 9"
done
