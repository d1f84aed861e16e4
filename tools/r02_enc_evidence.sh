# Round-2 encoder evidence on one MI355X box: trunk times vs MIOpen, whole-model times, per-layer trace, PMC, stamps.
O=gpurun_out/r02enc; mkdir -p $O
python tools/enc_trunk_time.py 48 96 128 192 256 384 512 > $O/enc_trunk_times.txt 2>&1
python tools/e2e_time.py > $O/e2e_times.txt 2>&1
bash tools/r02_enc_trace.sh final 256 > $O/trace256.log 2>&1; cp gpurun_out/enc_layers_final.txt $O/enc_layers_256.txt
bash tools/r02_enc_trace.sh final512 512 > $O/trace512.log 2>&1; cp gpurun_out/enc_layers_final512.txt $O/enc_layers_512.txt
bash tools/r02_enc_pmc.sh final 256 > $O/pmc.log 2>&1; cp gpurun_out/encpmc_final_summary.txt $O/enc_pmc_summary.txt
if [ -f variants/libdiinn_stamps.so ]; then
  for c in 64 512; do DIINN_HIP_LIB=variants/libdiinn_stamps.so python tools/stamp_report_enc.py 256 $c 9 wino; done > $O/wino_stamps.txt 2>&1
fi
timeout 600 python -m pytest tests/test_encoder_trunk.py tests/test_modules.py tests/test_scripts.py -q -m gpu 2>&1 | tail -3 > $O/tests.txt
for f in enc_trunk_times e2e_times tests; do tail -n 3 $O/$f.txt; done
