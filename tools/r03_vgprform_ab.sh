for i in 1 2; do
echo "== shipped"; python tools/liif_time.py 256 4 --only-ours 2>&1 | grep "HIP path"; python tools/metasr_time.py 256 4 --only-ours 2>&1 | grep "HIP path"; python tools/train_time.py 16 48 4 --only-ours 2>&1 | grep "ours\|training forward"; python tools/enc_trunk_time.py 48 256 --only-hip 2>&1 | grep "HIP trunk"
echo "== vf_base"; DIINN_HIP_LIB=variants/libdiinn_vf_base.so python tools/liif_time.py 256 4 --only-ours 2>&1 | grep "HIP path"; DIINN_HIP_LIB=variants/libdiinn_vf_base.so python tools/metasr_time.py 256 4 --only-ours 2>&1 | grep "HIP path"
echo "== vf_train"; DIINN_HIP_LIB=variants/libdiinn_vf_train.so python tools/train_time.py 16 48 4 --only-ours 2>&1 | grep "ours\|training forward"
echo "== vf_enc"; DIINN_HIP_LIB=variants/libdiinn_vf_enc.so python tools/enc_trunk_time.py 48 256 --only-hip 2>&1 | grep "HIP trunk"
done
