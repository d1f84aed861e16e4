"""Diagnostic: per-phase cycle shares of decode_kernel from in-kernel s_memtime stamps.
Needs a -DDIINN_STAMPS build:  tools/build_variant.sh stamps -DDIINN_STAMPS
   DIINN_HIP_LIB=variants/libdiinn_stamps.so python tools/stamp_report.py [LR] [scale]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diinn_amd._native as N, diinn_amd.decoder as D, diinn_amd.synth as synth

h = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
lib = N.load()
raw = C.CDLL(N.LIB_PATH)
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
feat = torch.randn(1, 64, h, h, device=dev)
hu = h * s
gx, gy = (hu + 15) // 16, (hu + 7) // 8
stamps = torch.zeros(gx * gy * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    D.decode_features(feat, packed, (hu, hu))
torch.cuda.synchronize()
raw.diinn_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
D.decode_features(feat, packed, (hu, hu))
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(-1, 8).astype(np.int64)
names = ["setup(axis,ptr)", "layer0", "layer1", "layer2", "layer3", "head+store"]
d = np.diff(t[:, :6], axis=1)
tot = t[:, 5] - t[:, 0]
print(f"waves {len(t)}  median cycles per wave-lifetime {np.median(tot):.0f}  (MFMA floor 6144*64 = 393216)")
for i, n in enumerate(names[1:]):
    print(f"  {n:16s} median {np.median(d[:, i]):9.0f} cycles  ({100*np.median(d[:, i])/np.median(tot):5.1f} %)   p90 {np.percentile(d[:, i], 90):9.0f}")
# start-time spread and gaps between consecutive WGs on a CU cannot be seen from here; report global span
rt = t[:, 7]
span_rt = (rt.max() - rt.min()) / 100e6
print(f"start-stamp span (s_memrealtime): {span_rt*1e3:.3f} ms; sum of wave lifetimes / (1024 waves) at 2.39 GHz = "
      f"{tot.sum()/1024/2.39e9*1e3:.3f} ms")
