"""Diagnostic: per-phase cycles of decode_bf16_coop_kernel from in-kernel s_memtime stamps (16 slots per wave).
   tools/build_variant.sh stamps -DDIINN_STAMPS
   DIINN_HIP_LIB=variants/libdiinn_stamps.so DIINN_BF16_KERNEL=4 python tools/stamp_report_coop.py [c5|c2]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diinn_amd._native as N, diinn_amd.decoder as D, diinn_amd.synth as synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
h, w, hu, wu = {"c5": (720, 1280, 2376, 4224), "c2": (256, 256, 1024, 1024)}[wl]
dev = torch.device("cuda:0")
lib = N.load(); raw = C.CDLL(N.LIB_PATH)
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
feat = torch.randn(1, 64, h, w, device=dev)
gx, gy = (wu + 15) // 16, (hu + 7) // 8
stamps = torch.zeros(gx * gy * 8 * 16, dtype=torch.int64, device=dev)   # room for the 8-wave kernel
for _ in range(3):
    D.decode_features(feat, packed, (hu, wu), compute="bf16_full")
torch.cuda.synchronize()
raw.diinn_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
D.decode_features(feat, packed, (hu, wu), compute="bf16_full")
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(-1, 16).astype(np.int64)
t = t[t[:, 0] > 0]
names = ["prologue (layer 0, staging, A loads)", "barrier", "layer 1 units", "barrier", "seed store + barrier",
         "layer 2 units", "barrier", "seed store + barrier", "layer 3 units", "barrier", "-", "head reduce + store"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
tot = t[:, 12] - t[:, 0]
if t[:, 14].max() > 0:
    for a, b, n in ((0, 13, "  prologue: setup (axis, addresses)"), (13, 14, "  prologue: A loads + staging loads/stores"), (0, 14, "  prologue: setup + loads + staging"), (14, 15, "  prologue: barrier"), (15, 1, "  prologue: layer 0 compute + qa writes")):
        d = t[:, b] - t[:, a]
        d = d[(t[:, a] > 0) & (t[:, b] > 0)]
        if len(d):
            print(f"{n:40s} median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}")
print(f"{wl}: waves {len(t)}  median wave lifetime {np.median(tot):.0f} cycles (MFMA floor 768*32 = 24576)")
for i, n in enumerate(names):
    a, b = idx[i], idx[i + 1]
    d = t[:, b] - t[:, a]
    print(f"  {n:38s} median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}  ({100*np.median(d)/np.median(tot):5.1f} %)")
