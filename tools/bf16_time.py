"""Scratch timing of the bf16 decode kernel alone (P precomputed once): python tools/bf16_time.py [c5|c2] [reps]
Select a kernel with DIINN_BF16_KERNEL, a variant library with DIINN_HIP_LIB."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.synth as synth, diinn_amd.decoder as D, diinn_amd._native as N
wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h, w, hu, wu = {"c5": (720, 1280, 2376, 4224), "c2": (256, 256, 1024, 1024), "c1": (48, 48, 96, 96)}[wl]
dev = torch.device("cuda:0")
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
feat = torch.randn(1, 64, h, w, device=dev)
ws = torch.empty(h * w * 1024, device=dev)
out = torch.empty(1, 3, hu, wu, device=dev)
lib = N.load(); st = torch.cuda.current_stream().cuda_stream
comp = int(os.environ.get("COMPUTE", "3"))
def runP():
    N.check(lib.diinn_precompute_P_ex(C.c_void_p(st), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(ws.data_ptr()), 1, h, w, 0, h, comp), "P")
def runD():
    N.check(lib.diinn_decode_band_ex(C.c_void_p(st), C.c_void_p(ws.data_ptr()), C.c_void_p(packed.data_ptr()),
                                     C.c_void_p(out.data_ptr()), 1, h, w, hu, wu, 0, hu, int(os.environ.get("SIN", "2")), comp), "D")
res = {}
for name, fn in (("P", runP), ("decode", runD)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    res[name] = (ts[0], ts[len(ts) // 2])
tag = os.environ.get("DIINN_HIP_LIB", "shipped").split("libdiinn_")[-1].replace(".so", "") + " k=" + os.environ.get("DIINN_BF16_KERNEL", "auto")
print(f"{wl} {tag:28s} P min/med {res['P'][0]:.3f}/{res['P'][1]:.3f} ms   decode min/med {res['decode'][0]:.3f}/{res['decode'][1]:.3f} ms")
