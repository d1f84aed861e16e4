"""How much do the per-kernel HIP events of bench.py's timed loop cost per step?  c1 / c2, 0..4 events per step.
usage: python tools/event_overhead.py [c1|c2] [steps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd._native as N, diinn_amd.decoder as D, diinn_amd.synth as synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
h, w, hu, wu = {"c1": (48, 48, 96, 96), "c2": (256, 256, 1024, 1024)}[wl]
dev = torch.device("cuda:0")
lib = N.load()
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
feat = torch.randn(1, 64, h, w, device=dev)
ws = torch.empty(h * w * 1024, device=dev)
out = torch.empty(1, 3, hu, wu, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def step(nev, evs):
    for k in range(nev // 2):
        evs[k].record()
    N.check(lib.diinn_precompute_P_ex(st, C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(ws.data_ptr()), 1, h, w, 0, h, 0), "P")
    if nev >= 3:
        evs[2].record()
    N.check(lib.diinn_decode_band_ex(st, C.c_void_p(ws.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(out.data_ptr()), 1, h, w, hu, wu, 0, hu, 2, 0), "D")
    if nev >= 2:
        evs[3].record()


for nev in (0, 2, 3, 4, 0, 4):
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    for i in range(10):
        step(nev, evs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(nev, evs[i])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{wl}: {nev} events per step: {dt * 1e3:.4f} ms per step")
