"""Scratch timing: fp32 vs bf16 decode kernels on one GPU."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.synth as synth, diinn_amd.decoder as D, diinn_amd._native as N
h = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
feat = torch.randn(1, 64, h, h, device=dev)
ws = torch.empty(h * h * 1024, device=dev)
out = torch.empty(1, 3, h * s, h * s, device=dev)
lib = N.load(); st = torch.cuda.current_stream().cuda_stream
N.check(lib.diinn_precompute_P(C.c_void_p(st), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(ws.data_ptr()), 1, h, h, 0, h), "P")
for name, comp in (("f32", 0), ("bf16", 1)):
    def run():
        N.check(lib.diinn_decode_band_ex(C.c_void_p(st), C.c_void_p(ws.data_ptr()), C.c_void_p(packed.data_ptr()),
                                         C.c_void_p(out.data_ptr()), 1, h, h, h * s, h * s, 0, h * s, 2, comp), "D")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = (h * s) ** 2 * 789504.0
    print(f"decode {name}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s  {(h*s)**2/ms/1e3:.1f} Mpix/s")
