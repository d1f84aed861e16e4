"""Scratch timing of the decode path on one GPU (not the bench contract; see bench.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.synth as synth, diinn_amd.decoder as D, diinn_amd._native as N
import ctypes as C

def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    sin_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda:0")
    packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
    feat = torch.randn(1, 64, h, h, device=dev)
    ws = torch.empty(h * h * 1024, device=dev)
    out = torch.empty(1, 3, h * s, h * s, device=dev)
    lib = N.load()
    st = torch.cuda.current_stream().cuda_stream
    def P():
        N.check(lib.diinn_precompute_P(C.c_void_p(st), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()),
                                       C.c_void_p(ws.data_ptr()), 1, h, h, 0, h), "P")
    def Dk():
        N.check(lib.diinn_decode_band(C.c_void_p(st), C.c_void_p(ws.data_ptr()), C.c_void_p(packed.data_ptr()),
                                      C.c_void_p(out.data_ptr()), 1, h, h, h * s, h * s, 0, h * s, sin_mode), "D")
    for name, fn, flop in (("P", P, h * h * 1179648.0), ("decode", Dk, (h * s) ** 2 * 789504.0)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{name}: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s ({flop / ms / 1e9 / 157.3 * 100:.1f}% of fp32 MFMA peak)", flush=True)
    print(f"Mpix/s (decode only): {(h*s)**2 / ms / 1e3:.1f}")

if __name__ == "__main__":
    main()
