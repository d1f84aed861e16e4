"""precompute_P_kernel time vs number of LR rows (fixed width): separates fixed cost from per-row cost."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd._native as N, diinn_amd.decoder as D, diinn_amd.synth as synth
dev = torch.device("cuda:0")
lib = N.load()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
feat = torch.randn(1, 64, H, W, device=dev)
ws = torch.empty(H * W * 1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rows in [r for r in (4, 8, 16, 32, 64, 128, 256, 512, 1024) if r <= H]:
    def run():
        N.check(lib.diinn_precompute_P(C.c_void_p(st), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()),
                                       C.c_void_p(ws.data_ptr()), 1, H, W, 0, rows), "P")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    blocks = (W // 32) * (rows // 4)
    ms_split = 1
    while ms_split < 16 and blocks * ms_split < 1024: ms_split *= 2
    print(f"rows {rows:5d}  cell-blocks {blocks:5d} x msplit {ms_split:2d} = {blocks*ms_split:5d} WGs  {ms:.3f} ms  "
          f"{rows*W*1179648/ms/1e9:.1f} TFLOP/s ({rows*W*1179648/ms/1e9/157.3*100:.0f}%)")
