"""r04: decode time of the three bit-equal fp32 kernels (1 throughput, 2 32-pixel latency, 3 16-pixel latency) over HR sizes,
to calibrate the launch-size rule in decode_tile_impl.   python tools/f32_kernel_choice.py [LR] [B]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.synth as synth, diinn_amd.decoder as D, diinn_amd._native as N
lr = int(sys.argv[1]) if len(sys.argv) > 1 else 48
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
lib = N.load(); st = torch.cuda.current_stream().cuda_stream
feat = torch.randn(b, 64, lr, lr, device=dev)
ws = torch.empty(b * lr * lr * 1024, device=dev)
N.check(lib.diinn_precompute_P_ex(C.c_void_p(st), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(ws.data_ptr()), b, lr, lr, 0, lr, 0), "P")
for hu in [int(a) for a in (sys.argv[3:] or "96 120 144 168 192 224 256 288 320 384 448 512".split())]:
    out = torch.empty(b, 3, hu, hu, device=dev)
    wgs = b * ((hu + 15) // 16) * ((hu + 7) // 8)
    res = []
    for k in (0, 1, 1, 3):                      # (2 was the 32-pixel latency kernel: deleted in round 5)
        N.debug_set("DIINN_F32_KERNEL", k)
        def run():
            N.check(lib.diinn_decode_band_ex(C.c_void_p(st), C.c_void_p(ws.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(out.data_ptr()), b, lr, lr, hu, hu, 0, hu, 2, 0), "D")
        for _ in range(3): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort(); res.append(ts[len(ts) // 2] * 1e3)
    N.debug_set("DIINN_F32_KERNEL", 0)
    print(f"LR {lr} B {b} HR {hu:4d}  wgs(16x8) {wgs:5d}  auto {res[0]:7.1f} us | throughput {res[1]:7.1f}  coop32 {res[2]:7.1f}  coop16 {res[3]:7.1f}")
