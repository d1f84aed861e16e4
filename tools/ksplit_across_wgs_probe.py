import ctypes as C, sys, torch
sys.path.insert(0, '/root/repo')
import diinn_amd.modules as M
from diinn_amd import _native
dev = torch.device("cuda:0"); lib = _native.load()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(b, cin, lr, taps=9, n=50):
    hw = lr * lr
    buf = torch.randn(b, cin + 64, lr, lr, device=dev) * 0.1
    k = 3 if taps == 9 else 1
    w = M.pack_conv_ksplit(torch.randn(64, cin, k, k) * 0.01).to(dev)
    bias = torch.zeros(64, device=dev); out = buf[:, cin:]
    def run():
        _native.check(lib.diinn_conv_ksplit(stream, C.c_void_p(buf.data_ptr()), (cin + 64) * hw, cin, taps, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, 0,
                                            C.c_void_p(out.data_ptr()), (cin + 64) * hw, None, 0, 1, b, lr, lr), "conv")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for lr in (48, 64):
    for cin in (64, 128, 256, 512):
        row = [f"{lr}x{lr} Cin {cin:3d}: whole {t(1, cin, lr):6.1f} us"]
        for s in (2, 4):
            if cin // s >= 64: row.append(f"K/{s} x B={s}: {t(s, cin // s, lr):6.1f}")
        print("  ".join(row), flush=True)
    print(f"{lr}x{lr} 1x1 Cin 576: {t(1, 576, lr, 1):6.1f} us   (K/3 x B=3: {t(3, 192, lr, 1):6.1f})")
