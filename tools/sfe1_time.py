import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
import diinn_amd._native as N
dev = torch.device("cuda:0"); lib = N.load()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
for lr in (48, 128, 256, 512):
    x = torch.randn(1, 3, lr, lr, device=dev); wt = torch.randn(64, 3, 3, 3, device=dev); b = torch.randn(64, device=dev)
    out = torch.empty(1, 64, lr, lr, device=dev)
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    a = t(lambda: lib.diinn_sfe1_forward(stream, ptr(x), 3, ptr(wt), ptr(b), ptr(out), 1, lr, lr))
    m = t(lambda: F.conv2d(x, wt, b, padding=1))
    print(lr, "hip %.1f us  miopen %.1f us" % (a, m))
