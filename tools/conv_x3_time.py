"""One 3x3 layer of the trunk on the split-bf16 kernel (diinn_conv3x3_x3) against the fp32 Winograd kernel (diinn_conv_wino):
error against a float64 convolution and time.  usage: python tools/conv_x3_time.py [H W]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd._native as N
import diinn_amd.modules as M

lib = N.load()
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
torch.manual_seed(0)

def run(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for cin, relu, use_res, b in ((64, 1, 0, 1), (128, 1, 0, 1), (256, 1, 0, 1), (512, 1, 0, 1), (64, 0, 1, 2)):
    x = torch.randn(b, cin, H, W, device=dev).relu_() * 0.7
    w = (torch.rand(64, cin, 3, 3, device=dev) * 2 - 1) / (cin * 9) ** 0.5 * 1.7
    bias = torch.randn(64, device=dev) * 0.1
    res = torch.randn(b, 64, H, W, device=dev) if use_res else None
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)
    if relu: ref = ref.relu()
    if use_res: ref = ref + res.double()
    wx, wu = M.pack_conv_x3(w).to(dev), M.pack_conv_wino(w).to(dev)
    o3, ow = torch.empty(b, 64, H, W, device=dev), torch.empty(b, 64, H, W, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rp, rbs = (C.c_void_p(res.data_ptr()), 64 * H * W) if use_res else (None, 0)
    f3 = lambda: N.check(lib.diinn_conv3x3_x3(st, C.c_void_p(x.data_ptr()), cin * H * W, cin, C.c_void_p(wx.data_ptr()),
                         C.c_void_p(bias.data_ptr()), rp, rbs, C.c_void_p(o3.data_ptr()), 64 * H * W, relu, b, H, W), "x3")
    fw = lambda: N.check(lib.diinn_conv_wino(st, C.c_void_p(x.data_ptr()), cin * H * W, cin, C.c_void_p(wu.data_ptr()),
                         C.c_void_p(bias.data_ptr()), rp, rbs, C.c_void_p(ow.data_ptr()), 64 * H * W, relu, b, H, W), "wino")
    t3, tw = run(f3), run(fw)
    sc = float(ref.abs().max())
    print(f"{H}x{W} B={b} Cin={cin:4d} relu={relu} res={use_res}: split bf16 {t3:7.1f} us err {float((o3.double()-ref).abs().max())/sc:.2e} | "
          f"fp32 Winograd {tw:7.1f} us err {float((ow.double()-ref).abs().max())/sc:.2e}", flush=True)
