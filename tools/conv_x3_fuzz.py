"""Seeded fuzz of the split-bf16 encoder kernels: N random 3x3 layers (every kernel form, ragged sizes, batches, ReLU, residual)
against a float64 convolution, and M whole trunks (random map sizes, both thresholds) against the fp32 trunk.
usage: python tools/conv_x3_fuzz.py [N] [M]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diinn_amd._native as N
import diinn_amd.modules as M

lib = N.load()
dev = torch.device("cuda:0")
n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_trunks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(77)
torch.manual_seed(77)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
worst = 0.0
for it in range(n_layers):
    b = int(rng.integers(1, 4)); cin = 16 * int(rng.integers(1, 37)); h = int(rng.integers(1, 90)); w = int(rng.integers(1, 90))
    relu = int(rng.integers(0, 2)); use_res = int(rng.integers(0, 2)); form = int(rng.integers(1, 4))
    N.debug_set("DIINN_ENC_X3_ROWS", form)
    x = torch.randn(b, cin, h, w, device=dev)
    wt = (torch.rand(64, cin, 3, 3, device=dev) * 2 - 1) / (cin * 9) ** 0.5 * 1.7
    bias = torch.randn(64, device=dev) * 0.1
    res = torch.randn(b, 64, h, w, device=dev) if use_res else None
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    if relu: ref = ref.relu()
    if use_res: ref = ref + res.double()
    wx = M.pack_conv_x3(wt).to(dev)
    out = torch.full((b, 64, h, w), float("nan"), device=dev)
    N.check(lib.diinn_conv3x3_x3(st, C.c_void_p(x.data_ptr()), cin * h * w, cin, C.c_void_p(wx.data_ptr()), C.c_void_p(bias.data_ptr()),
                                 C.c_void_p(res.data_ptr()) if use_res else None, 64 * h * w, C.c_void_p(out.data_ptr()), 64 * h * w,
                                 relu, b, h, w), "conv")
    torch.cuda.synchronize()
    err = float((out.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-9)
    worst = max(worst, err)
    assert err <= 2e-5 and bool(torch.isfinite(out).all()), (it, b, cin, h, w, relu, use_res, form, err)
print(f"{n_layers} layers: worst error {worst:.2e} of max|out| (bound 2e-5)")
N.debug_set("DIINN_ENC_X3_ROWS", 0)
net = M.DIINN(mode=3, init_q=False).to(dev).eval()
enc = net.encoder
worst_f = 0.0
with torch.no_grad():
    for it in range(n_trunks):
        b = int(rng.integers(1, 3)); h = int(rng.integers(8, 230)); w = int(rng.integers(8, 230))
        N.debug_set("DIINN_ENC_X3_MIN", 0 if it % 2 == 0 else 32768)
        N.debug_set("DIINN_ENC_X3_ROWS", [0, 4, 2, 3][it % 4])
        x = torch.rand(b, 3, h, w, device=dev)
        enc.hip_split_bf16 = False; f32 = enc(x)
        enc.hip_split_bf16 = True; f3 = enc(x)
        torch.cuda.synchronize()
        d = float((f3 - f32).abs().max()) / float(f32.abs().max())
        worst_f = max(worst_f, d)
        assert d <= 3e-5 and bool(torch.isfinite(f3).all()), (it, b, h, w, d)
print(f"{n_trunks} trunks: worst feature difference {worst_f:.2e} of max|feat| (bound 3e-5)")
