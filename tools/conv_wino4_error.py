import sys, ctypes as C, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import diinn_amd._native as N, diinn_amd.modules as M
from test_encoder_trunk import WINO_SHAPES
dev = torch.device("cuda:0"); lib = N.load()
gen = torch.Generator(device=dev).manual_seed(6)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
for (b, cin, h, w, relu, use_res) in WINO_SHAPES:
    x = torch.randn((b, cin, h, w), device=dev, generator=gen)
    wt = torch.randn((64, cin, 3, 3), device=dev, generator=gen) / (cin * 9) ** 0.5
    bias = torch.randn(64, device=dev, generator=gen)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    outs = {}
    for name, fn, pk in (("w2", lib.diinn_conv_wino, M.pack_conv_wino), ("w4", lib.diinn_conv_wino4, M.pack_conv_wino4)):
        out = torch.empty((b, 64, h, w), device=dev)
        assert fn(stream, ptr(x), cin * h * w, cin, ptr(pk(wt).to(dev)), ptr(bias), None, 0, ptr(out), 64 * h * w, 0, b, h, w) == 0
        outs[name] = float((out.double() - ref).abs().max())
    d32 = float((F.conv2d(x, wt, bias, padding=1).double() - ref).abs().max())
    print(f"{(b,cin,h,w)}: max|ref| {float(ref.abs().max()):.2f}  F(2,3) {outs['w2']:.2e}  F(4,3) {outs['w4']:.2e}  torch fp32 {d32:.2e}")
