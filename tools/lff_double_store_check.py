"""Repeat the LFF configuration of the RDN trunk (1x1, 576 -> 64, residual, two destinations) on conv1x1_stream_kernel
and compare both destinations with a float64 convolution every time.  This is the check that exposed the 128-bit
buffer-store hazard described at st_b128 in csrc/diinn_device.h (second destination wrong in ~1 of 6 launches).
usage: lff_double_store_check.py"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
import diinn_amd._native as N, diinn_amd.modules as M
dev = torch.device("cuda:0"); lib = N.load()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
gen = torch.Generator(device=dev).manual_seed(0)
b, cin, h, w = 1, 576, 200, 180
hw = h * w
ws = torch.randn(2 * 576 * hw + 1024 * hw, device=dev, generator=gen)
cur, nxt, gff = ws[:576 * hw].view(1, 576, h, w), ws[576 * hw:2 * 576 * hw].view(1, 576, h, w), ws[2 * 576 * hw:].view(1, 1024, h, w)
wt = torch.randn((64, cin, 1, 1), device=dev, generator=gen) / cin ** 0.5
bias = torch.randn(64, device=dev, generator=gen)
packed = M.pack_conv_ksplit(wt).to(dev)
ref = (F.conv2d(cur.double(), wt.double(), bias.double()) + cur[:, :64].double()).float()
bad = 0
for it in range(300):
    d = it % 16
    nxt[:, :64].fill_(float("nan")); gff[:, 64 * d:64 * d + 64].fill_(float("nan"))
    st = lib.diinn_conv_ksplit(stream, ptr(cur), 576 * hw, cin, 1, ptr(packed), ptr(bias), ptr(cur), 576 * hw,
                               ptr(nxt), 576 * hw, ptr(gff[:, 64 * d:]), 1024 * hw, 0, b, h, w)
    e0 = float((nxt[:, :64] - ref).abs().max()); e1 = float((gff[:, 64 * d:64 * d + 64] - ref).abs().max())
    if not (e0 < 1e-4 and e1 < 1e-4):
        bad += 1
        if bad <= 3:
            dd = ((nxt[:, :64] - ref).abs() > 1e-4) | torch.isnan(nxt[:, :64])
            print(" iter", it, "e0", e0, "e1", e1, "bad elems", int(dd.sum()), dd.nonzero()[:8].tolist())
print("bad iterations", bad, "of 300")
