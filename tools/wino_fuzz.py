"""A larger seeded fuzz of the Winograd kernels than the one in tests/: 250 random 3x3 convolutions (batch 1..3, 8..312
inputs, maps up to 139x179, ReLU / residual at random) against float64, 120 random maps and bands of the P kernel against
the direct kernel and band against full launch bit for bit.  Last run: 0 failures of 250 / 120."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.nn.functional as F
import diinn_amd._native as N, diinn_amd.modules as M, diinn_amd.decoder as D, diinn_amd.synth as synth
dev = torch.device("cuda:0"); lib = N.load()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
rng = np.random.default_rng(2024); gen = torch.Generator(device=dev).manual_seed(2024)
bad = 0
for it in range(250):
    b, cin = int(rng.integers(1, 4)), 8 * int(rng.integers(1, 40))
    h, w = int(rng.integers(1, 140)), int(rng.integers(1, 180))
    relu, use_res = int(rng.integers(2)), int(rng.integers(2))
    x = torch.randn((b, cin, h, w), device=dev, generator=gen)
    wt = torch.randn((64, cin, 3, 3), device=dev, generator=gen) / (cin * 9) ** 0.5
    bias = torch.randn(64, device=dev, generator=gen)
    res = torch.randn((b, 64, h, w), device=dev, generator=gen) if use_res else None
    out = torch.full((b, 64, h, w), float("nan"), device=dev)
    packed = M.pack_conv_wino(wt)
    st = lib.diinn_conv_wino(stream, ptr(x), cin * h * w, cin, ptr(packed), ptr(bias), ptr(res) if use_res else None, 64 * h * w, ptr(out), 64 * h * w, relu, b, h, w)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    if relu: ref = torch.relu(ref)
    if use_res: ref = ref + res.double()
    err = float((out.double() - ref).abs().max())
    if st != 0 or not err <= 2e-5 * max(1.0, float(ref.abs().max())):
        bad += 1; print("WINO BAD", b, cin, h, w, relu, use_res, st, err)
print("conv_wino fuzz: bad", bad, "of 250")
packed = D.pack_state_dict(synth.decoder_state_dict(5)).to(dev)
bad = 0
for it in range(120):
    b, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 200)), int(rng.integers(1, 150))
    feat = torch.randn((b, 64, h, w), device=dev, generator=gen)
    pw = torch.full((b, h, w, 1024), float("nan"), device=dev); pd = torch.full((b, h, w, 1024), float("nan"), device=dev)
    N.check(lib.diinn_precompute_P_ex(stream, ptr(feat), ptr(packed), ptr(pw), b, h, w, 0, h, N.COMPUTE_F32), "w")
    N.check(lib.diinn_precompute_P(stream, ptr(feat), ptr(packed), ptr(pd), b, h, w, 0, h), "d")
    scale = max(1.0, float(pd.abs().max()))
    ok = bool(torch.isfinite(pw).all()) and float((pw - pd).abs().max()) <= 1e-5 * scale
    r0 = int(rng.integers(0, h)); r1 = int(rng.integers(r0 + 1, h + 1)); f0, f1 = max(r0 - 1, 0), min(r1 + 1, h)
    fwin = feat[:, :, f0:f1].contiguous(); pwin = torch.full((b, r1 - r0, w, 1024), float("nan"), device=dev)
    N.check(lib.diinn_precompute_P_win(stream, ptr(fwin), f0, f1 - f0, ptr(packed), ptr(pwin), r0, r1 - r0, b, h, w, r0, r1, N.COMPUTE_F32), "win")
    ok = ok and bool(torch.equal(pwin, pw[:, r0:r1]))
    if not ok: bad += 1; print("P BAD", b, h, w, r0, r1)
print("P wino fuzz: bad", bad, "of 120")
