"""GPU parity: the HIP decode path (through the C ABI) against the golden
fixtures of the real reference and against the CPU oracle.

Tolerance (BASELINE.json north_star / SURVEY.md §8 d4): fp32 path
max|out - ref| <= 1e-4 * max(1, max|ref|); coordinate tables bit-exact."""
import numpy as np
import pytest
import torch

import diinn_amd.synth as synth
from conftest import golden_cases

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _tol(ref):
    return TOL * max(1.0, float(np.abs(ref).max()))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def _decode(sd, feat, size, dev, **kw):
    import diinn_amd.decoder as D
    packed = D.pack_state_dict(sd).to(dev)
    out = D.decode_features(torch.from_numpy(feat).to(dev), packed, size, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("sin_mode", [0, 1, 2])
def test_golden_fixtures(golden, dev, sin_mode):
    """Every reference output captured in tests/golden (diinn.py:163-173 run on the CPU)."""
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        got = _decode(sd, feat, (hu, wu), dev, sin_mode=sin_mode)
        ref = golden[f"out/{name}"]
        err = float(np.abs(got - ref).max())
        assert got.shape == ref.shape
        assert err <= _tol(ref), f"{name}: max err {err:.3e} > {_tol(ref):.3e} (sin_mode={sin_mode})"


# Regression-level bounds (VERDICT r03 "what's weak" 1): the contract above is 1e-4, the fp32 kernels sit ~3,000x below
# it at default-init weights, so a thousandfold loss of accuracy would stay green under it.  These hold the kernels
# NEAR the noise floor instead: 5e-7 absolute at default init (observed 2e-8 .. 6e-8: one wrong sine mode, a dropped
# term or a mis-rounded constant lands at 1e-6 or above), and on the x3 stress set 3x the reference's OWN fp32-vs-fp64
# distance, measured against float64 truth (ref64 = ref32 + d64, tests/golden/make_golden_r4.py).
REGRESSION_ABS = 5e-7


@pytest.mark.parametrize("sin_mode", [0, 1, 2])
def test_golden_fixtures_at_the_noise_floor(golden, golden_r4, dev, sin_mode):
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        got = _decode(sd, feat, (hu, wu), dev, sin_mode=sin_mode)
        ref32 = golden[f"out/{name}"]
        d64 = golden_r4[f"d64/{name}"].astype(np.float64)
        ref64 = ref32.astype(np.float64) + d64
        err32 = float(np.abs(got - ref32).max())
        err64 = float(np.abs(got.astype(np.float64) - ref64).max())
        if gain == 1.0:
            assert err32 <= REGRESSION_ABS and err64 <= REGRESSION_ABS, f"{name} sin_mode={sin_mode}: {err32:.3e} / {err64:.3e}"
        else:
            noise = float(np.abs(d64).max())                      # the reference's own distance from float64
            assert err64 <= 3.0 * noise, f"{name} sin_mode={sin_mode}: {err64:.3e} vs float64, reference itself {noise:.3e}"


@pytest.mark.parametrize("compute", ["f32", "bf16x3"])
@pytest.mark.parametrize("sin_mode", [0, 1, 2])
def test_siren_range_fixtures(golden_r4, dev, sin_mode, compute):
    """Reference outputs with trained-SIREN-range synthesis weights (Q.0 x 30, Q.1-3 x sqrt 6: layer-0 sine arguments of
    ~33 rad, ~67 with every tensor doubled on top; diinn.py:61-62,134): the regime where the hardware sine on
    revolutions and the pre-scaled weight image earn their keep.  fp32 and split bf16, all three sine modes, at the
    north_star bound -- and near the noise floor for fp32 (10x the reference's own fp32-vs-fp64 distance)."""
    from conftest import siren_cases
    for name, b, h, w, hu, wu, gain, qg in siren_cases(golden_r4):
        sd = synth.decoder_state_dict(123, gain, q_gain=qg)
        feat = synth.encoder_features(123, b, h, w)
        got = _decode(sd, feat, (hu, wu), dev, sin_mode=sin_mode, compute=compute)
        ref32 = golden_r4[f"out/{name}"]
        d64 = golden_r4[f"d64/{name}"].astype(np.float64)
        err = float(np.abs(got - ref32).max())
        assert err <= _tol(ref32), f"{name} {compute} sin_mode={sin_mode}: {err:.3e} > {_tol(ref32):.3e}"
        if compute == "f32":
            err64 = float(np.abs(got.astype(np.float64) - (ref32.astype(np.float64) + d64)).max())
            noise = float(np.abs(d64).max())
            assert err64 <= 10.0 * noise, f"{name} sin_mode={sin_mode}: {err64:.3e} vs float64, reference itself {noise:.3e}"


@pytest.mark.parametrize("compute,bound", [("bf16", 1e-2), ("bf16_full", 1.5e-2)])
def test_siren_range_fixtures_plain_bf16_informational(golden_r4, dev, compute, bound):
    """The plain bf16 modes on the SIREN-range fixtures: INFORMATIONAL, like the x3 stress set.  Their restated bounds
    (2e-3 / 3e-3 x max|ref|) are stated for default-init-range weights; with the synthesis weights of
    layers 1..3 scaled by sqrt 6 the bf16 operand rounding alone gives 3.5e-3 / 5e-3 (the oracle's emulation), 1.5e-2 / 2e-2
    with every tensor doubled on top.  What is asserted: the kernels track the emulation of their own roundings to a
    fraction of that distance, and stay inside a loose 1e-2 / 1.5e-2 (x4 for the doubled case).  Split bf16 holds the fp32
    bound on the same fixtures (test_siren_range_fixtures)."""
    import diinn_oracle as orc
    from conftest import siren_cases
    for name, b, h, w, hu, wu, gain, qg in siren_cases(golden_r4):
        sd = synth.decoder_state_dict(123, gain, q_gain=qg)
        feat = synth.encoder_features(123, b, h, w)
        got = _decode(sd, feat, (hu, wu), dev, compute=compute)
        ref = golden_r4[f"out/{name}"]
        scale = float(np.abs(ref).max())
        emu = orc.decode_hoisted_form(sd, feat, (hu, wu), bf16_operands=True, bf16_p=(compute == "bf16_full")).numpy()
        err_ref = float(np.abs(got - ref).max()) / scale
        err_emu = float(np.abs(got - emu).max()) / scale
        d_emu = float(np.abs(emu - ref).max()) / scale
        assert err_ref <= bound * (4.0 if gain > 1.0 else 1.0), (name, compute, err_ref)
        assert err_emu <= 0.6 * d_emu + 1e-4, (name, compute, err_emu, d_emu)


def test_device_axis_tables_bit_exact(golden, dev):
    """The coordinate/index code the decode kernel runs, against the reference's own tables
    (_make_pos_encoding diinn.py:94-110 + ATen nearest-exact)."""
    import ctypes as C
    import diinn_amd._native as N
    lib = N.load()
    for k in golden.files:
        if not k.startswith("idx/"):
            continue
        _, path, pair = k.split("/")
        n_in, n_out = map(int, pair.split("_"))
        idx = torch.empty(n_out, dtype=torch.int32, device=dev)
        rel = torch.empty(n_out, dtype=torch.float32, device=dev)
        st = lib.diinn_make_axis_tables_device(C.c_void_p(torch.cuda.current_stream().cuda_stream), n_in, n_out,
                                               int(path == "small"), C.c_void_p(idx.data_ptr()),
                                               C.c_void_p(rel.data_ptr()))
        N.check(st, "tables")
        torch.cuda.synchronize()
        assert np.array_equal(idx.cpu().numpy(), golden[k]), k
        assert np.array_equal(rel.cpu().numpy().view(np.uint32), golden[f"rel/{path}/{pair}"].view(np.uint32)), k


def test_oracle_fresh_seed_noninteger(dev):
    """A shape/seed with no fixture: HIP vs the CPU oracle (pinned to the reference by test_oracle_golden)."""
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(7)
    feat = synth.encoder_features(7, 1, 33, 47)
    size = (109, 155)   # x3.303 / x3.298
    ref = orc.decode_reference_form(sd, feat, size, 30000).numpy()
    got = _decode(sd, feat, size, dev)
    assert float(np.abs(got - ref).max()) <= _tol(ref)


def test_bands_equal_full(dev):
    """Row bands (the multi-GPU shard unit) reproduce the full decode bit-for-bit."""
    import diinn_amd.decoder as D
    sd = synth.decoder_state_dict(5)
    feat = torch.from_numpy(synth.encoder_features(5, 1, 40, 56)).to(dev)
    packed = D.pack_state_dict(sd).to(dev)
    size = (132, 185)
    full = D.decode_features(feat, packed, size)
    out = torch.zeros_like(full)
    for y0, y1 in [(0, 37), (37, 38), (38, 100), (100, 132)]:
        D.decode_features(feat, packed, size, out=out, rows=(y0, y1))
    torch.cuda.synchronize()
    assert torch.equal(full, out)


def test_full_size_config2_band_vs_oracle(dev):
    """BASELINE config 2 (256x256 LR x4): decode at full size on the GPU, check HR row bands
    against the oracle (the oracle on the whole image takes ~20 s; bands take ~1 s), plus
    size-independent properties: determinism and band/full equality."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(123)
    feat_np = synth.encoder_features(123, 1, 256, 256)
    feat = torch.from_numpy(feat_np).to(dev)
    packed = D.pack_state_dict(sd).to(dev)
    size = (1024, 1024)
    full = D.decode_features(feat, packed, size)
    again = D.decode_features(feat, packed, size)
    torch.cuda.synchronize()
    assert torch.equal(full, again)
    full_np = full.cpu().numpy()
    assert np.isfinite(full_np).all()
    for y0, y1 in [(0, 8), (508, 524), (1016, 1024)]:
        ref = orc.decode_reference_form(sd, feat_np, size, 30000, row_range=(y0, y1)).numpy()
        err = float(np.abs(full_np[:, :, y0:y1] - ref).max())
        assert err <= _tol(ref), f"rows {y0}:{y1} err {err:.3e}"
        assert err <= REGRESSION_ABS, f"rows {y0}:{y1} err {err:.3e}: off the noise floor (observed 3e-8)"


def test_full_size_config2_batch2_second_seed(dev):
    """Config 2's geometry with B = 2 and another weight / feature seed (the full-size tests were one seed, B = 1):
    random HR row bands of both images against the oracle at the noise floor, the two images decoded together bit-equal to
    each decoded alone."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(2024)
    feat_np = synth.encoder_features(2024, 2, 256, 256)
    feat = torch.from_numpy(feat_np).to(dev)
    packed = D.pack_state_dict(sd).to(dev)
    size = (1024, 1024)
    full = D.decode_features(feat, packed, size)
    single = [D.decode_features(feat[i:i + 1].contiguous(), packed, size) for i in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(full, torch.cat(single, 0))
    full_np = full.cpu().numpy()
    rng = np.random.default_rng(11)
    for _ in range(4):
        y0 = int(rng.integers(0, 1020))
        y1 = y0 + 4
        ref = orc.decode_reference_form(sd, feat_np, size, 30000, row_range=(y0, y1)).numpy()
        err = float(np.abs(full_np[:, :, y0:y1] - ref).max())
        assert err <= REGRESSION_ABS, f"rows {y0}:{y1} err {err:.3e}"


def test_module_interface_matches_reference_signature(dev):
    """ImplicitDecoder.forward(x, size, bsize) with list / torch.Size sizes and any bsize (diinn.py:163)."""
    import diinn_amd.decoder as D
    dec = D.ImplicitDecoder(mode=3, init_q=False)
    sd = synth.decoder_state_dict(123)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    dec = dec.to(dev).eval()
    x = torch.from_numpy(synth.encoder_features(123, 1, 48, 48)).to(dev)
    with torch.no_grad():
        a = dec(x, [96, 96])
        b = dec(x, torch.Size([96, 96]), 30000)
        c = dec(x, (96, 96), 7)      # reference would hang for bsize < H_up; ours ignores the knob
    assert torch.equal(a, b) and torch.equal(a, c)
    assert a.shape == (1, 3, 96, 96) and a.dtype == torch.float32 and a.is_contiguous()
    d = dec(x, [96, 96])             # grad enabled + bsize None: the reference builds a graph (sr_module.py:128); so do we
    assert d.requires_grad and d.grad_fn is not None
    assert float((d.detach() - a).abs().max()) <= 1e-6       # training forward kernel == inference kernel


def test_device_sine_accuracy(dev):
    """The synthesis activation (reference torch.sin, diinn.py:25-26) as the kernels evaluate it:
    max abs error vs float64 sin over the argument ranges that occur (default init: |x| <= 1.5;
    trained SIREN-style weights: tens).  Bounds are ~10x the observed errors."""
    import ctypes as C
    import diinn_amd._native as N
    lib = N.load()
    gen = torch.Generator().manual_seed(0)
    report = {}
    for span, bounds in ((4.0, (4e-7, 2e-6, 2e-6)), (100.0, (4e-7, 2e-5, 2e-6)), (1.0e4, (4e-7, 2e-3, 4e-6))):
        x = (torch.rand(1 << 20, generator=gen, dtype=torch.float64) * 2 - 1) * span
        x32 = x.to(torch.float32)
        ref = torch.sin(x32.to(torch.float64))
        xd = x32.to(dev)
        for mode, bound in zip((0, 1, 2), bounds):
            y = torch.empty_like(xd)
            N.check(lib.diinn_eval_sin_device(C.c_void_p(torch.cuda.current_stream().cuda_stream), mode,
                                              C.c_void_p(xd.data_ptr()), C.c_void_p(y.data_ptr()), xd.numel()), "sin")
            torch.cuda.synchronize()
            err = float((y.cpu().to(torch.float64) - ref).abs().max())
            report[(span, mode)] = err
            assert err <= bound, f"sin mode {mode} on [-{span},{span}]: max abs err {err:.3e} > {bound:.1e}"
    print("sine max abs error by (range, mode):", {k: f"{v:.2e}" for k, v in report.items()})


def test_config5_shape_noninteger_720p_bands_vs_oracle(dev):
    """BASELINE config 5 geometry (720x1280 LR -> 2376x4224 HR, x3.3) in fp32: full-size decode on
    the GPU, HR row bands against the oracle (the tables for these axes are pinned bit-exactly by
    the golden fixtures; the coordinates differ from exact arithmetic by 1.6e-4 here)."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(123)
    feat_np = synth.encoder_features(5, 1, 720, 1280)
    feat = torch.from_numpy(feat_np).to(dev)
    packed = D.pack_state_dict(sd).to(dev)
    size = (2376, 4224)
    out = D.decode_features(feat, packed, size)
    torch.cuda.synchronize()
    out_np = out.cpu().numpy()
    assert np.isfinite(out_np).all()
    for y0, y1 in [(0, 4), (1187, 1191), (2372, 2376)]:
        ref = orc.decode_reference_form(sd, feat_np, size, 30000, row_range=(y0, y1)).numpy()
        err = float(np.abs(out_np[:, :, y0:y1] - ref).max())
        assert err <= _tol(ref), f"rows {y0}:{y1} err {err:.3e}"


def test_x8_and_batch_bands_vs_oracle(dev):
    """x8 (BASELINE config 4's scale) on a 96x80 map with B=2, compared in full with the oracle."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(11)
    feat_np = synth.encoder_features(11, 2, 24, 20)
    size = (192, 160)
    ref = orc.decode_reference_form(sd, feat_np, size, 30000).numpy()
    got = _decode(sd, feat_np, size, dev)
    assert float(np.abs(got - ref).max()) <= _tol(ref)


def test_strided_input_and_stream(dev):
    """Non-contiguous x (the boundary makes it contiguous) and a non-default stream."""
    import diinn_amd.decoder as D
    sd = synth.decoder_state_dict(123)
    packed = D.pack_state_dict(sd).to(dev)
    base = torch.from_numpy(synth.encoder_features(123, 1, 48, 48)).to(dev)
    ref = D.decode_features(base, packed, (96, 96))
    strided = torch.empty(1, 64, 48, 96, device=dev)[..., ::2]
    strided.copy_(base)
    assert not strided.is_contiguous()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        got = D.decode_features(strided, packed, (96, 96))
    st.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(ref, got)


def test_bf16_path_restated_tolerance(golden, dev):
    """Optional bf16-operand path (BASELINE config 5).  Two checks per fixture:
    (1) against the fp32 reference with the restated tolerance 2e-3 * max|ref|;
    (2) against the oracle's bf16-operand emulation (same roundings, fp32 accumulate).  bf16
        rounding is discontinuous, so 1e-7 differences in an activation can flip a rounding (0.4 %
        of the value): the kernel and the emulation agree to a fraction of their common distance
        from the fp32 reference, not to fp32 rounding; the bound is the same restated tolerance."""
    import diinn_oracle as orc
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        got = _decode(sd, feat, (hu, wu), dev, compute="bf16")
        ref = golden[f"out/{name}"]
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max())
        # 2e-3 is stated for default-init weights; the x3 stress set triples every sine argument and
        # with it the effect of the 2^-9 operand rounding (observed 1.3e-2): bounded at 3e-2 there.
        rel = 2e-3 if gain == 1.0 else 3e-2
        assert err <= rel * scale + 1e-6, f"{name}: bf16 vs fp32 reference {err:.3e} (max|ref| {scale:.3e})"
        emu = orc.decode_hoisted_form(sd, feat, (hu, wu), bf16_operands=True).numpy()
        err2 = float(np.abs(got - emu).max())
        assert err2 <= rel * scale + 1e-6, f"{name}: bf16 kernel vs bf16 emulation {err2:.3e}"
        print(f"bf16 {name}: max|ref| {scale:.3f}  vs fp32 reference {err:.2e} ({err/scale:.1e} rel)  vs emulation {err2:.2e}")



def test_bf16_full_path(golden, dev):
    """DIINN_COMPUTE_BF16_FULL: bf16 operands in the hoisted 3x3 conv as well (precompute_P_bf16_kernel).
    (1) P itself against the oracle's bf16-operand convolution: products of bf16 values are exact in fp32,
        so only the summation order differs -> tight bound;
    (2) the decode against the fp32 reference with the tolerance restated for this mode, 3e-3 * max|ref| at
        default-init weights (emulated: 1.6e-3), 5e-2 on the x3 stress set (emulated: 2.3e-2);
    (3) against the emulation of the same roundings."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    import torch.nn.functional as F
    lib = N.load()
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        packed = D.pack_state_dict(sd).to(dev)
        f = torch.from_numpy(feat).to(dev)
        ws = torch.full((b * h * w * 1024,), float("nan"), device=dev)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(lib.diinn_precompute_P_ex(stream, C.c_void_p(f.data_ptr()), C.c_void_p(packed.data_ptr()),
                                          C.c_void_p(ws.data_ptr()), b, h, w, 0, h, N.COMPUTE_BF16_FULL), "P bf16")
        torch.cuda.synchronize()
        sw = orc.split_weights(sd)
        p_emu = F.conv2d(orc._bf16_round(torch.from_numpy(feat)), orc._bf16_round(sw["Wx"].view(1024, 64, 3, 3)), None, padding=1)
        p_emu = (p_emu + sw["bK"].view(1, -1, 1, 1)).permute(0, 2, 3, 1)
        p_got = ws.view(b, h, w, 1024).cpu()
        assert torch.isfinite(p_got).all()
        assert float((p_got - p_emu).abs().max()) <= 2e-5 * max(1.0, float(p_emu.abs().max())), name
        got = _decode(sd, feat, (hu, wu), dev, compute="bf16_full")
        ref = golden[f"out/{name}"]
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max())
        rel = 3e-3 if gain == 1.0 else 5e-2
        assert err <= rel * scale + 1e-6, f"{name}: bf16_full vs fp32 reference {err:.3e} (max|ref| {scale:.3e})"
        emu = orc.decode_hoisted_form(sd, feat, (hu, wu), bf16_operands=True, bf16_p=True).numpy()
        err2 = float(np.abs(got - emu).max())
        assert err2 <= rel * scale + 1e-6, f"{name}: bf16_full kernel vs emulation {err2:.3e}"
        print(f"bf16_full {name}: vs fp32 reference {err/scale:.1e} rel  vs emulation {err2/scale:.1e} rel")

def test_p_winograd_form(dev):
    """The hoisted 3x3 conv in Winograd F(2x2,3x3) form (precompute_P_wino_kernel: what the fp32 inference entry points
    run on maps of >= 32,768 cells) against (1) a float64 convolution and (2) the direct kernel (diinn_precompute_P),
    on odd maps, border / interior blocks and batch > 1; row bands through windows (odd first / last rows, so the
    first and last Winograd tile rows stick out of the band) are bit-identical to the same rows of the full launch."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    import torch.nn.functional as F
    lib = N.load()
    sd = synth.decoder_state_dict(123)
    packed = D.pack_state_dict(sd).to(dev)
    sw = orc.split_weights(sd)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())                    # noqa: E731
    for (b, h, w) in [(1, 192, 176), (2, 129, 131), (1, 181, 183), (1, 256, 130)]:
        assert b * h * w >= 32768
        feat = torch.from_numpy(synth.encoder_features(5, b, h, w)).to(dev)
        pw = torch.full((b, h, w, 1024), float("nan"), device=dev)
        pd = torch.full((b, h, w, 1024), float("nan"), device=dev)
        N.check(lib.diinn_precompute_P_ex(stream, ptr(feat), ptr(packed), ptr(pw), b, h, w, 0, h, N.COMPUTE_F32), "P wino")
        N.check(lib.diinn_precompute_P(stream, ptr(feat), ptr(packed), ptr(pd), b, h, w, 0, h), "P direct")
        ref = F.conv2d(feat.double(), sw["Wx"].view(1024, 64, 3, 3).double().to(dev), sw["bK"].reshape(-1).double().to(dev), padding=1).permute(0, 2, 3, 1)
        scale = max(1.0, float(ref.abs().max()))
        ew, ed = float((pw.double() - ref).abs().max()), float((pd.double() - ref).abs().max())
        assert torch.isfinite(pw).all() and ew <= 1e-5 * scale and ed <= 1e-5 * scale, (b, h, w, ew, ed)
        print(f"P {b}x{h}x{w}: Winograd vs float64 {ew:.2e}, direct vs float64 {ed:.2e} (max|P| {scale:.2f})")
        # bands through row windows
        for (r0, r1) in [(0, 7), (3, 10), (5, h), (h - 9, h - 2), (64, 65)]:
            f0, f1 = max(r0 - 1, 0), min(r1 + 1, h)
            fwin = feat[:, :, f0:f1].contiguous()
            pwin = torch.full((b, r1 - r0, w, 1024), float("nan"), device=dev)
            N.check(lib.diinn_precompute_P_win(stream, ptr(fwin), f0, f1 - f0, ptr(packed), ptr(pwin), r0, r1 - r0,
                                               b, h, w, r0, r1, N.COMPUTE_F32), "P win")
            assert torch.equal(pwin, pw[:, r0:r1]), (b, h, w, r0, r1)


def test_p_winograd_fuzz(dev):
    """Seeded random maps (1x1 .. 90x70, batch 1..3) and random row bands through the Winograd P kernel (every M-tile
    split the cost model picks on small maps) against the direct kernel and, band against full launch, bit for bit."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    lib = N.load()
    packed = D.pack_state_dict(synth.decoder_state_dict(5)).to(dev)
    rng = np.random.default_rng(3)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())                    # noqa: E731
    for _ in range(25):
        b, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 91)), int(rng.integers(1, 71))
        feat = torch.from_numpy(synth.encoder_features(int(rng.integers(100)), b, h, w)).to(dev)
        pw = torch.full((b, h, w, 1024), float("nan"), device=dev)
        pd = torch.full((b, h, w, 1024), float("nan"), device=dev)
        N.check(lib.diinn_precompute_P_ex(stream, ptr(feat), ptr(packed), ptr(pw), b, h, w, 0, h, N.COMPUTE_F32), "P wino")
        N.check(lib.diinn_precompute_P(stream, ptr(feat), ptr(packed), ptr(pd), b, h, w, 0, h), "P direct")
        scale = max(1.0, float(pd.abs().max()))
        assert torch.isfinite(pw).all() and float((pw - pd).abs().max()) <= 1e-5 * scale, (b, h, w)
        r0 = int(rng.integers(0, h))
        r1 = int(rng.integers(r0 + 1, h + 1))
        f0, f1 = max(r0 - 1, 0), min(r1 + 1, h)
        fwin = feat[:, :, f0:f1].contiguous()
        pwin = torch.full((b, r1 - r0, w, 1024), float("nan"), device=dev)
        N.check(lib.diinn_precompute_P_win(stream, ptr(fwin), f0, f1 - f0, ptr(packed), ptr(pwin), r0, r1 - r0,
                                           b, h, w, r0, r1, N.COMPUTE_F32), "P win")
        assert torch.equal(pwin, pw[:, r0:r1]), (b, h, w, r0, r1)


def test_random_shapes_vs_oracle(dev):
    """Fuzz: random LR/HR shapes, batch sizes and scales (up- and down-scaling, tile edges that do
    not divide the 16x8 workgroup block or the 4x32 cell block) against the oracle."""
    import diinn_oracle as orc
    rng = np.random.default_rng(7)
    sd = synth.decoder_state_dict(31)
    for case in range(12):
        b = int(rng.integers(1, 3))
        h, w = int(rng.integers(1, 45)), int(rng.integers(1, 70))
        hu, wu = int(rng.integers(1, 150)), int(rng.integers(1, 200))
        feat = synth.encoder_features(100 + case, b, h, w)
        ref = orc.decode_reference_form(sd, feat, (hu, wu), 30000).numpy()
        got = _decode(sd, feat, (hu, wu), dev)
        err = float(np.abs(got - ref).max())
        assert err <= _tol(ref), f"case {case}: B{b} {h}x{w} -> {hu}x{wu}: err {err:.3e}"


def test_random_geometries_with_bands_vs_oracle(dev):
    """Forty seeded random geometries for the DEFAULT fp32 path (what the optional split-bf16 mode already had): up- and
    down-scaling, non-integer ratios, batches, launches on both sides of the latency / throughput kernel switch (192
    workgroups); the whole image against the oracle at the noise floor, a random row band bit-equal to the same rows,
    rows outside the band untouched."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    rng = np.random.default_rng(4040)
    sd = synth.decoder_state_dict(43)
    packed = D.pack_state_dict(sd).to(dev)
    small = large = 0
    for it in range(40):
        b = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        sy, sx = rng.uniform(0.6, 7.0), rng.uniform(0.6, 7.0)
        hu, wu = max(1, int(h * sy)), max(1, int(w * sx))
        if b * hu * wu > 120_000:                                 # the oracle decodes the whole image: keep it to seconds
            hu, wu = min(hu, 200), min(wu, 200)
        wgs = b * ((wu + 15) // 16) * ((hu + 7) // 8)
        small += wgs <= 192
        large += wgs > 192
        feat_np = synth.encoder_features(300 + it, b, h, w)
        feat = torch.from_numpy(feat_np).to(dev)
        full = D.decode_features(feat, packed, (hu, wu))
        y0 = int(rng.integers(0, hu))
        y1 = int(rng.integers(y0 + 1, hu + 1))
        band = torch.full_like(full, float("nan"))
        D.decode_features(feat, packed, (hu, wu), out=band, rows=(y0, y1))
        torch.cuda.synchronize()
        ref = orc.decode_reference_form(sd, feat_np, (hu, wu), 30000).numpy()
        err = float(np.abs(full.cpu().numpy() - ref).max())
        assert err <= REGRESSION_ABS, f"case {it}: B{b} {h}x{w} -> {hu}x{wu}: err {err:.3e}"
        assert torch.equal(band[:, :, y0:y1], full[:, :, y0:y1]), (b, h, w, hu, wu, y0, y1)
        assert bool(torch.isnan(band[:, :, :y0]).all()) and bool(torch.isnan(band[:, :, y1:]).all())
    assert small >= 5 and large >= 5, (small, large)


@pytest.mark.parametrize("compute", ["f32", "bf16x3", "bf16", "bf16_full"])
def test_tiles_with_column_ranges_and_strides(dev, knobs, compute):
    """diinn_decode_tile_win (ABI v7; reference analogue: batched_step's column strips,
    diinn.py:149-160): (i) a 3 x 3 tiling with ragged cuts, each tile written straight into its window of one
    canvas through strides, stitches BIT-EXACTLY into the whole-image decode; (ii) a tile decoded into a strided view of
    a larger canvas leaves every other element of the canvas untouched; (iii) the same into a compact crop buffer.
    Every arithmetic mode; for the bf16 modes on a geometry large enough for the cooperative kernels, whose variant is
    chosen from the full image, never the tile."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    lib = N.load()
    big = compute in ("bf16", "bf16_full")
    b, h, w, hu, wu = (1, 120, 150, 396, 495) if big else (2, 40, 56, 132, 185)     # x3.3 both ways
    sd = synth.decoder_state_dict(19)
    packed = D.pack_state_dict(sd).to(dev)
    feat = torch.from_numpy(synth.encoder_features(19, b, h, w)).to(dev)
    full = D.decode_features(feat, packed, (hu, wu), compute=compute)
    pwin = torch.empty(b * h * w * 1024, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N.check(lib.diinn_precompute_P_win(stream, C.c_void_p(feat.data_ptr()), 0, h, C.c_void_p(packed.data_ptr()),
                                       C.c_void_p(pwin.data_ptr()), 0, h, b, h, w, 0, h, N.COMPUTE[compute]), "P")
    ycuts = [0, 37, 38 + 45, hu]
    xcuts = [0, 61, 61 + 3, wu]                                   # a 3-pixel-wide strip in the middle
    canvas = torch.full((b, 3, hu, wu), float("nan"), device=dev)
    for ya, yb in zip(ycuts[:-1], ycuts[1:]):
        for xa, xb in zip(xcuts[:-1], xcuts[1:]):
            D.decode_tile(pwin, 0, (b, h, w), packed, (hu, wu), (ya, yb), (xa, xb), canvas[:, :, ya:yb, xa:xb], compute=compute)
    torch.cuda.synchronize()
    assert torch.equal(canvas, full), compute
    # (ii) a larger canvas with a margin: only the tile's pixels change
    ya, yb, xa, xb = 50, 101, 23, 160
    wide = torch.full((b, 3, hu + 7, wu + 11), -7.0, device=dev)
    D.decode_tile(pwin, 0, (b, h, w), packed, (hu, wu), (ya, yb), (xa, xb), wide[:, :, 3 + ya:3 + yb, 5 + xa:5 + xb], compute=compute)
    torch.cuda.synchronize()
    assert torch.equal(wide[:, :, 3 + ya:3 + yb, 5 + xa:5 + xb], full[:, :, ya:yb, xa:xb])
    mask = torch.ones_like(wide, dtype=torch.bool)
    mask[:, :, 3 + ya:3 + yb, 5 + xa:5 + xb] = False
    assert bool((wide[mask] == -7.0).all())
    # (iii) a compact crop, and the band-sized P window of just the rows the tile reads
    (_, _), (r0, rn) = D.window_rows(h, hu, wu, ya, yb)
    pband = pwin.view(b, h, w, 1024)[:, r0:r0 + rn].contiguous().view(-1)
    crop = torch.empty((b, 3, yb - ya, xb - xa), device=dev)
    D.decode_tile(pband, r0, (b, h, w), packed, (hu, wu), (ya, yb), (xa, xb), crop, compute=compute)
    torch.cuda.synchronize()
    assert torch.equal(crop, full[:, :, ya:yb, xa:xb])
    if compute == "f32":                                          # the latency kernel takes small tiles: forced both ways
        for force in (1, 3):
            knobs("DIINN_F32_KERNEL", force)
            crop.fill_(0)
            D.decode_tile(pband, r0, (b, h, w), packed, (hu, wu), (ya, yb), (xa, xb), crop)
            torch.cuda.synchronize()
            assert torch.equal(crop, full[:, :, ya:yb, xa:xb]), force


def test_invalid_sizes_raise(dev):
    import diinn_amd.decoder as D
    import diinn_amd._native as N
    packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
    x = torch.zeros(1, 64, 4, 4, device=dev)
    with pytest.raises(ValueError):
        D.decode_features(x, packed, (8, 8, 8))              # reference: unpack error for len(size) != 2
    with pytest.raises(N.DiinnNativeError):
        D.decode_features(x, packed, (0, 8))
    with pytest.raises(ValueError):
        D.decode_features(torch.zeros(1, 32, 4, 4, device=dev), packed, (8, 8))


def test_against_plain_c_oracle(dev):
    """HIP path vs the independent plain-C restatement (oracle/diinn_oracle_c.c), a shape with no fixture."""
    import diinn_oracle_c as oc
    sd = synth.decoder_state_dict(77)
    feat = synth.encoder_features(77, 1, 9, 13)
    size = (31, 40)
    ref = oc.decode(sd, feat, size)
    got = _decode(sd, feat, size, dev)
    assert float(np.abs(got - ref).max()) <= _tol(ref)


@pytest.mark.parametrize("mode", [1, 2])
def test_modes_1_and_2(golden, dev, mode):
    """Decoder modes 1 and 2 (diinn.py:116-131): per-cell modulation chain + synthesis-only decode
    kernel, against the reference's outputs and, on another shape, the oracle."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(123, mode=mode)
    dec = D.ImplicitDecoder(mode=mode, init_q=False)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    dec = dec.to(dev).eval()
    feat = synth.encoder_features(123, 1, 24, 20)
    with torch.no_grad():
        got = dec(torch.from_numpy(feat).to(dev), (79, 66), 30000).cpu().numpy()
    ref = golden[f"mode{mode}/out_24x20_79x66"]
    assert float(np.abs(got - ref).max()) <= _tol(ref)
    feat2 = synth.encoder_features(5, 2, 17, 33)
    with torch.no_grad():
        got2 = dec(torch.from_numpy(feat2).to(dev), (40, 100), 30000).cpu().numpy()
    ref2 = orc.decode_reference_form(sd, feat2, (40, 100), 30000, mode=mode).numpy()
    assert float(np.abs(got2 - ref2).max()) <= _tol(ref2)


def test_latency_kernel_is_bit_identical_to_throughput_kernel(golden, dev, knobs):
    """Small and partly filled launches take decode_coop16_kernel (four waves share one 16-pixel tile on
    v_mfma_f32_16x16x4_f32: four products per instruction, added in k order).  Per output channel it performs
    decode_kernel's arithmetic in decode_kernel's order, so the two must agree bit for bit on every fixture, on
    row bands and on batches; the fixtures' reference outputs bound both.  diinn_decode_kernel_info reports the forced choice."""
    import ctypes as C
    import diinn_amd.decoder as D
    import diinn_amd._native as N
    cases = list(golden_cases(golden))
    info = (C.c_int * 4)()
    for name, b, h, w, hu, wu, gain in cases:
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        outs = {}
        for k in ("1", "3"):                       # throughput, 16-pixel latency
            knobs("DIINN_F32_KERNEL", int(k))
            outs[k] = _decode(sd, feat, (hu, wu), dev)
            assert N.load().diinn_decode_kernel_info(b, hu, wu, 0, hu, 0, wu, N.COMPUTE_F32, info) == 0 and info[0] == int(k)
        assert np.array_equal(outs["1"], outs["3"]), name
        ref = golden[f"out/{name}"]
        assert float(np.abs(outs["3"] - ref).max()) <= _tol(ref), name
    knobs("DIINN_F32_KERNEL", 1)
    sd = synth.decoder_state_dict(5)
    feat = torch.from_numpy(synth.encoder_features(5, 2, 19, 23)).to(dev)
    packed = D.pack_state_dict(sd).to(dev)
    full = D.decode_features(feat, packed, (61, 70))
    for force in (3,):
        knobs("DIINN_F32_KERNEL", force)
        out = torch.zeros_like(full)
        for y0, y1 in [(0, 17), (17, 18), (18, 61)]:
            D.decode_features(feat, packed, (61, 70), out=out, rows=(y0, y1))
        torch.cuda.synchronize()
        assert torch.equal(full, out), force


def test_bf16_kernel_variants_agree(golden, dev, knobs):
    """The bf16 decode kernels (one tile per wave, two tiles per wave, cooperative with 8 waves: one block per workgroup and persistent) are
    the same arithmetic (same products, same k-order; only the head's summation order differs) laid out differently:
    their outputs agree to a small fraction of the bf16 error (the cooperative kernels evaluate layer 0's sine on
    revolutions as well, and a 1e-7 difference in an activation can flip a bf16 rounding), and each meets the restated
    tolerance.  DIINN_BF16_KERNEL selects the kernel;
    the cooperative ones need a block's LR footprint to fit their seed slab (scales from about x3), otherwise the
    launch falls back by itself."""
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        if gain != 1.0:
            continue
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        ref = golden[f"out/{name}"]
        scale = float(np.abs(ref).max())
        outs = {}
        for k in ("1", "2", "8", "9"):                             # 9: the 8-wave kernel with persistent workgroups
            knobs("DIINN_BF16_KERNEL", int(k))
            outs[k] = _decode(sd, feat, (hu, wu), dev, compute="bf16")
            assert float(np.abs(outs[k] - ref).max()) <= 2e-3 * scale + 1e-6, (name, k)
        for k in ("2", "8", "9"):
            assert float(np.abs(outs[k] - outs["1"]).max()) <= 5e-4 * scale, (name, k)
        # the persistent form does the one-block form's arithmetic; only the head's 16 partial sums meet in another order
        assert float(np.abs(outs["9"] - outs["8"]).max()) <= 2e-6 * scale, name


@pytest.mark.parametrize("compute", ["bf16", "bf16_full"])
def test_bf16_persistent_kernel_blocks_bands_and_batches(dev, knobs, compute):
    """decode_bf16_coop8p_kernel walks super-tiles of 8 x 4 blocks with 256 persistent workgroups and prepares the next
    block inside the current one's last layer: ragged right / bottom edges, more super-tiles than one per XCD, batch > 1,
    and row bands whose first row is not a block boundary must all reproduce the one-block-per-workgroup kernel (to the
    head's summation order) and meet the restated bound against the oracle."""
    import diinn_oracle as orc
    import diinn_amd.decoder as D
    tol = {"bf16": 2e-3, "bf16_full": 3e-3}[compute]
    for (b, h, w, hu, wu, seed) in [(1, 40, 56, 132, 185, 3), (2, 33, 47, 109, 155, 4), (1, 60, 100, 333, 530, 5),
                                    (3, 9, 11, 36, 40, 6)]:
        sd = synth.decoder_state_dict(seed)
        feat_np = synth.encoder_features(seed, b, h, w)
        ref = orc.decode_reference_form(sd, feat_np, (hu, wu), 30000).numpy()
        scale = float(np.abs(ref).max())
        knobs("DIINN_BF16_KERNEL", 8)
        one = _decode(sd, feat_np, (hu, wu), dev, compute=compute)
        knobs("DIINN_BF16_KERNEL", 9)
        per = _decode(sd, feat_np, (hu, wu), dev, compute=compute)
        assert float(np.abs(per - ref).max()) <= tol * scale, (b, h, w, hu, wu)
        assert float(np.abs(per - one).max()) <= 2e-6 * scale, (b, h, w, hu, wu)
        # bands: rows [0, 13), [13, hu - 5), [hu - 5, hu) stitched into one image equal the whole decode bit for bit
        packed = D.pack_state_dict(sd).to(dev)
        feat = torch.from_numpy(feat_np).to(dev)
        full = D.decode_features(feat, packed, (hu, wu), compute=compute)
        out = torch.zeros_like(full)
        for y0, y1 in [(0, 13), (13, hu - 5), (hu - 5, hu)]:
            D.decode_features(feat, packed, (hu, wu), out=out, rows=(y0, y1), compute=compute)
        torch.cuda.synchronize()
        assert torch.equal(full, out), (b, h, w, hu, wu)


# ---------------------------------------------------------------------------------------------------------------------
# split bf16 (DIINN_COMPUTE_BF16X3, decode_bf16x3_kernel): the bf16 matrix cores at the fp32 path's tolerance


def test_bf16x3_meets_the_fp32_tolerance_on_every_fixture(golden, dev):
    """hi/lo bf16 operands, three products per term: the kernel is held to the SAME bound as the fp32 kernels,
    1e-4 x max(1, |ref|) against the reference's own outputs (tests/golden), stress weights included, and to the
    oracle's emulation of its roundings (decode_hoisted_form(bf16x3=True))."""
    import diinn_oracle as orc
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        ref = golden[f"out/{name}"]
        for sin_mode in (0, 1, 2):
            got = _decode(sd, feat, (hu, wu), dev, compute="bf16x3", sin_mode=sin_mode)
            err = float(np.abs(got - ref).max())
            assert got.shape == ref.shape and err <= _tol(ref), f"{name}/{sin_mode}: bf16x3 {err:.3e} > {_tol(ref):.3e}"
        emu = orc.decode_hoisted_form(sd, feat, (hu, wu), bf16x3=True).numpy()
        err2 = float(np.abs(got - emu).max())
        assert err2 <= _tol(ref), f"{name}: bf16x3 kernel vs emulation {err2:.3e}"
        f32 = _decode(sd, feat, (hu, wu), dev)
        print(f"bf16x3 {name}: max|ref| {float(np.abs(ref).max()):.3f}  vs reference {err:.2e}  vs emulation {err2:.2e}  "
              f"(fp32 kernel vs reference {float(np.abs(f32 - ref).max()):.2e}; bound {_tol(ref):.1e})")


def test_bf16x3_with_the_split_bf16_hoisted_conv_on_every_fixture(golden, dev, knobs):
    """On maps of >= 32,768 cells the mode also evaluates the hoisted 3x3 conv P in split bf16 (precompute_P_x3_kernel).
    Forced on for the fixtures (DIINN_P_X3_MIN = 0): the SAME fp32 bound against the reference's outputs, stress weights
    included, and the oracle's emulation of both roundings; the P image itself within 2e-5 of max|P| of the fp32 kernel's,
    full launches and row bands (nothing written outside the band)."""
    import ctypes as C
    import diinn_oracle as orc
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    knobs("DIINN_P_X3_MIN", 0)
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        ref = golden[f"out/{name}"]
        got = _decode(sd, feat, (hu, wu), dev, compute="bf16x3")
        err = float(np.abs(got - ref).max())
        assert err <= _tol(ref), f"{name}: bf16x3 + split-bf16 P {err:.3e} > {_tol(ref):.3e}"
        emu = orc.decode_hoisted_form(sd, feat, (hu, wu), bf16x3=True, bf16x3_p=True).numpy()
        assert float(np.abs(got - emu).max()) <= _tol(ref), name
        print(f"bf16x3 + split-bf16 P {name}: vs reference {err:.2e} (bound {_tol(ref):.1e})")
    lib = N.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    packed = D.pack_state_dict(synth.decoder_state_dict(5)).to(dev)
    for (b, h, w, r0, r1) in [(2, 45, 67, 0, 45), (1, 100, 33, 13, 77), (3, 8, 32, 2, 7), (1, 1, 1, 0, 1), (1, 64, 96, 0, 64)]:
        feat = torch.from_numpy(synth.encoder_features(9, b, h, w)).to(dev)
        outs = []
        for comp in (N.COMPUTE_F32, N.COMPUTE_BF16X3):
            P = torch.full((b, h, w, 1024), float("nan"), device=dev)
            N.check(lib.diinn_precompute_P_ex(st, C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()),
                                              C.c_void_p(P.data_ptr()), b, h, w, r0, r1, comp), "P")
            outs.append(P)
        torch.cuda.synchronize()
        a, x = outs
        assert float((x[:, r0:r1] - a[:, r0:r1]).abs().max()) <= 2e-5 * float(a[:, r0:r1].abs().max()), (b, h, w)
        assert bool(torch.isnan(x[:, :r0]).all()) and bool(torch.isnan(x[:, r1:]).all())
        algo = C.c_int(-1)
        N.check(lib.diinn_p_launch_info(b, h, w, r0, r1, N.COMPUTE_BF16X3, C.byref(algo)), "info")
        assert algo.value == N.P_ALGO_DIRECT_BF16X3


def test_bf16x3_ragged_shapes_batches_and_bands(dev):
    """Fresh seeds, non-integer scales, batch > 1, ragged edges; row bands stitch bit-exactly into the whole image."""
    import diinn_oracle as orc
    import diinn_amd.decoder as D
    for (b, h, w, hu, wu, seed) in [(1, 40, 56, 132, 185, 3), (2, 33, 47, 109, 155, 4), (3, 9, 11, 36, 40, 6), (1, 5, 3, 9, 4, 8)]:
        sd = synth.decoder_state_dict(seed)
        feat_np = synth.encoder_features(seed, b, h, w)
        ref = orc.decode_reference_form(sd, feat_np, (hu, wu), 30000).numpy()
        packed = D.pack_state_dict(sd).to(dev)
        feat = torch.from_numpy(feat_np).to(dev)
        full = D.decode_features(feat, packed, (hu, wu), compute="bf16x3")
        assert float(np.abs(full.cpu().numpy() - ref).max()) <= _tol(ref), (b, h, w, hu, wu)
        out = torch.zeros_like(full)
        cuts = sorted({0, min(13, hu), max(hu - 5, 0), hu})
        for y0, y1 in zip(cuts[:-1], cuts[1:]):
            D.decode_features(feat, packed, (hu, wu), out=out, rows=(y0, y1), compute="bf16x3")
        torch.cuda.synchronize()
        assert torch.equal(full, out), (b, h, w, hu, wu)


def test_bf16x3_persistent_kernel_is_bit_identical(dev, knobs):
    """decode_bf16x3h_kernel (persistent workgroups, hi weight pieces shared through an LDS ring with a barrier per stage,
    the next block's layer 0 evaluated inside the current block's last layer) does decode_bf16x3_kernel's arithmetic
    pixel by pixel: equal bit for bit on ragged edges, batches, fewer blocks than workgroups, many blocks per workgroup,
    and row bands that do not start on a block boundary."""
    import diinn_oracle as orc
    import diinn_amd.decoder as D
    for (b, h, w, hu, wu, seed) in [(1, 60, 100, 333, 530, 5), (2, 33, 47, 109, 155, 4), (3, 9, 11, 36, 40, 6),
                                    (1, 5, 3, 9, 4, 8), (1, 64, 64, 256, 256, 9)]:
        sd = synth.decoder_state_dict(seed)
        feat_np = synth.encoder_features(seed, b, h, w)
        packed = D.pack_state_dict(sd).to(dev)
        feat = torch.from_numpy(feat_np).to(dev)
        outs = {}
        for k in (1, 2):                                          # one block per workgroup, persistent
            knobs("DIINN_X3_KERNEL", k)
            for sin_mode in (0, 1, 2):
                outs[k, sin_mode] = D.decode_features(feat, packed, (hu, wu), compute="bf16x3", sin_mode=sin_mode)
        torch.cuda.synchronize()
        for sin_mode in (0, 1, 2):
            assert torch.equal(outs[1, sin_mode], outs[2, sin_mode]), (b, h, w, hu, wu, sin_mode)
        ref = orc.decode_reference_form(sd, feat_np, (hu, wu), 30000).numpy()
        assert float(np.abs(outs[2, 2].cpu().numpy() - ref).max()) <= _tol(ref), (b, h, w, hu, wu)
        knobs("DIINN_X3_KERNEL", 2)
        out = torch.zeros_like(outs[2, 2])
        cuts = sorted({0, min(13, hu), max(hu - 5, 0), hu})
        for y0, y1 in zip(cuts[:-1], cuts[1:]):
            D.decode_features(feat, packed, (hu, wu), out=out, rows=(y0, y1), compute="bf16x3")
        torch.cuda.synchronize()
        assert torch.equal(outs[2, 2], out), (b, h, w, hu, wu)


def test_bf16x3_random_geometries_track_the_fp32_kernels(dev):
    """Forty seeded random geometries -- up- and down-scaling, non-integer ratios, batches, sizes on both sides of the
    one-block / persistent switch (512 blocks), a random row band each -- decoded in split bf16 and in fp32: the two
    agree to 2e-6 at default-init weights (both are within 1e-7 of the reference there), every value finite, the band
    bit-equal to the same rows of the whole image."""
    import diinn_amd.decoder as D
    rng = np.random.default_rng(20260)
    sd = synth.decoder_state_dict(41)
    packed = D.pack_state_dict(sd).to(dev)
    seen_persistent = seen_one_block = 0
    for it in range(40):
        b = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        sy, sx = rng.uniform(0.6, 9.0), rng.uniform(0.6, 9.0)
        hu, wu = max(1, int(h * sy)), max(1, int(w * sx))
        if b * hu * wu > 1_500_000:
            hu, wu = min(hu, 600), min(wu, 600)
        blocks = b * ((wu + 15) // 16) * ((hu + 7) // 8)
        seen_persistent += blocks >= 512
        seen_one_block += blocks < 512
        feat = torch.from_numpy(synth.encoder_features(100 + it, b, h, w)).to(dev)
        x3 = D.decode_features(feat, packed, (hu, wu), compute="bf16x3")
        f32 = D.decode_features(feat, packed, (hu, wu))
        y0 = int(rng.integers(0, hu))
        y1 = int(rng.integers(y0 + 1, hu + 1))
        band = torch.full_like(x3, float("nan"))
        D.decode_features(feat, packed, (hu, wu), out=band, rows=(y0, y1), compute="bf16x3")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(x3).all()), (b, h, w, hu, wu)
        assert float((x3 - f32).abs().max()) <= 2e-6, (b, h, w, hu, wu, float((x3 - f32).abs().max()))
        assert torch.equal(band[:, :, y0:y1], x3[:, :, y0:y1]), (b, h, w, hu, wu, y0, y1)
        assert bool(torch.isnan(band[:, :, :y0]).all()) and bool(torch.isnan(band[:, :, y1:]).all())   # rows outside untouched
    assert seen_persistent >= 5 and seen_one_block >= 5


def test_bf16x3_full_size_config2_band_vs_oracle(dev):
    """BASELINE config 2 at full size in the split-bf16 mode: HR row bands against the oracle at the fp32 bound."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(123)
    feat_np = synth.encoder_features(123, 1, 256, 256)
    feat = torch.from_numpy(feat_np).to(dev)
    packed = D.pack_state_dict(sd).to(dev)
    full = D.decode_features(feat, packed, (1024, 1024), compute="bf16x3")
    again = D.decode_features(feat, packed, (1024, 1024), compute="bf16x3")
    torch.cuda.synchronize()
    assert torch.equal(full, again)
    full_np = full.cpu().numpy()
    for y0, y1 in [(0, 8), (508, 524), (1016, 1024)]:
        ref = orc.decode_reference_form(sd, feat_np, (1024, 1024), 30000, row_range=(y0, y1)).numpy()
        err = float(np.abs(full_np[:, :, y0:y1] - ref).max())
        assert err <= _tol(ref), f"rows {y0}:{y1} err {err:.3e}"


# ---------------------------------------------------------------------------------------------------------------------
# non-finite inputs: the reference's relu / conv / sin propagate NaN (diinn.py:133-138), and so
# does the HIP path: relu0 is the NaN-propagating v_maximum3_f32 (csrc/diinn_device.h)
# ---------------------------------------------------------------------------------------------------------------------
def _same_nonfinite(got, ref, what):
    gn, rn = ~np.isfinite(got), ~np.isfinite(ref)
    assert np.array_equal(gn, rn), f"{what}: non-finite pixels differ ({gn.sum()} vs {rn.sum()} in the reference)"
    assert rn.any() and not rn.all(), what
    fin = ~rn
    assert float(np.abs(got[fin] - ref[fin]).max()) <= _tol(ref[fin]), what


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), float("-inf")])
def test_nonfinite_feature_cell_propagates_like_the_reference(dev, bad):
    """One NaN / Inf feature value poisons exactly the HR pixels whose nearest cell has it in its 3x3 window -- the
    unfold's support -- and leaves every other pixel untouched (the Winograd form of the hoisted conv has the same
    support: a patch element only enters the outputs whose window holds it)."""
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(11)
    for (h, w, hu, wu, cy, cx) in [(20, 24, 66, 80, 7, 9), (20, 24, 66, 80, 0, 23), (12, 10, 24, 20, 11, 0)]:
        feat = synth.encoder_features(11, 1, h, w).copy()
        feat[0, 17, cy, cx] = bad
        ref = orc.decode_reference_form(sd, feat, (hu, wu), 30000).numpy()
        got = _decode(sd, feat, (hu, wu), dev)
        _same_nonfinite(got, ref, f"{bad} at ({cy},{cx}) of {h}x{w}")
        assert np.isnan(got[~np.isfinite(got)]).all()              # and what comes out is NaN, as in the reference


@pytest.mark.parametrize("compute", ["bf16", "bf16_full", "bf16x3"])
def test_nonfinite_feature_cell_bf16(dev, compute):
    import diinn_oracle as orc
    sd = synth.decoder_state_dict(11)
    feat = synth.encoder_features(11, 1, 20, 24).copy()
    feat[0, 3, 9, 9] = float("nan")
    ref = orc.decode_reference_form(sd, feat, (80, 96), 30000).numpy()
    got = _decode(sd, feat, (80, 96), dev, compute=compute)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    fin = ~np.isnan(ref)
    assert float(np.abs(got[fin] - ref[fin]).max()) <= 3e-3 * float(np.abs(ref[fin]).max())


def test_nonfinite_weights_propagate_like_the_reference(dev):
    """A NaN in a modulation weight, a synthesis weight or a bias makes the reference's whole output NaN; a NaN in one
    head bias only that colour plane.  (With v_max_f32 as relu a NaN modulation value became k = 0 and the pixel a
    finite colour: round 2's behaviour.)"""
    import diinn_oracle as orc
    feat = synth.encoder_features(3, 1, 10, 12)
    size = (33, 40)
    for key, index, whole in [("K.2.0.weight", (5, 300, 0, 0), True), ("K.1.0.weight", (9, 17, 0, 0), True),
                              ("Q.1.0.weight", (0, 0, 0, 0), True), ("K.0.0.bias", (100,), True),
                              ("last_layer.bias", (1,), False)]:
        sd = {k: v.copy() for k, v in synth.decoder_state_dict(3).items()}
        sd[key][index] = float("nan")
        ref = orc.decode_reference_form(sd, feat, size, 30000).numpy()
        got = _decode(sd, feat, size, dev)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), key
        assert np.isnan(ref).all() == whole, key
        if not whole:
            fin = ~np.isnan(ref)
            assert float(np.abs(got[fin] - ref[fin]).max()) <= _tol(ref[fin]), key


def test_image_without_derived_sections_is_refused(dev):
    """the gather-packed image of a training step has empty derived sections (WLR, BQR, Q0R, WLB, WPB, ...) and
    no DIINN_PACKED_MAGIC: its validity word is zero, or -- round 6, once its Winograd section 13 has been derived on the device
    for the training forward's P -- DIINN_PACKED_MAGIC_WPU, which only diinn_precompute_P_wpu accepts.  The inference entry
    points read the other derived sections; handed such an image they answer NaN everywhere instead of decoding with empty
    weights (the launch functions cannot look into device memory).  The entry points of the training path accept the same image."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_amd.training as T
    lib = N.load()
    sd = synth.decoder_state_dict(5)
    params = [torch.from_numpy(sd[n]).to(dev) for n in T.PARAM_NAMES]
    gathered = T.pack_on_device(params)
    host = D.pack_state_dict(sd).to(dev)
    off, size = C.c_size_t(), C.c_size_t()
    N.check(lib.diinn_packed_section(6, C.byref(off), C.byref(size)), "section")
    word = off.value + 3
    assert gathered[word:word + 1].view(torch.int32).item() == (N.PACKED_MAGIC_WPU if T.TRAIN_P_WINOGRAD else 0)
    assert host[word:word + 1].view(torch.int32).item() == N.PACKED_MAGIC
    feat = torch.from_numpy(synth.encoder_features(5, 1, 24, 20)).to(dev)
    good = D.decode_features(feat, host, (79, 66))
    assert bool(torch.isfinite(good).all())
    for compute in ("f32", "bf16", "bf16_full", "bf16x3"):
        out = D.decode_features(feat, gathered, (79, 66), compute=compute)
        torch.cuda.synchronize()
        assert bool(torch.isnan(out).all()), compute
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = torch.zeros(24 * 20 * 1024, device=dev)
    for comp in (N.COMPUTE_F32, N.COMPUTE_BF16_FULL):             # the stand-alone P entry points of the inference path
        P.zero_()
        N.check(lib.diinn_precompute_P_ex(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(gathered.data_ptr()),
                                          C.c_void_p(P.data_ptr()), 1, 24, 20, 0, 24, comp), "P_ex")
        torch.cuda.synchronize()
        assert bool(torch.isnan(P).all()), comp
    # the training path's P (direct kernel, permutation sections only) is fine with it
    N.check(lib.diinn_precompute_P(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(gathered.data_ptr()),
                                   C.c_void_p(P.data_ptr()), 1, 24, 20, 0, 24), "P")
    Pg = torch.empty_like(P)
    N.check(lib.diinn_precompute_P(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(host.data_ptr()),
                                   C.c_void_p(Pg.data_ptr()), 1, 24, 20, 0, 24), "P")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(P).all()) and torch.equal(P, Pg)
    # ... and the Winograd entry point of the training forward takes either image, bit-identically (the device-derived section 13
    # equals the host packer's); an image with neither word: NaN
    Pw, Pwh = torch.empty_like(P), torch.empty_like(P)
    N.check(lib.diinn_precompute_P_wpu(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(gathered.data_ptr()),
                                       C.c_void_p(Pw.data_ptr()), 1, 24, 20, 0, 24), "P_wpu")
    N.check(lib.diinn_precompute_P_wpu(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(host.data_ptr()),
                                       C.c_void_p(Pwh.data_ptr()), 1, 24, 20, 0, 24), "P_wpu")
    blank = gathered.clone()
    blank[word] = 0.0
    N.check(lib.diinn_precompute_P_wpu(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(blank.data_ptr()),
                                       C.c_void_p(Pg.data_ptr()), 1, 24, 20, 0, 24), "P_wpu")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(Pw).all()) and torch.equal(Pw, Pwh) and bool(torch.isnan(Pg).all())
    assert float((Pw - P).abs().max()) <= 1e-5 * max(1.0, float(P.abs().max()))      # Winograd vs direct: reassociation
