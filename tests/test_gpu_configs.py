"""GPU parity at BASELINE.json's FULL sizes: c3 (512^2 x4), the north-star target (1024^2 x4), c4 (1024^2 x8)
and c5 (720x1280 x3.3, fp32 + both bf16 modes).  The oracle cannot decode these images whole in seconds, so
each test decodes the full image on the GPU and checks
  * HR row bands at the top, middle and bottom against the oracle fed a feature crop (band cells + halo);
  * size-independent properties: determinism (two runs bit-equal), a band decoded through the row-window
    entry points (diinn_decode_win, band-sized buffers = the multi-GPU unit) bit-equal to the same rows of
    the full decode, every value finite;
  * the ABI's size limits just past the largest config.
Tolerances: fp32 1e-4 * max(1, max|ref|) (north_star); bf16 2e-3 * max|ref|, bf16_full 3e-3 * max|ref|
(restated, SURVEY §8 d4 / DESIGN §3.4)."""
import ctypes as C

import numpy as np
import pytest
import torch

import diinn_amd.synth as synth

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def weights(dev):
    import diinn_amd.decoder as D
    sd = synth.decoder_state_dict(123)
    return sd, D.pack_state_dict(sd).to(dev)


def _features(dev, h, w, seed):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    return torch.randn((1, 64, h, w), device=dev, generator=gen)


def _crop_for(feat, h, hu, wu, y0, y1):
    import diinn_amd.decoder as D
    (a0, an), _ = D.window_rows(h, hu, wu, y0, y1)
    return a0, feat[:, :, a0:a0 + an].contiguous()


def _check_bands(sd, feat, out, size, bands, rel_tol, floor_one=True, emu=None):
    """out rows vs the reference-form oracle on feature crops; returns the worst error / its bound."""
    import diinn_oracle as orc
    h = feat.shape[2]
    hu, wu = size
    worst = 0.0
    for (y0, y1) in bands:
        a0, crop = _crop_for(feat, h, hu, wu, y0, y1)
        crop = crop.cpu()
        ref = orc.decode_reference_form(sd, crop, size, 30000, row_range=(y0, y1), feat_row0=a0, full_h=h).numpy()
        got = out[:, :, y0:y1].cpu().numpy()
        scale = float(np.abs(ref).max())
        tol = rel_tol * (max(1.0, scale) if floor_one else scale)
        err = float(np.abs(got - ref).max())
        assert err <= tol, f"rows {y0}:{y1}: err {err:.3e} > {tol:.3e}"
        worst = max(worst, err / tol)
        if emu is not None:
            e = orc.decode_hoisted_form(sd, crop, size, row_range=(y0, y1), feat_row0=a0, full_h=h, **emu).numpy()
            err2 = float(np.abs(got - e).max())
            assert err2 <= tol, f"rows {y0}:{y1}: vs emulation {err2:.3e} > {tol:.3e}"
    return worst


def _window_equals_full(feat, packed, size, full, y0, y1, compute="f32"):
    import diinn_amd.decoder as D
    h = feat.shape[2]
    a0, crop = _crop_for(feat, h, size[0], size[1], y0, y1)
    band = D.decode_window(crop, a0, h, packed, size, (y0, y1), compute=compute)
    torch.cuda.synchronize()
    assert torch.equal(band, full[:, :, y0:y1]), f"window decode of rows {y0}:{y1} differs from the full decode"


def _full_size_case(dev, weights, h, w, hu, wu, seed, bands, win_band, compute="f32"):
    import diinn_amd.decoder as D
    sd, packed = weights
    feat = _features(dev, h, w, seed)
    ws = torch.empty(h * w * 1024, device=dev)
    full = D.decode_features(feat, packed, (hu, wu), workspace=ws, compute=compute)
    again = D.decode_features(feat, packed, (hu, wu), workspace=ws, compute=compute)
    torch.cuda.synchronize()
    assert torch.equal(full, again), "decode is not deterministic"
    assert bool(torch.isfinite(full).all())
    del again, ws
    worst = _check_bands(sd, feat, full, (hu, wu), bands, TOL, emu={"bf16x3": True} if compute == "bf16x3" else None)
    _window_equals_full(feat, packed, (hu, wu), full, *win_band, compute=compute)
    print(f"{h}x{w} -> {hu}x{wu} ({compute}): worst band error = {worst:.3f} of the 1e-4 bound")


def test_config3_512_x4(dev, weights):
    """BASELINE config 3: 512x512 LR x4 -> 2048x2048."""
    _full_size_case(dev, weights, 512, 512, 2048, 2048, 3, [(0, 4), (1022, 1026), (2044, 2048)], (512, 1024))


def test_target_1024_x4(dev, weights):
    """north_star target shape: 1024x1024 LR x4 -> 4096x4096 (16.8 Mpixel, P workspace 4.3 GB)."""
    _full_size_case(dev, weights, 1024, 1024, 4096, 4096, 4, [(0, 4), (2046, 2050), (4092, 4096)], (1536, 2048))


def test_target_1024_x4_split_bf16(dev, weights):
    """The same shape in the optional split-bf16 mode (DIINN_COMPUTE_BF16X3; 131,072 blocks over 256 persistent
    workgroups), held to the SAME 1e-4 bound as fp32, and to the oracle's emulation of its roundings."""
    _full_size_case(dev, weights, 1024, 1024, 4096, 4096, 4, [(0, 4), (2046, 2050), (4092, 4096)], (1536, 2048),
                    compute="bf16x3")


def test_config4_1024_x8(dev, weights):
    """BASELINE config 4: 1024x1024 LR x8 -> 8192x8192 (67 Mpixel, 805 MB of output), one GPU's view and an
    1/8 band through the window entry points (what each of the 8 GPUs runs)."""
    _full_size_case(dev, weights, 1024, 1024, 8192, 8192, 5, [(0, 2), (4095, 4098), (8188, 8192)], (7168, 8192))


def test_size_limits_just_past_config4(dev, weights):
    """The kernels index with 32 bits in places: extents past the supported range are refused with
    DIINN_ERR_TOO_LARGE instead of being mis-computed; the largest BASELINE config is inside."""
    import diinn_amd._native as N
    lib = N.load()
    _, packed = weights
    dummy = torch.zeros(16, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def band(h, w, hu, wu, y0, y1, b=1):
        return lib.diinn_decode_band_ex(stream, C.c_void_p(dummy.data_ptr()), C.c_void_p(packed.data_ptr()),
                                        C.c_void_p(dummy.data_ptr()), b, h, w, hu, wu, y0, y1, N.SIN_DEFAULT, N.COMPUTE_F32)
    assert band(1024, 1024, 46341, 46341, 0, 8) == N.ERR_TOO_LARGE          # Hu*Wu >= 2e9 pixels
    assert band(1024, 1024, 1 << 20, 1024, 0, (1 << 20)) == N.ERR_TOO_LARGE  # > 65535 workgroup rows in one band
    assert band(70000, 16, 70000, 16, 0, 8) == N.ERR_TOO_LARGE               # H > 65535
    assert band(16, 16, 32, 32, 0, 8, b=70000) == N.ERR_TOO_LARGE            # B > 65535
    # windows that do not hold what the band reads are refused, not read out of bounds
    assert lib.diinn_decode_band_win(stream, C.c_void_p(dummy.data_ptr()), 10, 4, C.c_void_p(packed.data_ptr()),
                                     C.c_void_p(dummy.data_ptr()), 0, 8, 1, 64, 64, 256, 256, 0, 8, N.SIN_DEFAULT,
                                     N.COMPUTE_F32) == N.ERR_INVALID_ARG
    torch.cuda.synchronize()


@pytest.mark.parametrize("compute,rel", [("f32", None), ("bf16x3", None), ("bf16", 2e-3), ("bf16_full", 3e-3)])
def test_config5_720p_x3p3(dev, weights, compute, rel):
    """BASELINE config 5: 720x1280 LR -> 2376x4224 HR (x3.3, non-integer) in fp32 and on both bf16 MFMA paths.
    bf16 bands are checked against the fp32 oracle at the restated tolerance and against the oracle's emulation
    of the same operand roundings."""
    import diinn_amd.decoder as D
    sd, packed = weights
    h, w, hu, wu = 720, 1280, 2376, 4224
    feat = _features(dev, h, w, 6)
    ws = torch.empty(h * w * 1024, device=dev)
    full = D.decode_features(feat, packed, (hu, wu), workspace=ws, compute=compute)
    again = D.decode_features(feat, packed, (hu, wu), workspace=ws, compute=compute)
    torch.cuda.synchronize()
    assert torch.equal(full, again)
    assert bool(torch.isfinite(full).all())
    bands = [(0, 4), (1187, 1191), (2372, 2376)]
    if compute == "f32":
        worst = _check_bands(sd, feat, full, (hu, wu), bands, TOL)
    elif compute == "bf16x3":                                     # split bf16: the fp32 bound
        worst = _check_bands(sd, feat, full, (hu, wu), bands, TOL, emu={"bf16x3": True})
    else:
        emu = {"bf16_operands": True, "bf16_p": compute == "bf16_full"}
        worst = _check_bands(sd, feat, full, (hu, wu), bands, rel, floor_one=False, emu=emu)
    _window_equals_full(feat, packed, (hu, wu), full, 1188, 1485, compute=compute)
    print(f"c5 {compute}: worst band error = {worst:.3f} of its bound")
