"""CPU: the evaluation metrics of the reference's test_step, re-implemented without
torchmetrics/torchvision; checked against independent numpy restatements and known values."""
import numpy as np
import torch

import diinn_amd.metrics as M


def _ssim_numpy(a, b, data_range=1.0, size=11, sigma=1.5, k1=0.01, k2=0.03):
    """Direct (slow) gaussian-window SSIM on reflect-padded single-channel images, cropped like torchmetrics."""
    pad = (size - 1) // 2
    x = np.arange(size) - (size - 1) / 2.0
    g = np.exp(-(x / sigma) ** 2 / 2.0)
    g /= g.sum()
    win = np.outer(g, g)
    ap, bp = np.pad(a, pad, mode="reflect"), np.pad(b, pad, mode="reflect")
    h, w = a.shape
    vals = []
    for y in range(pad, h - pad):
        for xx in range(pad, w - pad):
            pa, pb = ap[y:y + size, xx:xx + size], bp[y:y + size, xx:xx + size]
            mu_a, mu_b = (win * pa).sum(), (win * pb).sum()
            s_aa, s_bb = (win * pa * pa).sum() - mu_a ** 2, (win * pb * pb).sum() - mu_b ** 2
            s_ab = (win * pa * pb).sum() - mu_a * mu_b
            c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
            vals.append(((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (s_aa + s_bb + c2)))
    return float(np.mean(vals))


def test_psnr_known_values():
    a = torch.zeros(1, 3, 8, 8)
    b = torch.full((1, 3, 8, 8), 0.1)
    assert abs(float(M.psnr(a, b)) - 20.0) < 1e-4            # mse = 0.01 -> 20 dB
    assert abs(float(M.psnr(a * 255, b * 255, data_range=255)) - 20.0) < 1e-3


def test_ssim_matches_direct_evaluation_and_bounds():
    rng = np.random.default_rng(0)
    a = rng.random((1, 1, 24, 20)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape).astype(np.float32), 0, 1)
    got = float(M.ssim(torch.from_numpy(a), torch.from_numpy(b)))
    want = _ssim_numpy(a[0, 0].astype(np.float64), b[0, 0].astype(np.float64))
    assert abs(got - want) < 1e-4
    assert abs(float(M.ssim(torch.from_numpy(a), torch.from_numpy(a))) - 1.0) < 1e-6
    rgb = torch.rand(2, 3, 16, 16)
    assert 0.0 < float(M.ssim(rgb, (rgb + 0.2).clamp(0, 1))) < 1.0


def test_calc_psnr_shave_and_luma():
    torch.manual_seed(0)
    sr, hr = torch.rand(1, 3, 20, 20), torch.rand(1, 3, 20, 20)
    d = sr - hr
    luma = (d * torch.tensor([65.738, 129.057, 25.064]).view(1, 3, 1, 1) / 256).sum(1)[..., 4:-4, 4:-4]
    assert abs(float(M.calc_psnr(sr, hr, "benchmark", scale=4)) - float(-10 * torch.log10((luma ** 2).mean()))) < 1e-5
    assert abs(float(M.calc_psnr(sr, hr, "div2k", scale=2)) - float(-10 * torch.log10((d[..., 8:-8, 8:-8] ** 2).mean()))) < 1e-5
    assert abs(float(M.calc_psnr(sr, hr)) - float(M.psnr(sr, hr))) < 1e-5


def test_resize_fn_shapes_and_constant():
    x = torch.full((3, 17, 23), 0.25)
    y = M.resize_fn(x, (5, 9))
    assert y.shape == (3, 5, 9) and torch.allclose(y, torch.full_like(y, 0.25), atol=1e-6)
    assert M.resize_fn(torch.rand(2, 3, 8, 8), (16, 12)).shape == (2, 3, 16, 12)
