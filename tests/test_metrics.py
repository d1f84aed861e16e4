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


# ---------------------------------------------------------------------------------------------------
# known-answer vectors (tests/golden/make_metrics_vectors.py): float64 restatements of the published algorithms the
# reference delegates to (torchmetrics PSNR / SSIM defaults, Pillow-style antialiased bicubic) and the reference's OWN
# calc_psnr body executed from its source -- evaluated in the build container, stored as data
# ---------------------------------------------------------------------------------------------------
import os

import pytest

import diinn_amd.synth as synth
from conftest import ROOT


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(ROOT, "tests", "golden", "metrics_vectors.npz"))


def _images(seed, shape, noise):
    a = (synth.uniform(seed, "metrics/target", shape, 0.5) + np.float32(0.5)).astype(np.float32)
    n = synth.uniform(seed, "metrics/noise", shape, noise)
    return a, np.clip(a + n, 0.0, 1.0).astype(np.float32)


def _cases(vec):
    for k in vec.files:
        if k.startswith("meta/") and k != "meta/resize_src":
            m = vec[k]
            yield k[5:], tuple(int(v) for v in m[:4]), float(m[4])


def test_psnr_ssim_calc_psnr_against_known_answers(vec):
    """fp32 implementation vs float64 expected values: PSNR/calc_psnr to 1e-4 dB, SSIM to 2e-5."""
    n = 0
    for name, shape, noise in _cases(vec):
        t, p = _images(11, shape, noise)
        tp, tt = torch.from_numpy(p), torch.from_numpy(t)
        assert abs(float(M.psnr(tp, tt, data_range=1)) - float(vec[f"psnr/{name}"])) < 1e-4, name
        assert abs(float(M.ssim(tp, tt, data_range=1)) - float(vec[f"ssim/{name}"])) < 2e-5, name
        assert abs(float(M.calc_psnr(tp, tt)) - float(vec[f"calc_psnr_none/{name}"])) < 1e-4, name
        assert abs(float(M.calc_psnr(tp, tt, dataset="div2k", scale=2)) - float(vec[f"calc_psnr_div2k_x2/{name}"])) < 1e-4
        assert abs(float(M.calc_psnr(tp, tt, dataset="benchmark", scale=3)) - float(vec[f"calc_psnr_benchmark_x3/{name}"])) < 1e-4
        n += 1
    assert n == 3


def test_antialiased_bicubic_against_known_answers(vec):
    """resize_fn (sr_module.py:16-19, srdata.py:163-166) down-, non-integer and up-scaling vs the float64 filter."""
    shape = tuple(int(v) for v in vec["meta/resize_src"])
    t, _ = _images(5, shape, 0.1)
    n = 0
    for k in vec.files:
        if not k.startswith("resize/"):
            continue
        size = tuple(int(v) for v in k[7:].split("x"))
        got = M.resize_fn(torch.from_numpy(t), size).numpy().astype(np.float64)
        assert got.shape == vec[k].shape
        assert float(np.abs(got - vec[k]).max()) < 2e-6, k
        n += 1
    assert n == 4
    t, p = _images(7, (1, 3, 48, 60), 0.05)
    lr = (16, 20)
    got = float(M.psnr(M.resize_fn(torch.from_numpy(p), lr), M.resize_fn(torch.from_numpy(t), lr), data_range=1))
    assert abs(got - float(vec["lr_psnr/x3_48x60"])) < 1e-4


@pytest.mark.gpu
def test_test_step_on_device_matches_known_answer_metrics(vec):
    """SRLitModule.test_step (sr_module.py:159-180) on the GPU: decode with the HIP path, then the three metrics on
    device; each must equal the same metric evaluated from the returned prediction on the CPU in float64
    (the generator's restatements, imported from tests/golden), and the device metrics must reproduce the
    known answers on the fixture images."""
    import importlib.util
    import diinn_amd.modules as MM
    spec = importlib.util.spec_from_file_location("mmv", os.path.join(ROOT, "tests", "golden", "make_metrics_vectors.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    dev = torch.device("cuda:0")
    for name, shape, noise in _cases(vec):
        t, p = _images(11, shape, noise)
        tp, tt = torch.from_numpy(p).to(dev), torch.from_numpy(t).to(dev)
        assert abs(float(M.psnr(tp, tt, data_range=1)) - float(vec[f"psnr/{name}"])) < 1e-4
        assert abs(float(M.ssim(tp, tt, data_range=1)) - float(vec[f"ssim/{name}"])) < 2e-5
    torch.manual_seed(0)
    model = MM.SRLitModule(arch="diinn", mode=3, init_q=False).to(dev).eval()
    hr = torch.rand(1, 3, 48, 60, device=dev)
    batch = {3: (M.resize_fn(hr, (16, 20)).clamp(0, 1), hr, "synthetic")}
    res = model.test_step(batch, 0)
    _, preds = model.step(batch, model.hparams.eval_bsize)
    pred = preds[3].cpu().numpy()
    hr_np = hr.cpu().numpy()
    assert abs(float(res[3]["psnr_res"]) - gen.psnr64(pred, hr_np)) < 1e-3
    assert abs(float(res[3]["ssim_res"]) - gen.ssim64(pred, hr_np)) < 5e-5
    want_lr = gen.psnr64(gen.resize64(pred, (16, 20)), gen.resize64(hr_np, (16, 20)))
    assert abs(float(res[3]["lr_psnr_res"]) - want_lr) < 1e-3
