"""Quality metrics the reference's evaluation reports, without torchmetrics /
torchvision (absent from the target image):

  psnr, ssim      what ``torchmetrics.functional.peak_signal_noise_ratio`` /
                  ``structural_similarity_index_measure`` compute with the arguments the reference
                  passes (sr_module.py:167-170: ``data_range=1``, everything else default)
  calc_psnr       reference sr_module.py:21-38 (border-shaved PSNR, luma for 'benchmark')
  resize_fn       reference sr_module.py:16-19 / srdata.py:163-166 (antialiased bicubic)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """10 log10(data_range^2 / MSE), MSE over every element of the batch (torchmetrics default, base 10,
    elementwise_mean reduction with dim=None)."""
    mse = torch.mean((pred.to(torch.float32) - target.to(torch.float32)) ** 2)
    return 10.0 * torch.log10(torch.as_tensor(data_range ** 2, device=mse.device, dtype=mse.dtype) / mse)


def _gaussian_window(size: int, sigma: float, device, dtype) -> torch.Tensor:
    x = torch.arange(size, device=device, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-(x / sigma) ** 2 / 2.0)
    g = g / g.sum()
    return g[:, None] * g[None, :]


def ssim(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0, kernel_size: int = 11,
         sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """Mean SSIM with a gaussian window (torchmetrics defaults: 11x11, sigma 1.5, k1 0.01, k2 0.03):
    inputs are reflect-padded by (kernel-1)/2, filtered per channel, the SSIM map is cropped by the
    same margin and averaged per image, then over the batch."""
    pred = pred.to(torch.float32)
    target = target.to(torch.float32)
    b, c, h, w = pred.shape
    pad = (kernel_size - 1) // 2
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    win = _gaussian_window(kernel_size, sigma, pred.device, pred.dtype).expand(c, 1, kernel_size, kernel_size)
    p = F.pad(pred, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t], dim=0)
    out = F.conv2d(stack, win, groups=c)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(b, dim=0)
    s_pp = e_pp - mu_p * mu_p
    s_tt = e_tt - mu_t * mu_t
    s_pt = e_pt - mu_p * mu_t
    num = (2 * mu_p * mu_t + c1) * (2 * s_pt + c2)
    den = (mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2)
    m = (num / den)[..., pad:-pad, pad:-pad] if pad > 0 else num / den
    return m.reshape(b, -1).mean(-1).mean()


# border shave per evaluation protocol: (pixels shaved beyond `scale`, luma-only)
_SHAVE = {"benchmark": (0, True), "div2k": (6, False)}
_LUMA = (65.738 / 256, 129.057 / 256, 25.064 / 256)


def calc_psnr(sr: torch.Tensor, hr: torch.Tensor, dataset=None, scale=1, rgb_range=1) -> torch.Tensor:
    """The EDSR-lineage PSNR the reference's validation reports (behaviour of sr_module.py:21-38, pinned against that
    function's own outputs in tests/test_metrics.py): the error image is normalised by ``rgb_range``; protocol
    'benchmark' scores the luma of the error (BT.601 weights / 256) inside a border of ``scale`` pixels, 'div2k' all
    channels inside a border of ``scale + 6``; ``dataset=None`` scores everything."""
    err = (sr - hr) / rgb_range
    if dataset is not None:
        if dataset not in _SHAVE:
            raise NotImplementedError(f"calc_psnr: unknown protocol {dataset!r}")
        extra, luma = _SHAVE[dataset]
        if luma and err.size(1) > 1:
            err = torch.einsum("bchw,c->bhw", err, err.new_tensor(_LUMA))
        m = int(scale) + extra
        err = err[..., m:-m, m:-m]
    return -10 * torch.log10(err.square().mean())


def resize_fn(img: torch.Tensor, size) -> torch.Tensor:
    """Antialiased bicubic resize of a [..., H, W] tensor (what torchvision's
    ``Resize(size, BICUBIC, antialias=True)`` lowers to for tensors)."""
    squeeze = img.dim() == 3
    x = img.unsqueeze(0) if squeeze else img
    integer = not torch.is_floating_point(x)
    if integer:                                  # uint8 images (the data loaders): resize in fp32, clamp, round, cast back
        x = x.to(torch.float32)
    y = F.interpolate(x, size=tuple(int(s) for s in size), mode="bicubic", align_corners=False, antialias=True)
    if integer:
        y = y.clamp_(0, 255).round_().to(img.dtype)
    return y.squeeze(0) if squeeze else y
