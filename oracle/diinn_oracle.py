"""CPU ORACLE for the DIINN implicit-decoder hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, what the reference's
``ImplicitDecoder.forward`` (mode 3, init_q=False) computes:
``/root/reference/src/models/components/diinn.py:163-173`` and the helpers it
calls.  It is the *checker* for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product package never imports anything from ``oracle/``.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference decoder in the build container, runs it on inputs regenerated from
``synth.py`` and commits its outputs and coordinate/index tables under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here
against those fixtures (tables bit-exact, outputs <= 1e-6).

Functions (each cites the reference lines it follows):
  axis_centres / nearest_exact_index / axis_tables / scale_ratio
      -> diinn.py:94-110 (_make_pos_encoding) and :166 (ratio), plus ATen's
         CPU nearest-exact index
  unfold3x3                -> diinn.py:168 (F.unfold(x, 3, padding=1))
  decode_reference_form    -> diinn.py:132-139 (step, mode 3), :149-160
                              (batched_step), :163-173 (forward)
  decode_hoisted_form      -> same result via the per-cell hoist the HIP path
                              uses; a debugging aid.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

HIDDEN = 256
IN_CHANNELS = 64
UNFOLD = IN_CHANNELS * 9


# --------------------------------------------------------------------------
# coordinates and nearest-exact indices  (diinn.py:94-110)
# --------------------------------------------------------------------------
def axis_centres(n: int) -> np.ndarray:
    """``-1 + 1/n + 2/n * arange(n).float()`` exactly as torch evaluates it
    (diinn.py:98-99,102-103): the python scalars are doubles rounded to fp32
    when they meet the fp32 tensor; mul then add, each rounded to fp32."""
    c0 = np.float32(-1.0 + 1.0 / n)
    c1 = np.float32(2.0 / n)
    i = np.arange(n, dtype=np.float32)
    return (c1 * i).astype(np.float32) + c0


SMALL_OUTPUT_SUM = 128  # ATen: (out_H + out_W) <= 128 selects the other CPU kernel


def uses_small_output_kernel(hu: int, wu: int) -> bool:
    """ATen picks between two CPU nearest-exact kernels by output size
    (aten/src/ATen/native/cpu/UpSampleKernel.cpp, _use_vectorized_kernel_cond_2d):
    the NCHW tensors the reference interpolates (diinn.py:106,168) take the
    'generic' kernel unless out_H + out_W <= 128.  The two kernels round the
    source index differently, so the rule is part of the reference's behaviour."""
    return (int(hu) + int(wu)) <= SMALL_OUTPUT_SUM


def nearest_exact_index(n_in: int, n_out: int, small_output: bool = False) -> np.ndarray:
    """Source index of ``F.interpolate(mode='nearest-exact')`` on the CPU
    (diinn.py:106,168), verified against torch 2.10 on >200k random size pairs
    including every exact-tie pair found (tests/golden: idx/*).

    generic kernel (out_H+out_W > 128), HelperInterpNearestExact: in fp32 with a
    fused multiply-add ``r = max(fma(s, j+0.5, -0.5), 0)``; index =
    ``floorf(float(double(r) + 0.5))``.
    small-output kernel, nearest_exact_idx (ATen/native/UpSample.h:360-367 ->
    :326-337): ``floorf(float((j + 0.5) * double(s)))`` (double product).
    Both use ``s = float(n_in) / n_out`` and clamp to n_in-1."""
    s = np.float32(n_in) / np.float32(n_out)
    if small_output:
        j = np.arange(n_out, dtype=np.float64) + 0.5
        idx = np.floor((j * np.float64(s)).astype(np.float32)).astype(np.int64)
        return np.minimum(idx, n_in - 1).astype(np.int32)
    j = np.arange(n_out, dtype=np.float32) + np.float32(0.5)
    # single-rounding fma(s, j, -0.5): exact in float64 (24x24-bit product), one rounding to fp32
    r = (s.astype(np.float64) * j.astype(np.float64) - 0.5).astype(np.float32)
    r = np.maximum(r, np.float32(0.0))
    idx = np.floor((r.astype(np.float64) + 0.5).astype(np.float32)).astype(np.int64)
    return np.minimum(idx, n_in - 1).astype(np.int32)


def axis_tables(n_in: int, n_out: int, small_output: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """(idx[n_out] int32, rel[n_out] fp32) for one axis:
    ``rel = (up - in[idx]) * n_in`` in fp32 op order (diinn.py:106-108)."""
    idx = nearest_exact_index(n_in, n_out, small_output)
    g_in = axis_centres(n_in)
    g_out = axis_centres(n_out)
    rel = (g_out - g_in[idx]).astype(np.float32) * np.float32(n_in)
    return idx, rel.astype(np.float32)


def scale_ratio(h: int, w: int, hu: int, wu: int) -> np.float32:
    """``x.new_tensor([(H*W)/(Hu*Wu)])`` (diinn.py:166): python double -> fp32."""
    return np.float32((h * w) / (hu * wu))


# --------------------------------------------------------------------------
# decoder maths
# --------------------------------------------------------------------------
def _as_t(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.detach().to(torch.float32).cpu()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def unfold3x3(feat: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] -> [B,C*9,H,W], channel index c*9+ky*3+kx, zero padding 1
    (diinn.py:168: ``F.unfold(x, 3, padding=1).view(B, C*9, H, W)``).
    Written with explicit shifted copies instead of F.unfold."""
    b, c, h, w = feat.shape
    padded = F.pad(feat, (1, 1, 1, 1))
    out = feat.new_empty((b, c, 9, h, w))
    for ky in range(3):
        for kx in range(3):
            out[:, :, ky * 3 + kx] = padded[:, :, ky:ky + h, kx:kx + w]
    return out.view(b, c * 9, h, w)


def _conv1x1(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return F.conv2d(x, w.view(w.shape[0], -1, 1, 1), b)


def _step_mode3(sd: Dict[str, torch.Tensor], x: torch.Tensor, syn: torch.Tensor, mode: int = 3) -> torch.Tensor:
    """diinn.py:116-139 with K[i]/Q[i] = Conv1x1 + ReLU / sin (diinn.py:53-78).
    mode 3 (:132-139): K[i] sees [q; x].  mode 2 (:124-131): K[i] sees [k; x].  mode 1 (:116-123): K[i] sees k."""
    k = torch.relu(_conv1x1(x, sd["K.0.0.weight"], sd["K.0.0.bias"]))
    q = k * torch.sin(_conv1x1(syn, sd["Q.0.0.weight"], sd["Q.0.0.bias"]))
    for i in range(1, 4):
        if mode == 3:
            kin = torch.cat([q, x], dim=1)
        elif mode == 2:
            kin = torch.cat([k, x], dim=1)
        else:
            kin = k
        k = torch.relu(_conv1x1(kin, sd[f"K.{i}.0.weight"], sd[f"K.{i}.0.bias"]))
        q = k * torch.sin(_conv1x1(q, sd[f"Q.{i}.0.weight"], sd[f"Q.{i}.0.bias"]))
    return _conv1x1(q, sd["last_layer.weight"], sd["last_layer.bias"])


def make_syn_inp(b: int, h: int, w: int, hu: int, wu: int) -> Tuple[torch.Tensor, np.ndarray, np.ndarray]:
    """[B,3,Hu,Wu] = (rel_h, rel_w, ratio) (diinn.py:165-167) + the index tables."""
    small = uses_small_output_kernel(hu, wu)
    idx_h, rel_h = axis_tables(h, hu, small)
    idx_w, rel_w = axis_tables(w, wu, small)
    syn = torch.empty((b, 3, hu, wu), dtype=torch.float32)
    syn[:, 0] = torch.from_numpy(rel_h)[None, :, None]
    syn[:, 1] = torch.from_numpy(rel_w)[None, None, :]
    syn[:, 2] = float(scale_ratio(h, w, hu, wu))
    return syn, idx_h, idx_w


@torch.no_grad()
def decode_reference_form_f64(sd, feat, size: Sequence[int]) -> torch.Tensor:
    """The same mathematics in float64 (coordinates and indices stay the reference's fp32 tables):
    a ground truth against which the fp32 reference and the fp32 HIP path can both be measured."""
    sd64 = {k: _as_t(v).double() for k, v in sd.items()}
    feat = _as_t(feat).double()
    b, c, h, w = feat.shape
    hu, wu = int(size[0]), int(size[1])
    syn, idx_h, idx_w = make_syn_inp(b, h, w, hu, wu)
    u = unfold3x3(feat)
    x = u[:, :, torch.from_numpy(idx_h.astype(np.int64))][:, :, :, torch.from_numpy(idx_w.astype(np.int64))]
    return _step_mode3(sd64, x, syn.double())


@torch.no_grad()
def decode_reference_form(sd, feat, size: Sequence[int], bsize: Optional[int] = None,
                          row_range: Optional[Tuple[int, int]] = None, mode: int = 3,
                          feat_row0: int = 0, full_h: Optional[int] = None) -> torch.Tensor:
    """Reference-faithful CPU decode: unfold -> nearest-exact replicate ->
    9 conv1x1 + 3 cat + 4 sin, optional column strips of ``bsize//Hu`` columns
    (diinn.py:149-160).  ``row_range=(y0,y1)`` restricts the output to an HR
    row band (what one GPU computes under tile sharding); the maths per pixel
    is unchanged.

    ``feat_row0`` / ``full_h``: ``feat`` holds only LR rows [feat_row0, feat_row0 + feat.shape[2]) of a map
    of height ``full_h`` (for checking bands of maps too large to unfold whole on the CPU).  The crop must
    contain the band's cells and their 3x3 halo, so that the unfold's zero padding only ever stands for
    rows that are outside the full map as well."""
    sd = {k: _as_t(v) for k, v in sd.items()}
    feat = _as_t(feat)
    b, c, hc, w = feat.shape
    h = hc if full_h is None else int(full_h)
    hu, wu = int(size[0]), int(size[1])
    syn, idx_h, idx_w = make_syn_inp(b, h, w, hu, wu)
    y0, y1 = (0, hu) if row_range is None else row_range
    rows = idx_h[y0:y1].astype(np.int64)
    lo, hi = int(rows.min()), int(rows.max())
    if not (feat_row0 <= max(lo - 1, 0) and min(hi + 1, h - 1) <= feat_row0 + hc - 1):
        raise ValueError("feature crop does not cover the band's cells + halo")
    ih = torch.from_numpy(rows - feat_row0)
    iw = torch.from_numpy(idx_w.astype(np.int64))
    u = unfold3x3(feat)
    x = u[:, :, ih][:, :, :, iw]  # nearest-exact replication through the tables
    syn = syn[:, :, y0:y1]
    if bsize is None:
        return _step_mode3(sd, x, syn, mode)
    hh = y1 - y0
    cols = max(int(bsize) // hh, 1)  # reference hangs when bsize < Hu (diinn.py:155); clamp instead
    preds = []
    ql = 0
    while ql < wu:
        qr = min(ql + cols, wu)
        preds.append(_step_mode3(sd, x[..., ql:qr], syn[..., ql:qr], mode))
        ql = qr
    return torch.cat(preds, dim=-1)


def split_weights(sd) -> Dict[str, torch.Tensor]:
    """Algebraic split of the reference weights:
    Wx[4,256,64,3,3] (feature half of every K[i], a 3x3 conv kernel),
    Wq[3,256,256] (q half of K[1..3])."""
    sd = {k: _as_t(v) for k, v in sd.items()}
    wx = [sd["K.0.0.weight"].view(HIDDEN, UNFOLD)]
    wq = []
    for i in range(1, 4):
        wfull = sd[f"K.{i}.0.weight"].view(HIDDEN, HIDDEN + UNFOLD)
        wq.append(wfull[:, :HIDDEN])      # cat([q, x]) -> first 256 input channels are q (diinn.py:136)
        wx.append(wfull[:, HIDDEN:])
    return {
        "Wx": torch.stack(wx).view(4, HIDDEN, IN_CHANNELS, 3, 3).contiguous(),
        "bK": torch.stack([sd[f"K.{i}.0.bias"] for i in range(4)]),
        "Wq": torch.stack(wq).contiguous(),
        "Q0": sd["Q.0.0.weight"].view(HIDDEN, 3),
        "Qw": torch.stack([sd[f"Q.{i}.0.weight"].view(HIDDEN, HIDDEN) for i in range(1, 4)]),
        "bQ": torch.stack([sd[f"Q.{i}.0.bias"] for i in range(4)]),
        "L": sd["last_layer.weight"].view(3, HIDDEN),
        "bL": sd["last_layer.bias"],
    }


@torch.no_grad()
def precompute_P(sd, feat) -> torch.Tensor:
    """P[b, y, x, i*256+ch] = Wx_i . unfold(feat)[b,:,y,x] + bK_i  (one 3x3 conv 64->1024)."""
    sw = split_weights(sd)
    feat = _as_t(feat)
    p = F.conv2d(feat, sw["Wx"].view(4 * HIDDEN, IN_CHANNELS, 3, 3), sw["bK"].view(-1), padding=1)
    return p.permute(0, 2, 3, 1).contiguous()  # [B,H,W,1024]


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round to nearest even) -> fp32: what an MFMA bf16 operand holds."""
    return t.to(torch.bfloat16).to(torch.float32)


@torch.no_grad()
def decode_hoisted_form(sd, feat, size: Sequence[int], bf16_operands: bool = False, bf16_p: bool = False,
                        row_range: Optional[Tuple[int, int]] = None, feat_row0: int = 0,
                        full_h: Optional[int] = None, bf16_p_storage: bool = False,
                        bf16x3: bool = False, bf16x3_p: bool = False) -> torch.Tensor:
    """Same function evaluated the way the HIP kernels do: per-cell P, then the
    per-pixel 256->512 stacked layers.  Not reference-faithful in summation
    order; agrees with decode_reference_form to ~1e-7.

    ``bf16_operands=True`` emulates the optional bf16 path (BASELINE config 5): the weights and
    the activation entering layers 1..3 are rounded to bf16, products accumulate in fp32;
    P, biases, sine, layer 0 and the head (on the unrounded last activation) stay fp32.
    ``bf16_p=True`` additionally rounds the features and the 3x3 conv weights of P to bf16 (fp32 accumulate,
    fp32 bias): the DIINN_COMPUTE_BF16_FULL mode.
    ``bf16_p_storage=True`` rounds the finished P image (bias included) to bf16, as a kernel that kept P in HBM as
    bf16 would: an experiment switch, measured by tools/bf16_p_storage_error.py; no kernel does it.
    ``bf16x3=True`` emulates the split-bf16 mode (DIINN_COMPUTE_BF16X3, decode_bf16x3_kernel): weights and
    activation of layers 1..3 as hi = bf16(v), lo = bf16(v - hi); a product is w_lo.q_hi + w_hi.q_lo + w_hi.q_hi with
    fp32 accumulation; the synthesis branch in revolutions as in the bf16 mode; everything else fp32.
    ``bf16x3_p=True`` (with ``bf16x3``): the hoisted conv P in the same split arithmetic (features and Wx as hi + lo, three
    products, fp32 accumulation, fp32 bias): what the mode runs on maps of >= 32,768 cells (precompute_P_x3_kernel).
    ``row_range`` / ``feat_row0`` / ``full_h``: an HR row band from a feature crop, as in decode_reference_form."""
    if bf16x3 and (bf16_operands or bf16_p):
        raise ValueError("bf16x3 is a mode of its own")
    sw = split_weights(sd)
    feat = _as_t(feat)
    b, c, hc, w = feat.shape
    h = hc if full_h is None else int(full_h)
    hu, wu = int(size[0]), int(size[1])
    small = uses_small_output_kernel(hu, wu)
    idx_h, rel_h = axis_tables(h, hu, small)
    idx_w, rel_w = axis_tables(w, wu, small)
    y0, y1 = (0, hu) if row_range is None else row_range
    rows = idx_h[y0:y1].astype(np.int64)
    if not (feat_row0 <= max(int(rows.min()) - 1, 0) and min(int(rows.max()) + 1, h - 1) <= feat_row0 + hc - 1):
        raise ValueError("feature crop does not cover the band's cells + halo")
    if bf16x3 and bf16x3_p:
        wx = sw["Wx"].view(4 * HIDDEN, IN_CHANNELS, 3, 3)
        fh = _bf16_round(feat)
        fl = _bf16_round(feat - fh)
        wh = _bf16_round(wx)
        wl = _bf16_round(wx - wh)
        p = (F.conv2d(fh, wl, None, padding=1) + F.conv2d(fl, wh, None, padding=1)) + F.conv2d(fh, wh, None, padding=1)
        p = (p + sw["bK"].view(1, -1, 1, 1)).permute(0, 2, 3, 1).contiguous()
    elif bf16_p:
        p = F.conv2d(_bf16_round(feat), _bf16_round(sw["Wx"].view(4 * HIDDEN, IN_CHANNELS, 3, 3)), None, padding=1)
        p = (p + sw["bK"].view(1, -1, 1, 1)).permute(0, 2, 3, 1).contiguous()
    else:
        p = precompute_P(sd, feat)  # [B,hc,W,1024]
    if bf16_p_storage:
        p = _bf16_round(p)
    pp = p[:, torch.from_numpy(rows - feat_row0)][:, :, torch.from_numpy(idx_w.astype(np.int64))]
    nh = y1 - y0
    pp = pp.view(b, nh, wu, 4, HIDDEN)
    syn = torch.empty((nh, wu, 3))
    syn[..., 0] = torch.from_numpy(rel_h[y0:y1])[:, None]
    syn[..., 1] = torch.from_numpy(rel_w)[None, :]
    syn[..., 2] = float(scale_ratio(h, w, hu, wu))
    q = torch.relu(pp[:, :, :, 0]) * torch.sin(syn @ sw["Q0"].t() + sw["bQ"][0])
    rnd = _bf16_round if bf16_operands else (lambda t: t)
    inv_2pi = torch.tensor(0.15915494309189533577, dtype=torch.float32)
    def mm3(a, wgt):                                   # a [.., K] x wgt [M, K]: three bf16 products
        ah = _bf16_round(a)
        al = _bf16_round(a - ah)
        wh = _bf16_round(wgt)
        wl = _bf16_round(wgt - wh)
        return (ah @ wl.t() + al @ wh.t()) + ah @ wh.t()

    for i in range(1, 4):
        if bf16x3:
            k = torch.relu(mm3(q, sw["Wq"][i - 1]) + pp[:, :, :, i])
            rev = mm3(q, sw["Qw"][i - 1] * inv_2pi) + sw["bQ"][i] * inv_2pi
            q = k * torch.sin(rev.double() * (2.0 * np.pi)).float()
            continue
        qi = rnd(q)
        k = torch.relu(qi @ rnd(sw["Wq"][i - 1]).t() + pp[:, :, :, i])
        if bf16_operands:
            # the bf16 kernels keep the synthesis branch in REVOLUTIONS: weights are multiplied by fp32(1/(2 pi))
            # before the bf16 rounding and the bias after it (packed sections 7 and 10, diinn_host.cpp), and the
            # sine is taken of 2 pi x (here in float64, so that only the operand roundings are emulated)
            rev = qi @ rnd(sw["Qw"][i - 1] * inv_2pi).t() + sw["bQ"][i] * inv_2pi
            q = k * torch.sin(rev.double() * (2.0 * np.pi)).float()
        else:
            q = k * torch.sin(qi @ sw["Qw"][i - 1].t() + sw["bQ"][i])
    out = q @ sw["L"].t() + sw["bL"]
    return out.permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------
# training path (reference: step() under autograd, diinn.py:132-139 via forward(bsize=None) :170-171;
# caller SRLitModule.training_step, sr_module.py:127-129)
# --------------------------------------------------------------------------
def reference_gradients(sd, feat, size: Sequence[int], grad_out, dtype=torch.float32):
    """d sum(out * grad_out) / d(feat, every parameter) by autograd through the reference-form graph
    (unfold -> nearest-exact replicate -> the 9 conv1x1 of step()).  Pinned by
    tests/golden/diinn_golden_grad.npz (the real reference's .grad values).  ``dtype=torch.float64``
    gives a ground truth for error budgets.  Returns (out, d_feat, {name: grad})."""
    params = {k: _as_t(v).to(dtype).requires_grad_(True) for k, v in sd.items()}
    f = _as_t(feat).to(dtype).requires_grad_(True)
    b, c, h, w = f.shape
    hu, wu = int(size[0]), int(size[1])
    syn, idx_h, idx_w = make_syn_inp(b, h, w, hu, wu)
    u = unfold3x3(f)
    x = u[:, :, torch.from_numpy(idx_h.astype(np.int64))][:, :, :, torch.from_numpy(idx_w.astype(np.int64))]
    out = _step_mode3(params, x, syn.to(dtype))
    (out * _as_t(grad_out).to(dtype)).sum().backward()
    return out.detach(), f.grad, {k: v.grad for k, v in params.items()}


@torch.no_grad()
def saved_planes(sd, feat, size: Sequence[int]):
    """What the training forward kernel leaves for the backward pass, restated on the CPU:
    acts[i, 0] = k_i (rectified modulation), acts[i, 1] = s_i (sine argument), i = 0..3, as
    [256, B*Hu*Wu] planes with pixel index (b*Hu + y)*Wu + x; plus the decoder output."""
    sd = {k: _as_t(v) for k, v in sd.items()}
    f = _as_t(feat)
    b, c, h, w = f.shape
    hu, wu = int(size[0]), int(size[1])
    syn, idx_h, idx_w = make_syn_inp(b, h, w, hu, wu)
    u = unfold3x3(f)
    x = u[:, :, torch.from_numpy(idx_h.astype(np.int64))][:, :, :, torch.from_numpy(idx_w.astype(np.int64))]
    n = b * hu * wu
    acts = torch.empty((4, 2, HIDDEN, n))

    def plane(t):
        return t.permute(1, 0, 2, 3).reshape(HIDDEN, n)

    k = torch.relu(_conv1x1(x, sd["K.0.0.weight"], sd["K.0.0.bias"]))
    s = _conv1x1(syn, sd["Q.0.0.weight"], sd["Q.0.0.bias"])
    acts[0, 0], acts[0, 1] = plane(k), plane(s)
    q = k * torch.sin(s)
    for i in range(1, 4):
        k = torch.relu(_conv1x1(torch.cat([q, x], dim=1), sd[f"K.{i}.0.weight"], sd[f"K.{i}.0.bias"]))
        s = _conv1x1(q, sd[f"Q.{i}.0.weight"], sd[f"Q.{i}.0.bias"])
        acts[i, 0], acts[i, 1] = plane(k), plane(s)
        q = k * torch.sin(s)
    return _conv1x1(q, sd["last_layer.weight"], sd["last_layer.bias"]), acts
