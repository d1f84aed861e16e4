"""CPU ORACLE for the LIIF comparison decoder.  TEST INFRASTRUCTURE ONLY.

Restates ``LIIF.query_rgb`` / ``forward`` of /root/reference/src/models/components/liif.py:59-127,148-155
(local_ensemble, feat_unfold and cell_decode all on -- the constructor defaults the reference uses) with
explicit per-axis tables instead of ``F.grid_sample``.  Only ``tests/`` may import it.

Parity status: PINNED by tests/golden/liif_golden.npz (outputs and axis tables captured from the real
reference by tests/golden/make_golden_liif.py; tests/test_liif.py checks this file against them).

Per axis (n_in LR samples, n_out HR samples) and ensemble shift v in {-1, +1}:
    c[j]   = fp32(fp32(2/n_out) * j) + fp32(-1 + 1/n_out)                     make_coord, liif.py:33-47
    c_     = clamp(c + fp32(v/n_in + 1e-6), -1 + 1e-6, 1 - 1e-6)              liif.py:91-93
    idx    = nearbyint((c_ + 1) * fp32(n_in/2) - 0.5)                         grid_sample nearest, align_corners=False
             (ATen's vectorised CPU kernel, GridSamplerKernel.cpp ComputeLocation::unnormalize; round half to even)
    q      = fp32(fp32(2/n_in) * idx) + fp32(-1 + 1/n_in)                     feat_coord, liif.py:82-84,98-101
    rel    = (c - q) * fp32(n_in)                                             liif.py:102-104
    rel_cell = fp32(2/n_out) * fp32(n_in)                                     liif.py:55-56,108-110
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from diinn_oracle import axis_centres, unfold3x3

HIDDEN = 256
EPS = 1e-6


def liif_axis_tables(n_in: int, n_out: int, v: int) -> Tuple[np.ndarray, np.ndarray]:
    """(idx int32 [n_out], rel fp32 [n_out]) of one axis for ensemble shift v = -1 or +1."""
    c = axis_centres(n_out)
    shift = np.float32(v * (2 / n_in / 2) + EPS)
    c_ = np.clip(c + shift, np.float32(-1 + 1e-6), np.float32(1 - 1e-6)).astype(np.float32)
    x = (c_ + np.float32(1)) * np.float32(np.float32(n_in) / np.float32(2)) - np.float32(0.5)
    idx = np.rint(x.astype(np.float32)).astype(np.int32)
    q = axis_centres(n_in)[idx]
    rel = ((c - q).astype(np.float32) * np.float32(n_in)).astype(np.float32)
    return idx, rel


def liif_rel_cell(n_in: int, n_out: int) -> np.float32:
    return np.float32(np.float32(2 / n_out) * np.float32(n_in))


@torch.no_grad()
def liif_query_reference_form(sd: Dict[str, np.ndarray], feat, size: Sequence[int]) -> torch.Tensor:
    """[B,3,Hu,Wu]: gather the unfolded features of the 4 shifted nearest cells, run the 580->256^4->3 ReLU
    MLP on each, blend by the diagonally swapped areas (liif.py:59-127)."""
    w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()}
    feat = torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32))
    b, c, h, wd = feat.shape
    hu, wu = int(size[0]), int(size[1])
    u = unfold3x3(feat)                                           # [B,576,H,W]
    cell_h, cell_w = float(liif_rel_cell(h, hu)), float(liif_rel_cell(wd, wu))
    preds, areas = [], []
    for vx in (-1, 1):
        ih, rh = liif_axis_tables(h, hu, vx)
        for vy in (-1, 1):
            iw, rw = liif_axis_tables(wd, wu, vy)
            q = u[:, :, torch.from_numpy(ih.astype(np.int64))][:, :, :, torch.from_numpy(iw.astype(np.int64))]
            q = q.permute(0, 2, 3, 1)                             # [B,Hu,Wu,576]
            rel = torch.empty((b, hu, wu, 4))
            rel[..., 0] = torch.from_numpy(rh)[None, :, None]
            rel[..., 1] = torch.from_numpy(rw)[None, None, :]
            rel[..., 2] = cell_h
            rel[..., 3] = cell_w
            x = torch.cat([q, rel], dim=-1).reshape(-1, 580)
            for i in (0, 2, 4, 6):
                x = torch.relu(F.linear(x, w[f"imnet.layers.{i}.weight"], w[f"imnet.layers.{i}.bias"]))
            x = F.linear(x, w["imnet.layers.8.weight"], w["imnet.layers.8.bias"])
            preds.append(x.view(b, hu, wu, 3))
            areas.append((rel[..., 0] * rel[..., 1]).abs() + 1e-9)
    tot = torch.stack(areas).sum(dim=0)
    areas = [areas[3], areas[2], areas[1], areas[0]]              # diagonal swap, liif.py:121-123
    out = 0
    for pred, area in zip(preds, areas):
        out = out + pred * (area / tot).unsqueeze(-1)
    return out.permute(0, 3, 1, 2).contiguous()
