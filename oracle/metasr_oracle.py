"""CPU ORACLE for the MetaSR comparison decoder.  TEST INFRASTRUCTURE ONLY.

Restates ``MetaSR.query_rgb`` / ``forward`` of /root/reference/src/models/components/metasr.py:70-104,119-135
with explicit per-axis tables instead of ``F.grid_sample``.  Only ``tests/`` may import it.

Parity status: PINNED by tests/golden/metasr_golden.npz (outputs and axis tables captured from the real
reference by tests/golden/make_golden_metasr.py; tests/test_metasr.py checks this file against them).

Per axis (n_in LR samples, n_out HR samples):
    c[j]   = fp32(fp32(2/n_out) * j) + fp32(-1 + 1/n_out)                     make_coord, metasr.py:42-57
    cell   = fp32(2/n_out)                                                    metasr.py:64-66
    c_     = c - cell/2                                                       metasr.py:80-82
    cq     = clamp(c_ + 1e-6, -1 + 1e-6, 1 - 1e-6)                            metasr.py:83
    idx    = nearbyint((cq + 1) * fp32(n_in/2) - 0.5)                         grid_sample nearest (ATen vectorised CPU)
    q      = (fp32(fp32(2/n_in) * idx) + fp32(-1 + 1/n_in)) - fp32(1/n_in)    feat_coord shifted to the cell's corner, :74-78
    rel    = (c_ - q) * fp32(n_in/2)                                          metasr.py:93-95
    r_rev  = fp32(2/Hu) * fp32(H/2)                                           metasr.py:97 (rows axis only)
Per pixel: w = imnet([rel_h, rel_w, r_rev]) viewed [576, 3]; rgb = unfold3x3(feat)[cell] . w    (:99-104)
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from diinn_oracle import axis_centres, unfold3x3


def metasr_axis_tables(n_in: int, n_out: int) -> Tuple[np.ndarray, np.ndarray]:
    c = axis_centres(n_out)
    cell = np.float32(2 / n_out)
    c_ = (c - cell / np.float32(2)).astype(np.float32)
    cq = np.clip((c_ + np.float32(1e-6)).astype(np.float32), np.float32(-1 + 1e-6), np.float32(1 - 1e-6)).astype(np.float32)
    x = (cq + np.float32(1)) * np.float32(np.float32(n_in) / np.float32(2)) - np.float32(0.5)
    idx = np.rint(x.astype(np.float32)).astype(np.int32)
    fc = (axis_centres(n_in) - np.float32((2 / n_in) / 2)).astype(np.float32)
    rel = ((c_ - fc[idx]).astype(np.float32) * np.float32(n_in / 2)).astype(np.float32)
    return idx, rel


def metasr_r_rev(h: int, hu: int) -> np.float32:
    return np.float32(np.float32(2 / hu) * np.float32(h / 2))


@torch.no_grad()
def metasr_query_reference_form(sd: Dict[str, np.ndarray], feat, size: Sequence[int]) -> torch.Tensor:
    w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()}
    feat = torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32))
    b, c, h, wd = feat.shape
    hu, wu = int(size[0]), int(size[1])
    ih, rh = metasr_axis_tables(h, hu)
    iw, rw = metasr_axis_tables(wd, wu)
    u = unfold3x3(feat)
    q = u[:, :, torch.from_numpy(ih.astype(np.int64))][:, :, :, torch.from_numpy(iw.astype(np.int64))]
    q = q.permute(0, 2, 3, 1).reshape(-1, 1, 576)
    inp = torch.empty((b, hu, wu, 3))
    inp[..., 0] = torch.from_numpy(rh)[None, :, None]
    inp[..., 1] = torch.from_numpy(rw)[None, None, :]
    inp[..., 2] = float(metasr_r_rev(h, hu))
    x = torch.relu(F.linear(inp.view(-1, 3), w["imnet.layers.0.weight"], w["imnet.layers.0.bias"]))
    x = F.linear(x, w["imnet.layers.2.weight"], w["imnet.layers.2.bias"]).view(-1, 576, 3)
    out = torch.bmm(q, x).view(b, hu, wu, 3)
    return out.permute(0, 3, 1, 2).contiguous()
