#!/usr/bin/env python3
"""Golden fixtures for the LIIF comparison decoder, from the REAL reference (build container only).

Runs ``LIIF.query_rgb`` (src/models/components/liif.py:59-127) on synthetic features with synthetic
``imnet`` weights regenerated from ``synth.py`` (inputs are never stored), a full ``forward`` through the
RDN encoder on a tiny image, and pulls the per-axis index / relative-coordinate tables out of the
reference's own grid_sample calls.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_liif.py
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import diinn_amd.synth as synth  # noqa: E402
from src.models.components.liif import LIIF  # noqa: E402  (the reference)

# (name, B, H, W, Hu, Wu, gain)
CASES = [
    ("liif_24x20_x3", 1, 24, 20, 72, 60, 1.0),
    ("liif_nonint_17x23_50x71", 2, 17, 23, 50, 71, 1.0),
    ("liif_x4_32x32_stress", 1, 32, 32, 128, 128, 2.0),
    ("liif_down_12x9_to_7x20", 1, 12, 9, 7, 20, 1.0),
]
TABLE_PAIRS = [(24, 72), (20, 60), (17, 50), (23, 71), (32, 128), (12, 7), (9, 20), (48, 96), (256, 1024), (64, 200), (5, 5), (3, 97)]


def imnet_state(model, seed, gain):
    shapes = {k: list(v.shape) for k, v in model.state_dict().items() if k.startswith("imnet.")}
    return synth.state_dict_for(shapes, seed, "liif.", gain=gain)


def reference_axis_tables(model, n_in, n_out, v):
    """idx / rel of one axis pulled from the reference's grid_sample + feat_coord arithmetic (liif.py:82-104)."""
    coord = model.make_coord((n_out, 1), "cpu")[:, 0]                       # [n_out] axis-0 coordinates
    feat_coord = model.make_coord((n_in, 1), "cpu", flatten=False)[:, 0, 0]  # [n_in]
    c_ = coord.clone()
    c_ += v * (2 / n_in / 2) + 1e-6
    c_.clamp_(-1 + 1e-6, 1 - 1e-6)
    grid = torch.stack([torch.zeros_like(c_), c_], dim=-1).view(1, 1, n_out, 2)      # (x, y) order of grid_sample
    src = torch.arange(n_in, dtype=torch.float32).view(1, 1, n_in, 1)
    idx = F.grid_sample(src, grid, mode="nearest", align_corners=False)[0, 0, 0].numpy().astype(np.int32)
    q = F.grid_sample(feat_coord.view(1, 1, n_in, 1), grid, mode="nearest", align_corners=False)[0, 0, 0]
    rel = (coord - q) * n_in
    return idx, rel.numpy().astype(np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    model = LIIF().eval()
    with torch.no_grad():
        for name, b, h, w, hu, wu, gain in CASES:
            sd = imnet_state(model, 123, gain)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
            feat = torch.from_numpy(synth.encoder_features(123, b, h, w))
            coord, cell = model.make_coord_and_cell(feat, (hu, wu))
            y = model.reshape_pred(model.query_rgb(feat, coord, cell), (hu, wu))
            y2 = model.reshape_pred(model.batched_predict(feat, coord, cell, 777), (hu, wu))
            assert (y - y2).abs().max() < 1e-6
            out[f"out/{name}"] = y.numpy().astype(np.float32)
            out[f"meta/{name}"] = np.array([b, h, w, hu, wu, gain], dtype=np.float64)
            print(name, tuple(y.shape), "max|y|=%.4f" % float(y.abs().max()))
        for n_in, n_out in TABLE_PAIRS:
            for v in (-1, 1):
                idx, rel = reference_axis_tables(model, n_in, n_out, v)
                out[f"idx/{n_in}_{n_out}_{v}"] = idx
                out[f"rel/{n_in}_{n_out}_{v}"] = rel
        # full model: key names/shapes and an end-to-end forward through the RDN encoder
        full = {k: list(v.shape) for k, v in model.state_dict().items()}
        out["liif/shapes_json"] = np.array(json.dumps(full))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(full, 123, "liifnet.").items()})
        img = torch.from_numpy(synth.uniform(123, "img:1x3x12x10", (1, 3, 12, 10), 0.5) + np.float32(0.5))
        out["liif/out_1x3x12x10_to_31x27"] = model(img, [31, 27], 300).numpy()
    np.savez(os.path.join(HERE, "liif_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "liif_golden.npz"))


if __name__ == "__main__":
    main()
