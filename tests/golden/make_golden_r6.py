#!/usr/bin/env python3
"""Round-6 golden vectors ("encoder parity off default init is one map, two gains").

Runs only in the build container (needs /root/reference, read-only; never on the GPU box).

The reference's RDN encoder (src/models/components/rdn.py:37-105) with PER-LAYER gains -- every layer's weight and bias scaled
by its own factor 2^u, u ~ U(-0.6, 1.0), keyed by the layer's name (synth.layer_gain): what a trained network looks like to
the rounding of a Winograd transform, which a uniformly scaled default init is not -- on one map per kernel family of the trunk:
    1 x 256 x 256   F(4x4,3x3), exactly one round of work items (no split)
    1 x 320 x 180   F(4x4,3x3), one round and a remainder: the last round split over the input channels
    1 x 100 x 120   F(2x2,3x3)
    2 x  48 x  48   the split-K kernel (the reference's own timing protocol's map, a batch of two)
fp32 samples (the parity contract: 2e-5 x max|ref|) and the same model in float64 (the truth), max|feat|, per gain seed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r6.py      (several minutes of CPU time)
"""
import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import diinn_amd.synth as synth  # noqa: E402
from src.models.components.rdn import make_rdn as ref_make_rdn  # noqa: E402  (the reference)

NSAMPLES = 16384
CASES = [(77, 1, 256, 256), (77, 1, 320, 180), (78, 1, 100, 120), (78, 2, 48, 48)]


def main():
    out = {}
    with torch.no_grad():
        enc = ref_make_rdn().eval()
        shapes = {k: list(v.shape) for k, v in enc.state_dict().items()}
        out["rdn/shapes_json"] = np.array(json.dumps(shapes))
        for (gseed, b, h, w) in CASES:
            enc.float()
            sd = synth.state_dict_for(shapes, 123, "enc.", layer_gain_seed=gseed)
            enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            img = torch.from_numpy(synth.uniform(7, f"img:{b}x{h}x{w}", (b, 3, h, w), 0.5) + np.float32(0.5))
            t0 = time.time()
            f = enc(img).numpy()
            f64 = enc.double()(img.double()).numpy()
            key = f"rdn_layer_gain/{gseed}/{b}x{h}x{w}"
            idx = np.random.default_rng(1000 * h + w).choice(f.size, size=min(NSAMPLES, f.size), replace=False).astype(np.int64)
            out[f"{key}/values"] = f.reshape(-1)[idx].astype(np.float32)
            out[f"{key}/values64"] = f64.reshape(-1)[idx].astype(np.float64)
            out[f"{key}/absmax"] = np.float32(np.abs(f).max())
            gains = [synth.layer_gain(gseed, "enc." + k.rsplit(".", 1)[0]) for k in shapes if k.endswith(".weight")]
            print(f"{key}: {time.time() - t0:.1f} s; max|feat| {np.abs(f).max():.3f}, |ref32 - ref64| {np.abs(f - f64).max():.2e}; "
                  f"layer gains {min(gains):.2f} .. {max(gains):.2f}, product^(1/n) {np.exp(np.mean(np.log(gains))):.3f}", flush=True)
    np.savez(os.path.join(HERE, "diinn_golden_r6.npz"), **out)
    print("wrote", os.path.join(HERE, "diinn_golden_r6.npz"))


if __name__ == "__main__":
    main()
