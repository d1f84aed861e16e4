#!/usr/bin/env python3
"""Golden vectors for the RDN encoder on maps big enough to take the Winograd kernels.

Runs only in the build container (needs /root/reference, read-only; never on the GPU box).  Imports the reference's
``make_rdn`` (src/models/components/rdn.py:108-116), loads synthetic weights regenerated from ``synth.py`` by parameter
name, runs the encoder on the CPU on synthetic images and stores OUTPUT SAMPLES only: the values at 16,384 seeded
random positions plus per-channel sums (whole outputs would be 2.5 and 15.7 MB).  Inputs are never stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rdn_big.py      (about a minute of CPU time)
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

# (B, H, W): 9,600 pixels -> Winograd 3x3 layers with one output half per workgroup; 61,440 -> both halves per
# workgroup and the streaming 1x1 kernel; 2 x 50 x 90 -> batch > 1, odd tile counts
CASES = [(1, 96, 100), (1, 240, 256), (2, 50, 90)]
NSAMPLES = 16384


def sample_index(n, seed):
    return np.random.default_rng(seed).choice(n, size=min(NSAMPLES, n), replace=False).astype(np.int64)


def main():
    torch.manual_seed(0)
    enc = ref_make_rdn().eval()
    shapes = {k: list(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(shapes, 123, "enc.").items()})
    out = {"rdn/shapes_json": np.array(json.dumps(shapes))}
    with torch.no_grad():
        for (b, h, w) in CASES:
            img = torch.from_numpy(synth.uniform(7, f"img:{b}x{h}x{w}", (b, 3, h, w), 0.5) + np.float32(0.5))
            t0 = time.time()
            y = enc(img).numpy()
            idx = sample_index(y.size, 1000 * h + w)
            key = f"{b}x{h}x{w}"
            out[f"rdn/{key}/values"] = y.reshape(-1)[idx].astype(np.float32)
            out[f"rdn/{key}/channel_sums"] = y.astype(np.float64).sum(axis=(0, 2, 3))
            out[f"rdn/{key}/absmax"] = np.float32(np.abs(y).max())
            print(f"{key}: {time.time() - t0:.1f} s on the CPU, |y|max {np.abs(y).max():.3f}")
    np.savez(os.path.join(HERE, "rdn_big_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "rdn_big_golden.npz"))


if __name__ == "__main__":
    main()
