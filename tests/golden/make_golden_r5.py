#!/usr/bin/env python3
"""Round-5 golden vectors: the reference on maps that take the encoder's DEFAULT arithmetic.

Runs only in the build container (needs /root/reference, read-only; never on the GPU box).

  (a) ``diinn/200x180_431x377``: the real ``DIINN(mode=3, init_q=False)`` (src/models/components/diinn.py:8-19: RDN
      encoder + implicit decoder) end to end on a 200 x 180 image -> 431 x 377 -- a map whose 3x3 encoder layers run
      Winograd F(4x4,3x3) here (1,500 work items of... 2 x 36 blocks; the last round split over the input channels),
      so one assertion says "DIINN.forward at >= 192^2 matches the reference's output".  Stored: 16,384 sampled
      outputs in fp32 AND from the same model run in float64 (the truth the reference itself is some distance from),
      per-channel sums, max |out|.
  (b) ``rdn_gain/<g>/1x192x200``: the reference's encoder with every weight and bias scaled by g = 1.5 and 2.0 (per-layer
      gain off the default init: max|feat| 1.5 -> 3.8 -> 66): how the F(4x4) error grows off default init.  fp32 and float64
      samples, max |feat|.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r5.py      (a few minutes of CPU time)
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
from src.models.components.diinn import DIINN as RefDIINN  # noqa: E402  (the reference)
from src.models.components.rdn import make_rdn as ref_make_rdn  # noqa: E402

NSAMPLES = 16384
E2E = (1, 200, 180, 431, 377)
GAIN_CASES = [(1.5, 1, 192, 200), (2.0, 1, 192, 200)]


def sample_index(n, seed):
    return np.random.default_rng(seed).choice(n, size=min(NSAMPLES, n), replace=False).astype(np.int64)


def main():
    torch.manual_seed(0)
    out = {}
    with torch.no_grad():
        b, h, w, hu, wu = E2E
        net = RefDIINN(mode=3, init_q=False).eval()
        full = {k: list(v.shape) for k, v in net.state_dict().items()}
        out["diinn/shapes_json"] = np.array(json.dumps(full))
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(full, 123, "diinn.").items()})
        img = torch.from_numpy(synth.uniform(11, f"img:{b}x{h}x{w}", (b, 3, h, w), 0.5) + np.float32(0.5))
        t0 = time.time()
        y = net(img, [hu, wu], 30000).numpy()
        t1 = time.time()
        y64 = net.double()(img.double(), [hu, wu], 30000).numpy()
        key = f"diinn/{h}x{w}_{hu}x{wu}"
        idx = sample_index(y.size, 1000 * hu + wu)
        out[f"{key}/values"] = y.reshape(-1)[idx].astype(np.float32)
        out[f"{key}/values64"] = y64.reshape(-1)[idx].astype(np.float64)
        out[f"{key}/channel_sums"] = y.astype(np.float64).sum(axis=(0, 2, 3))
        out[f"{key}/absmax"] = np.float32(np.abs(y).max())
        print(f"{key}: fp32 {t1 - t0:.1f} s, float64 {time.time() - t1:.1f} s; max|y| {np.abs(y).max():.4f}, "
              f"|ref32 - ref64| {np.abs(y - y64).max():.2e}")

        enc = ref_make_rdn().eval()
        shapes = {k: list(v.shape) for k, v in enc.state_dict().items()}
        out["rdn/shapes_json"] = np.array(json.dumps(shapes))
        for (gain, b, h, w) in GAIN_CASES:
            enc.float()
            enc.load_state_dict({k: torch.from_numpy(v)
                                 for k, v in synth.state_dict_for(shapes, 123, "enc.", gain=gain).items()})
            img = torch.from_numpy(synth.uniform(7, f"img:{b}x{h}x{w}", (b, 3, h, w), 0.5) + np.float32(0.5))
            t0 = time.time()
            f = enc(img).numpy()
            f64 = enc.double()(img.double()).numpy()
            key = f"rdn_gain/{gain}/{b}x{h}x{w}"
            idx = sample_index(f.size, 1000 * h + w)
            out[f"{key}/values"] = f.reshape(-1)[idx].astype(np.float32)
            out[f"{key}/values64"] = f64.reshape(-1)[idx].astype(np.float64)
            out[f"{key}/absmax"] = np.float32(np.abs(f).max())
            print(f"{key}: {time.time() - t0:.1f} s; max|feat| {np.abs(f).max():.3f}, |ref32 - ref64| {np.abs(f - f64).max():.2e}")
    np.savez(os.path.join(HERE, "diinn_golden_r5.npz"), **out)
    print("wrote", os.path.join(HERE, "diinn_golden_r5.npz"))


if __name__ == "__main__":
    main()
