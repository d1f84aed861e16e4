#!/usr/bin/env python3
"""Round-4 additions to the golden fixtures, from the REAL reference decoder (build container only; /root/reference is
read-only and never travels):

  d64/<case>     for every case of diinn_golden.npz: float32(ref64 - ref32), where ref64 is the reference module run in
                 float64 (``dec.double()``) on the same inputs and ref32 the committed fp32 output (regenerated here and
                 asserted bit-equal).  ref64 = ref32 + d64 to ~1e-15: the truth the fp32 reference itself is ~6e-8
                 away from, and the yardstick of the regression-level bounds in tests/test_gpu_parity.py.
  out/siren_*    reference outputs (fp32) with SIREN-range synthesis weights (synth.SIREN_Q_GAIN: Q.0 x 30, Q.1-3 x
                 sqrt 6): layer-0 sine arguments reach ~33 rad on the raw coordinates (diinn.py:61-62,134) while
                 |out| stays O(0.1-1); + their d64, and meta (b, h, w, hu, wu, gain, bsize, q_first, q_hidden).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r4.py
"""
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

import diinn_amd.synth as synth  # noqa: E402
from src.models.components.diinn import ImplicitDecoder  # noqa: E402  (the reference)

# (name, B, H, W, Hu, Wu, gain, bsize)
SIREN_CASES = [
    ("siren_c1_48x48_x2", 1, 48, 48, 96, 96, 1.0, 30000),
    ("siren_nonint_40x56_132x185", 1, 40, 56, 132, 185, 1.0, None),
    ("siren_gain2_batch2_24x20_x3", 2, 24, 20, 72, 60, 2.0, 30000),     # every tensor x2 on top: |out| ~ 1
]


def run(sd, feat, size, bsize):
    dec = ImplicitDecoder(mode=3, init_q=False)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    dec.eval()
    with torch.no_grad():
        y32 = dec(feat, list(size), bsize)
        y64 = dec.double()(feat.double(), list(size), bsize)
    return y32.numpy().astype(np.float32), y64.numpy()


def main():
    torch.set_num_threads(8)
    old = np.load(os.path.join(HERE, "diinn_golden.npz"))
    out = {}
    for k in old.files:
        if not k.startswith("meta/"):
            continue
        name = k[5:]
        b, h, w, hu, wu, gain, bs = old[k]
        b, h, w, hu, wu = int(b), int(h), int(w), int(hu), int(wu)
        sd = synth.decoder_state_dict(123, float(gain))
        feat = torch.from_numpy(synth.encoder_features(123, b, h, w))
        y32, y64 = run(sd, feat, (hu, wu), None if bs < 0 else int(bs))
        assert np.array_equal(y32, old[f"out/{name}"]), f"{name}: the reference no longer reproduces the committed fixture"
        d = (y64 - y32.astype(np.float64)).astype(np.float32)
        out[f"d64/{name}"] = d
        print(f"{name}: max|ref32 - ref64| = {np.abs(d).max():.3e}  max|ref| = {np.abs(y32).max():.4f}")
    for name, b, h, w, hu, wu, gain, bsize in SIREN_CASES:
        sd = synth.decoder_state_dict(123, gain, q_gain=synth.SIREN_Q_GAIN)
        feat = torch.from_numpy(synth.encoder_features(123, b, h, w))
        y32, y64 = run(sd, feat, (hu, wu), bsize)
        out[f"out/{name}"] = y32
        out[f"d64/{name}"] = (y64 - y32.astype(np.float64)).astype(np.float32)
        out[f"meta/{name}"] = np.array([b, h, w, hu, wu, gain, -1 if bsize is None else bsize, *synth.SIREN_Q_GAIN],
                                       dtype=np.float64)
        q0 = np.abs(sd["Q.0.0.weight"].reshape(256, 3)[:, :2]).sum(1).max()
        print(f"{name}: max|ref32 - ref64| = {np.abs(out[f'd64/{name}']).max():.3e}  max|ref| = {np.abs(y32).max():.4f}  "
              f"layer-0 sine arguments up to ~{q0:.1f} rad")
    np.savez(os.path.join(HERE, "diinn_golden_r4.npz"), **out)
    print("wrote", os.path.join(HERE, "diinn_golden_r4.npz"))


if __name__ == "__main__":
    main()
