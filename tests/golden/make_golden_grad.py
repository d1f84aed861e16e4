#!/usr/bin/env python3
"""Generate gradient fixtures from the REAL reference decoder under autograd.

Build container only (needs /root/reference, read-only).  Runs the reference
``ImplicitDecoder(mode=3).forward(x, size, None)`` with autograd on -- the training path of
sr_module.py:127-129 -- for synthetic weights/features regenerated from ``synth.py``, with the scalar
loss  sum(out * R)  (R from synth.py, so d loss / d out = R), and stores d loss / d (features and every
parameter).  The three [256,832] K weights are stored for every 8th output row (all columns), the rest
in full.  Inputs are never stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grad.py
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

# (name, B, H, W, Hu, Wu, gain)
CASES = [
    ("grad_b2_12x10_31x27", 2, 12, 10, 31, 27, 1.0),
    ("grad_b1_9x14_36x56_stress", 1, 9, 14, 36, 56, 3.0),
]
ROW_STRIDE = 8


def main():
    torch.manual_seed(0)
    out = {}
    for name, b, h, w, hu, wu, gain in CASES:
        dec = ImplicitDecoder(mode=3, init_q=False)
        sd = synth.decoder_state_dict(seed=123, gain=gain)
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        dec.train()
        feat = torch.from_numpy(synth.encoder_features(123, b, h, w)).requires_grad_(True)
        r = torch.from_numpy(synth.uniform(123, f"gradw:{name}", (b, 3, hu, wu), 1.0))
        y = dec(feat, [hu, wu], None)
        (y * r).sum().backward()
        out[f"meta/{name}"] = np.array([b, h, w, hu, wu, gain], dtype=np.float64)
        out[f"out/{name}"] = y.detach().numpy().astype(np.float32)
        out[f"grad/{name}/feat"] = feat.grad.numpy().astype(np.float32)
        for pname, p in dec.named_parameters():
            g = p.grad.numpy().astype(np.float32)
            if pname.startswith("K.") and pname.endswith("weight"):
                g = g[::ROW_STRIDE]
            out[f"grad/{name}/{pname}"] = g
        print(name, "max|dfeat|=%.4f" % float(feat.grad.abs().max()),
              "max|dK3|=%.4f" % float(dec.K[3][0].weight.grad.abs().max()))
    np.savez(os.path.join(HERE, "diinn_golden_grad.npz"), **out)
    print("wrote", os.path.join(HERE, "diinn_golden_grad.npz"))


if __name__ == "__main__":
    main()
