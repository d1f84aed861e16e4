#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference decoder.

Runs only in the build container (needs /root/reference, read-only; never on
the GPU box).  Imports ``ImplicitDecoder`` from the reference
(src/models/components/diinn.py:39-173), loads synthetic weights regenerated
from ``synth.py`` by parameter name, runs ``forward`` on synthetic features and
stores *outputs only* (plus the 1-D coordinate/index tables extracted from the
reference's own ``_make_pos_encoding`` / ``F.interpolate(nearest-exact)``).
Inputs are never stored: every consumer regenerates them from ``synth.py``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
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
import torch.nn.functional as F  # noqa: E402

import diinn_amd.synth as synth  # noqa: E402
from src.models.components.diinn import ImplicitDecoder  # noqa: E402  (the reference)

# (name, B, H, W, Hu, Wu, gain, bsize)   -- SURVEY.md §8(d3) parity shapes
CASES = [
    ("c1_48x48_x2", 1, 48, 48, 96, 96, 1.0, None),
    ("c1_48x48_x2_stress", 1, 48, 48, 96, 96, 3.0, 30000),
    ("nonint_40x56_132x185", 1, 40, 56, 132, 185, 1.0, 30000),
    ("x4_64x64", 1, 64, 64, 256, 256, 1.0, 30000),
    ("odd_37x53_120x171", 1, 37, 53, 120, 171, 1.0, None),
    ("batch2_24x20_x3", 2, 24, 20, 72, 60, 1.0, None),
    ("down_31x33_to_17x50", 1, 31, 33, 17, 50, 1.0, None),   # HR smaller than LR on one axis; small-output kernel
    ("tie_small_4x3_110x9", 1, 4, 3, 110, 9, 1.0, None),     # Hu+Wu<=128: ATen small-output index kernel, exact tie at row 27
    ("tie_generic_6x5_165x12", 1, 6, 5, 165, 12, 1.0, 30000),  # generic index kernel, exact tie at row 27
    ("tiny_1x1_to_5x7", 1, 1, 1, 5, 7, 1.0, None),
]

# axis pairs whose tables are pinned (includes every BASELINE config axis)
TABLE_PAIRS = [(48, 96), (40, 132), (56, 185), (64, 256), (37, 120), (53, 171), (24, 72), (20, 60),
               (31, 17), (33, 50), (256, 1024), (512, 2048), (1024, 4096), (1024, 8192),
               (720, 2376), (1280, 4224), (7, 1000), (1500, 8999), (3, 3), (1, 5), (5, 1),
               # exact-tie pairs ((j+0.5)*n_in/n_out integral) where the candidate fp32 formulas disagree
               (2, 97), (4, 110), (6, 165), (2, 1449), (2, 591), (10, 8165), (78, 7527), (300, 7050),
               (24, 660), (118, 2419), (16, 984)]


def reference_decoder(seed, gain):
    dec = ImplicitDecoder(mode=3, init_q=False)
    sd = synth.decoder_state_dict(seed=seed, gain=gain)
    missing = dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return dec.eval()


_DEC = None


def reference_axis_tables(n_in, n_out, other):
    """Pull the 1-D tables out of the reference's 2-D grid (diinn.py:94-110).
    ``other`` is the size of the second output axis: ATen switches CPU kernels
    (and index rounding) at out_H + out_W <= 128, so it is part of the key."""
    global _DEC
    if _DEC is None:
        _DEC = ImplicitDecoder(mode=3, init_q=False)
    x = torch.zeros(1, 64, n_in, 2)
    rel = _DEC._make_pos_encoding(x, (n_out, other))[0, 0, :, 0].numpy().copy()
    src = torch.arange(n_in, dtype=torch.float32).view(1, 1, n_in, 1).expand(1, 1, n_in, 2).contiguous()
    idx = F.interpolate(src, size=(n_out, other), mode="nearest-exact")[0, 0, :, 0].numpy().astype(np.int32)
    return idx, rel


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    with torch.no_grad():
        for name, b, h, w, hu, wu, gain, bsize in CASES:
            dec = reference_decoder(123, gain)
            feat = torch.from_numpy(synth.encoder_features(123, b, h, w))
            y = dec(feat, [hu, wu], bsize)
            y2 = dec(feat, torch.Size([hu, wu]), None if bsize else 30000)
            assert torch.equal(y, y2) or (y - y2).abs().max() < 1e-6, name
            out[f"out/{name}"] = y.numpy().astype(np.float32)
            out[f"meta/{name}"] = np.array([b, h, w, hu, wu, gain, -1 if bsize is None else bsize], dtype=np.float64)
            print(name, tuple(y.shape), "max|y|=%.4f" % float(y.abs().max()))
        # decoder modes 1 and 2 (ablation variants of the reference, diinn.py:116-131)
        for mode in (1, 2):
            dec = ImplicitDecoder(mode=mode, init_q=False)
            sdm = synth.decoder_state_dict(seed=123, mode=mode)
            dec.load_state_dict({k: torch.from_numpy(v) for k, v in sdm.items()}, strict=True)
            feat = torch.from_numpy(synth.encoder_features(123, 1, 24, 20))
            y = dec.eval()(feat, [79, 66], 30000)
            out[f"mode{mode}/out_24x20_79x66"] = y.numpy().astype(np.float32)
            print("mode", mode, tuple(y.shape), "max|y|=%.4f" % float(y.abs().max()))
        for n_in, n_out in TABLE_PAIRS:
            idx, rel = reference_axis_tables(n_in, n_out, 129)      # generic kernel (sum > 128)
            out[f"idx/generic/{n_in}_{n_out}"] = idx
            out[f"rel/generic/{n_in}_{n_out}"] = rel
            if n_out + 1 <= 128:
                idx, rel = reference_axis_tables(n_in, n_out, 1)    # small-output kernel
                out[f"idx/small/{n_in}_{n_out}"] = idx
                out[f"rel/small/{n_in}_{n_out}"] = rel
        # ratio constant (diinn.py:166) for a few shapes
        for (h, w, hu, wu) in [(48, 48, 96, 96), (40, 56, 132, 185), (720, 1280, 2376, 4224), (1024, 1024, 8192, 8192)]:
            r = torch.zeros(1).new_tensor([(h * w) / (hu * wu)]).numpy()
            out[f"ratio/{h}_{w}_{hu}_{wu}"] = r.astype(np.float32)
        # callers' side: RDN encoder key names/shapes and a DIINN end-to-end output
        import json
        from src.models.components.diinn import DIINN as RefDIINN
        from src.models.components.rdn import make_rdn as ref_make_rdn
        enc = ref_make_rdn()
        shapes = {k: list(v.shape) for k, v in enc.state_dict().items()}
        out["rdn/shapes_json"] = np.array(json.dumps(shapes))
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(shapes, 123, "enc.").items()})
        img = torch.from_numpy(synth.uniform(123, "img:1x3x12x10", (1, 3, 12, 10), 0.5) + np.float32(0.5))
        out["rdn/out_1x3x12x10"] = enc.eval()(img).numpy()
        net = RefDIINN(mode=3, init_q=False).eval()
        full = {k: list(v.shape) for k, v in net.state_dict().items()}
        out["diinn/shapes_json"] = np.array(json.dumps(full))
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(full, 123, "diinn.").items()})
        out["diinn/out_1x3x12x10_to_31x27"] = net(img, [31, 27], 30000).numpy()
        print("rdn + diinn e2e captured", len(shapes), len(full))
    np.savez(os.path.join(HERE, "diinn_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "diinn_golden.npz"))


if __name__ == "__main__":
    main()
