"""CPU experiment: what would storing the P image in HBM as bf16 cost in accuracy?
The oracle's hoisted form emulates the bf16_full arithmetic with and without a bf16-rounded P; errors are relative
to max|reference| (the restated bf16_full bound is 3e-3).  usage: python tools/bf16_p_storage_error.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import diinn_oracle as orc          # noqa: E402
import diinn_amd.synth as synth     # noqa: E402

cases = [("48x48 x2", 1, 48, 48, 96, 96), ("40x56 x3.3", 1, 40, 56, 132, 185), ("64x64 x4", 1, 64, 64, 256, 256),
         ("37x53 -> 120x171", 1, 37, 53, 120, 171), ("c5 crop 45x80 x3.3", 1, 45, 80, 148, 264)]
print(f"{'case':22s} {'seed':>4s} {'bf16_full':>10s} {'+bf16 P':>10s}   (max|err| / max|ref|; bound 3e-3)")
for name, b, h, w, hu, wu in cases:
    for seed in (123, 7):
        sd = synth.decoder_state_dict(seed)
        feat = synth.encoder_features(seed, b, h, w)
        ref = orc.decode_reference_form(sd, feat, (hu, wu), 30000).numpy()
        scale = float(np.abs(ref).max())
        a = orc.decode_hoisted_form(sd, feat, (hu, wu), bf16_operands=True, bf16_p=True).numpy()
        c = orc.decode_hoisted_form(sd, feat, (hu, wu), bf16_operands=True, bf16_p=True, bf16_p_storage=True).numpy()
        print(f"{name:22s} {seed:4d} {np.abs(a - ref).max() / scale:10.2e} {np.abs(c - ref).max() / scale:10.2e}")
