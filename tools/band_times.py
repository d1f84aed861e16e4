#!/usr/bin/env python3
"""Per-rank compute time of the row-band sharding on ONE GPU: for each BASELINE multi-GPU config the time of the
slowest band (precompute_P on the band's LR rows + decode of the band), i.e. what every rank would spend per
decode if it held its feature rows.  With the measured whole-image time it gives the compute-side speed-up the
band partition allows (the feature hand-off is extra)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diinn_amd.decoder as D  # noqa: E402
import diinn_amd.sharded as S  # noqa: E402
import diinn_amd.synth as synth  # noqa: E402


def t_ms(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
    for name, lr, sc, worlds in (("c3 512x512 x4", 512, 4, (1, 2, 4)), ("c4 1024x1024 x8", 1024, 8, (1, 8))):
        hu = wu = lr * sc
        feat = torch.randn(1, 64, lr, lr, device=dev)
        ws = torch.empty(lr * lr * 1024, device=dev)
        out = torch.empty(1, 3, hu, wu, device=dev)
        base = None
        for world in worlds:
            worst = 0.0
            for rank in sorted({0, world // 2, world - 1}):
                y0, y1 = S.band_for_rank(hu, rank, world)
                worst = max(worst, t_ms(lambda: D.decode_features(feat, packed, (hu, wu), out=out, workspace=ws, rows=(y0, y1)),
                                        n=3 if hu * wu > 3e7 else 5))
            base = base or worst
            print(f"{name}: {world} band(s): slowest band {worst:9.3f} ms  -> compute-side speed-up {base / worst:5.2f}x "
                  f"({hu * wu / worst / 1e3:8.1f} Mpix/s aggregate)", flush=True)
        del feat, ws, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
