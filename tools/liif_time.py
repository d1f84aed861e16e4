#!/usr/bin/env python3
"""LIIF decoder timing on one GPU: HIP path (precompute_P + liif_kernel) vs the reference's op sequence
(unfold, 4 x grid_sample nearest, 580->256^4->3 MLP, area blend; liif.py:59-127) in PyTorch-ROCm eager mode,
query chunks of 30000 like the reference's evaluation (sr_module.py:85).  usage: liif_time.py [LR] [SCALE]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diinn_amd.decoder as D  # noqa: E402
import diinn_amd.modules as M  # noqa: E402
import diinn_amd.synth as synth  # noqa: E402


def make_coord(shape, dev):
    seqs = [(-1 + 1 / n) + (2 / n) * torch.arange(n, device=dev).float() for n in shape]
    return torch.stack(torch.meshgrid(*seqs, indexing="ij"), dim=-1)


def eager_query(imnet, feat, coord, cell):
    u = F.unfold(feat, 3, padding=1).view(feat.shape[0], feat.shape[1] * 9, feat.shape[2], feat.shape[3])
    h, w = feat.shape[-2:]
    fc = make_coord((h, w), feat.device).permute(2, 0, 1).unsqueeze(0)
    preds, areas = [], []
    for vx in (-1, 1):
        for vy in (-1, 1):
            c_ = coord.clone()
            c_[:, :, 0] += vx / h + 1e-6
            c_[:, :, 1] += vy / w + 1e-6
            c_.clamp_(-1 + 1e-6, 1 - 1e-6)
            g = c_.flip(-1).unsqueeze(1)
            qf = F.grid_sample(u, g, mode="nearest", align_corners=False)[:, :, 0, :].permute(0, 2, 1)
            qc = F.grid_sample(fc, g, mode="nearest", align_corners=False)[:, :, 0, :].permute(0, 2, 1)
            rel = coord - qc
            rel[:, :, 0] *= h
            rel[:, :, 1] *= w
            rc = cell.clone()
            rc[:, :, 0] *= h
            rc[:, :, 1] *= w
            preds.append(imnet(torch.cat([qf, rel, rc], dim=-1)))
            areas.append((rel[:, :, 0] * rel[:, :, 1]).abs() + 1e-9)
    tot = torch.stack(areas).sum(0)
    areas = areas[::-1]
    return sum(p * (a / tot).unsqueeze(-1) for p, a in zip(preds, areas))


def main():
    only_ours = "--only-ours" in sys.argv             # for profiling: skip the eager comparison
    argv = [a for a in sys.argv if not a.startswith("--")]
    lr = int(argv[1]) if len(argv) > 1 else 256
    sc = int(argv[2]) if len(argv) > 2 else 4
    dev = torch.device("cuda:0")
    hu = wu = lr * sc
    net = M.LIIF().to(dev).eval()
    feat = torch.from_numpy(synth.encoder_features(123, 1, lr, lr)).to(dev)
    packed = D.pack_liif_state_dict(net.imnet.state_dict(), prefix="").to(dev)
    ws = torch.empty(lr * lr * 1024, device=dev)
    out = torch.empty((1, 3, hu, wu), device=dev)

    def ours():
        D.liif_decode_features(feat, packed, (hu, wu), out=out, workspace=ws)

    if only_ours:
        for _ in range(5):
            ours()
        torch.cuda.synchronize()
        return
    coord = make_coord((hu, wu), dev).view(1, -1, 2)
    cell = torch.ones_like(coord)
    cell[:, :, 0] *= 2 / hu
    cell[:, :, 1] *= 2 / wu

    @torch.no_grad()
    def eager():
        preds = []
        for ql in range(0, coord.shape[1], 30000):
            preds.append(eager_query(net.imnet, feat, coord[:, ql:ql + 30000], cell[:, ql:ql + 30000]))
        return torch.cat(preds, dim=1)

    with torch.no_grad():
        ref = eager().view(1, hu, wu, 3).permute(0, 3, 1, 2)
        ours()
        err = float((out - ref).abs().max())
    for name, fn, n in (("HIP path", ours, 10), ("eager ops", eager, 2)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"LIIF decode {lr}x{lr} x{sc} ({hu * wu} px) {name:10s}: {ms:9.2f} ms  {hu * wu / ms / 1e3:8.2f} Mpix/s")
    print(f"max|HIP - eager| = {err:.2e}")


if __name__ == "__main__":
    main()
