#!/usr/bin/env python3
"""RDN encoder forward time under MIOpen's default (immediate-mode) solver choice vs an exhaustive find
(torch.backends.cudnn.benchmark = True), and channels-last.  usage: enc_find_time.py [LR ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diinn_amd.modules as M  # noqa: E402


def t_ms(fn, n=5):
    for _ in range(3):
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
    sizes = [int(a) for a in sys.argv[1:]] or [48, 256]
    enc = M.make_rdn().to(dev).eval()
    for lr in sizes:
        x = torch.rand(1, 3, lr, lr, device=dev)
        res = {}
        with torch.no_grad():
            torch.backends.cudnn.benchmark = False
            res["default"] = t_ms(lambda: enc(x))
            torch.backends.cudnn.benchmark = True
            res["benchmark=True"] = t_ms(lambda: enc(x))
            enc_cl = enc.to(memory_format=torch.channels_last)
            xc = x.contiguous(memory_format=torch.channels_last)
            res["benchmark + channels_last"] = t_ms(lambda: enc_cl(xc))
            enc.to(memory_format=torch.contiguous_format)
        print(f"LR {lr}x{lr}: " + ", ".join(f"{k} {v:.2f} ms" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
