#!/usr/bin/env python3
"""RDN encoder forward: HIP trunk (conv_ksplit_kernel, C ABI diinn_rdn_forward) vs PyTorch-ROCm/MIOpen, eager and
hipGraph-replayed, over input sizes -- locates the cross-over behind RDN.hip_trunk_max_pixels.
usage: enc_trunk_time.py [SIZE ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diinn_amd.modules as M  # noqa: E402


def t_ms(fn, n=10):
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


def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def main():
    dev = torch.device("cuda:0")
    only_hip = "--only-hip" in sys.argv                      # for profiling: HIP trunk alone, eager
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [24, 48, 64, 96, 128, 192, 256]
    enc = M.make_rdn().to(dev).eval()
    print(f"{'LR':>9s} {'MIOpen ms':>10s} {'MIOpen graph':>13s} {'HIP trunk ms':>13s} {'HIP graph':>10s} {'max diff':>9s}")
    with torch.no_grad():
        for lr in sizes:
            x = torch.rand(1, 3, lr, lr, device=dev)
            if only_hip:
                enc.hip_trunk_max_pixels = 1 << 30
                print(f"{lr:4d}x{lr:<4d} HIP trunk {t_ms(lambda: enc(x)):10.3f} ms", flush=True)
                continue
            enc.hip_trunk_max_pixels = None
            ref = enc(x)
            t_mi = t_ms(lambda: enc(x))
            t_mig = t_ms(graphed(lambda: enc(x)))
            enc.hip_trunk_max_pixels = 1 << 30
            got = enc(x)
            t_hip = t_ms(lambda: enc(x))
            t_hipg = t_ms(graphed(lambda: enc(x)))
            print(f"{lr:4d}x{lr:<4d} {t_mi:10.3f} {t_mig:13.3f} {t_hip:13.3f} {t_hipg:10.3f} {float((got - ref).abs().max()):9.1e}", flush=True)


if __name__ == "__main__":
    main()
