#!/usr/bin/env python3
"""Soak of the small-map kernels (conv_t16_kernel / conv1x1_t16_kernel): the RDN trunk on several small maps, forward after forward on
alternating inputs, each result compared bit for bit with the first result for that input -- alone and beside a stream that keeps the
memory system busy (a timing-dependent hazard between the LDS-DMA stages and their readers would show as a differing output).
usage: t16_soak.py [FORWARDS=3000]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diinn_amd.modules as M  # noqa: E402
from diinn_amd import _native  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = M.make_rdn().to(dev).eval()
    lib = _native.load()
    shapes = [(1, 48, 48), (1, 32, 32), (2, 24, 40), (1, 17, 52), (4, 24, 24)]
    inputs = {s: [torch.rand(s[0], 3, s[1], s[2], device=dev) for _ in range(3)] for s in shapes}
    for s in shapes:
        assert lib.diinn_conv_t16_applies(*s) == 1, s
    side = torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    launches = 0
    with torch.no_grad():
        want = {s: [enc(x).clone() for x in xs] for s, xs in inputs.items()}
        for load in (False, True):
            bad = 0
            for i in range(n):
                s = shapes[i % len(shapes)]
                k = (i // len(shapes)) % 3
                if load and i % 4 == 0:
                    with torch.cuda.stream(side):
                        big[: 128 << 20].copy_(big[128 << 20:])      # 128 MB through HBM beside the trunk
                got = enc(inputs[s][k])
                bad += int(not torch.equal(got, want[s][k]))
                launches += 146
            torch.cuda.synchronize()
            print(f"{n} forwards ({n * 146:,} launches of the two kernels + the global fusion's) {'beside a copy stream' if load else 'alone'}: {bad} differing outputs", flush=True)
            assert bad == 0
    print("ok")


if __name__ == "__main__":
    main()
