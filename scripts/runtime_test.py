#!/usr/bin/env python3
"""runtime_test.py on the MI355X path: forward time of bicubic, MetaSR, LIIF and DIINN on a 48x48 input at
output sizes 48*s, s in {2, 3, 4, 6, 8} -- the reference's runtime comparison (runtime_test.py:8-62; its
``IMSISR(3, False)`` is the old name of ``DIINN(mode=3, init_q=False)``).

The reference times whole models (encoder included) with torch.utils.benchmark and leaves autograd on;
inference on the HIP path runs under torch.no_grad().  ``--lr N`` changes the input size, ``--graphs`` replays
the three networks from hipGraphs (a 48x48 input is launch-bound in the encoder)."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.utils.benchmark as benchmark  # noqa: E402

from diinn_amd.modules import BICUBIC_NET, DIINN, LIIF, MetaSR  # noqa: E402


def main():
    ap = ArgumentParser()
    ap.add_argument("--lr", type=int, default=48)
    ap.add_argument("--scales", type=int, nargs="+", default=[2, 3, 4, 6, 8])
    ap.add_argument("--runs", type=int, default=100)
    ap.add_argument("--graphs", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    models = {
        "bicubic": BICUBIC_NET().to(dev),
        "metasr": MetaSR(graphs=args.graphs).to(dev).eval(),
        "liif": LIIF(graphs=args.graphs).to(dev).eval(),
        "diinn": DIINN(3, False, graphs=args.graphs).to(dev).eval(),
    }
    x = torch.rand(1, 3, args.lr, args.lr, device=dev)
    for s in args.scales:
        size = args.lr * s
        for name, m in models.items():
            def run(m=m):
                with torch.no_grad():
                    return m(x, (size, size))
            run()                                            # warm-up (MIOpen find, weight packing, graph capture)
            t = benchmark.Timer(stmt="run()", globals={"run": run}, num_threads=1).timeit(args.runs)
            print(f"{size:5d} {name:8s} {t.mean * 1e3:9.3f} ms", flush=True)


if __name__ == "__main__":
    main()
