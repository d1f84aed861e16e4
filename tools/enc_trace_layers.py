#!/usr/bin/env python3
"""Per-layer durations of the RDN trunk from a rocprofv3 --kernel-trace CSV of tools/enc_trunk_time.py --only-hip:
the trunk is 147 conv_ksplit / conv_wino launches in a fixed order, so launch index mod 147 names the layer.
usage: enc_trace_layers.py kernel_trace.csv [H W]"""
import csv
import sys
from collections import defaultdict

PEAK = 157.3e12


def main():
    path = sys.argv[1]
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 256)
    rows = [r for r in csv.DictReader(open(path)) if any(k in r["Kernel_Name"] for k in ("conv_ksplit", "conv_wino", "conv1x1_stream"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // 147 * 147
    rows = rows[len(rows) - n:]                                  # whole forwards only, the last ones (warm)
    cins = [64] + ([64 * (c + 1) for c in range(8)] + [576]) * 16 + [1024, 64]
    taps = [9] + ([9] * 8 + [1]) * 16 + [1, 9]
    dur = defaultdict(list)
    gaps = []
    for i, r in enumerate(rows):
        k = i % 147
        dur[(cins[k], taps[k])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if k:
            gaps.append(int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
    fw = n // 147
    tot = 0.0
    for (cin, tp), v in sorted(dur.items(), key=lambda kv: (kv[0][1] == 1, kv[0][0])):
        us = sum(v) / len(v) / 1e3
        fl = 2.0 * 64 * cin * tp * h * w
        per_fw = sum(v) / fw / 1e6
        tot += per_fw
        print(f"Cin {cin:5d} taps {tp}: {us:8.1f} us  {100 * fl / (us * 1e-6) / PEAK:5.1f} % of peak   {len(v) // fw:3d} per forward = {per_fw:6.3f} ms")
    print(f"kernels {tot:.2f} ms per forward; gaps between launches {sum(gaps) / fw / 1e6:.3f} ms ({sum(gaps) / len(gaps) / 1e3:.1f} us each)")


if __name__ == "__main__":
    main()
