#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats output directory to compact rows
(kernel names shortened) and write it under profiles/.

    python tools/summarize_rocprof.py gpurun_out/prof_r01 profiles/r01_kernel_stats.csv "command line"
"""
import csv
import glob
import os
import sys


def short(name: str) -> str:
    name = name.strip('"')
    if len(name) > 90:
        name = name[:60] + "...<" + str(len(name)) + " chars>"
    return name


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    stats = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))
    if not stats:
        raise SystemExit("no *_kernel_stats.csv under " + src)
    rows = []
    for p in stats:
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                rows.append(r)
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst, "w", newline="") as f:
        if note:
            f.write(f"# {note}\n")
        f.write("# source: rocprofv3 --kernel-trace --stats (kernel_stats.csv), names shortened\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    print(open(dst).read())


if __name__ == "__main__":
    main()
