#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from counter_collection.csv files.
    python tools/pmc_summary.py gpurun_out/pmc2 [gpurun_out/pmc3 ...]"""
import collections, csv, glob, os, sys

def main():
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                if "at::native" in k:
                    continue
                print(f"{d}  {k}")
                for c, vals in sorted(v.items()):
                    print(f"    {c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
if __name__ == "__main__":
    main()
