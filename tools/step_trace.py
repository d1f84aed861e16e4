import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last decode_kernel<2, true, true> (training forward of the last step) and print until the end of that step
idx = [i for i, r in enumerate(rows) if "decode_kernel<2, true, true>" in r["Kernel_Name"]]
i0 = idx[-1] - 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
tot = collections.Counter()
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name}")
    tot["gap"] += max(0, s - prev_end)
    tot["busy"] += e - s
    prev_end = max(prev_end, e)
print("total busy %.1f us, gaps %.1f us, span %.1f us" % (tot["busy"] / 1e3, tot["gap"] / 1e3, (prev_end - t0) / 1e3))
# where do library GEMMs (Cijk_*) / MIOpen kernels sit in the whole trace?
for i, r in enumerate(rows):
    if r["Kernel_Name"].startswith(("Cijk_", "igemm", "im2col", "miopen", "MIOpen")):
        print("library kernel at trace position", i, "of", len(rows), ":", r["Kernel_Name"][:50], "| before:", rows[i - 1]["Kernel_Name"][:60], "| after:", rows[i + 1]["Kernel_Name"][:60])
