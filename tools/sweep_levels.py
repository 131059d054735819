"""Per-level durations of the LU sweeps from a rocprofv3 kernel trace (csv) of tools/bench_ndlu.py: the last apply's launches."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "nd_fwd_kernel" in r["Kernel_Name"] or "nd_bwd_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size") or r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1) or 1), int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X", 256))))
rows.sort()
n = int(sys.argv[2])  # launches per apply
last = rows[-n:]
t0 = last[0][0]
prev_end = None
for s, e, k, g, w in last:
    lpr = k.split("<")[1].split(",")[2].strip() if "<" in k else "16"
    name = ("fwd" if "nd_fwd" in k else "bwd") + {"64": "8", "16": "32", "4": "128"}.get(lpr, "?")
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{name:6s} wgs {g // w:7d}  {1e-3 * (e - s):8.2f} us  gap {gap:6.2f} us")
    prev_end = e
print(f"total {1e-3 * (last[-1][1] - t0):.1f} us, kernels {1e-3 * sum(e - s for s, e, *_ in last):.1f} us")
