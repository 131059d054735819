"""Summarise a rocprofv3 kernel trace (csv): GPU busy time, idle gaps by size, kernels by total time inside a window.

usage: trace_gaps.py KERNEL_TRACE.csv [--last-ms MS]   (window = the last MS milliseconds of the trace; default all)
"""
import argparse
import csv
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--last-ms", type=float, default=0.0)
ap.add_argument("--top", type=int, default=25)
args = ap.parse_args()
rows = []
with open(args.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(r[1] for r in rows)
if args.last_ms > 0:
    rows = [r for r in rows if r[0] >= t_end - args.last_ms * 1e6]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0
cur_end = rows[0][0]
gaps = []
for s, e, _ in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
        busy += e - s
        cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
print(f"window {1e-6 * (t1 - t0):.2f} ms, {len(rows)} kernels, busy {1e-6 * busy:.2f} ms ({100 * busy / (t1 - t0):.1f} %)")
buckets = [(0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 20e3), (20e3, 50e3), (50e3, 200e3), (200e3, 1e6), (1e6, 1e12)]
for lo, hi in buckets:
    g = [x for x in gaps if lo <= x < hi]
    if g:
        print(f"  gaps {lo / 1e3:7.0f}-{hi / 1e3:9.0f} us: {len(g):7d}  total {1e-6 * sum(g):8.2f} ms")
by = defaultdict(lambda: [0, 0])
for s, e, k in rows:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    k = k.split("(")[0]
    by[k][0] += 1
    by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[: args.top]:
    print(f"  {k[:70]:70s} {c:7d} x {1e-3 * t / c:8.2f} us = {1e-6 * t:8.2f} ms")
