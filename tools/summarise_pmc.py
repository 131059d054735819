"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/collect_profiles.sh): the SpMV on SROOF and
the two sweep kernels of the exact LU on S500k.  FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950
wide coalesced reads; counters are in KiB.  Prints a JSON record (profiles/rNN_spmv_traffic.json, rNN_lu_sweeps_traffic.json)."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])


def per_kernel(path, want):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if any(w in name for w in want):
                m = re.search(r"((?:nd_|spmv_)\w+(?:<[^>]*>)?)", name)
                acc[m.group(1) if m else name].append(float(r["Counter_Value"]))
    return acc


rec = {}
# ---- SpMV: the last 10 launches of each variant (timed loop of tools/spmv_only.py) ----
fs, ws = per_kernel(out / "spmv_pmc_FETCH_SIZE.csv", ["spmv_"]), per_kernel(out / "spmv_pmc_WRITE_SIZE.csv", ["spmv_"])
spmv = {}
for k in fs:
    f_kib = sum(fs[k][-10:]) / len(fs[k][-10:])
    w_kib = sum(ws[k][-10:]) / len(ws[k][-10:]) if k in ws else 0.0
    spmv[k] = {"fetch_size_kib": f_kib, "write_size_kib": w_kib, "traffic_bytes": int((2 * f_kib + w_kib) * 1024), "launches_averaged": len(fs[k][-10:])}
rec["spmv"] = spmv
# ---- LU sweeps: one apply = the launches of all levels; the timed loop runs 100 applies after 10 + 1: sum over one apply ----
fl, wl = per_kernel(out / "lu_pmc_FETCH_SIZE.csv", ["nd_fwd_kernel", "nd_bwd_kernel"]), per_kernel(out / "lu_pmc_WRITE_SIZE.csv", ["nd_fwd_kernel", "nd_bwd_kernel"])
tot_f = sum(sum(v) for v in fl.values())
tot_w = sum(sum(v) for v in wl.values())
nlaunch = sum(len(v) for v in fl.values())
rec["lu_sweeps"] = {"launches_counted": nlaunch, "fetch_size_kib_total": tot_f, "write_size_kib_total": tot_w,
                    "by_kernel": {k: {"launches": len(v), "fetch_kib": sum(v), "write_kib": sum(wl.get(k, []))} for k, v in fl.items()}}
print(json.dumps(rec, indent=1))
