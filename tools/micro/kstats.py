"""Prints name / calls / average of the kernels of an Arnoldi step from a rocprofv3 *_kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("cgs_", "spmv_", "nd_fwd", "nd_bwd", "basis_gemm", "nd_perm", "copyBuffer")):
        short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print(f"  {short[:70]:70s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
