set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fac_s30k -- python3 tools/bench_ndlu.py --case S30k --refactors 9 > gpurun_out/fac_s30k.log 2>&1
f=$(find gpurun_out/fac_s30k -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot / 1e6, "launches", calls)
for r in rows[:28]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    print(f"  {n[:64]:64s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.2f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
tail -3 gpurun_out/fac_s30k.log
