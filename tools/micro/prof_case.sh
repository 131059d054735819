# kernel statistics of a few eigen-solves of one case:  bash tools/micro/prof_case.sh S500k 2
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
case=${1:-S30k}; reps=${2:-4}
timeout -k 10 300 python3 tools/micro/solve_loop.py $case $reps
LSA_SPTRSV_GRAPH=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$case -- python3 tools/micro/solve_loop.py $case $reps > gpurun_out/prof_$case.log 2>&1
f=$(find gpurun_out/prof_$case -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 2), "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:26]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    print(f"  {n[:64]:64s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
