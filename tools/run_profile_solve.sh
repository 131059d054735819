# kernel statistics of repeated eigen-solves of one case (development aid): bash tools/run_profile_solve.sh S500k 3
set -e
cd /root/repo
export TMPDIR=/tmp
case=${1:-S500k}
mkdir -p gpurun_out/r3_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof/solve_$case -- python3 tools/time_solve.py --case $case --reps ${2:-3} > gpurun_out/r3_prof/solve_$case.log 2>&1
f=$(find gpurun_out/r3_prof/solve_$case -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r3_prof/solve_${case}_kernel_stats.csv
rm -rf gpurun_out/r3_prof/solve_$case
grep "^solve" gpurun_out/r3_prof/solve_$case.log | tail -2
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r3_prof/solve_${case}_kernel_stats.csv")))
for r in rows[:22]:
    nm=r["Name"]
    nm=nm.split("(anonymous namespace)::")[-1] if "anonymous" in nm else nm
    nm=nm.replace("void ","")[:70]
    print(f"{nm:72s} calls {int(r['Calls']):7d} total ms {int(r['TotalDurationNs'])/1e6:9.2f} avg us {float(r['AverageNs'])/1e3:8.1f}")
PY
