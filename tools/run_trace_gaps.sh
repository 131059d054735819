set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_prof/gaps -- python3 tools/time_solve.py --case S30k --reps 6 > gpurun_out/r3_prof/gaps.log 2>&1
f=$(find gpurun_out/r3_prof/gaps -name '*kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py $f --last-ms 70 --top 12
rm -rf gpurun_out/r3_prof/gaps
grep "^solve" gpurun_out/r3_prof/gaps.log | tail -2
