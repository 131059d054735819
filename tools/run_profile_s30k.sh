set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof/s30k -- python3 tools/time_solve.py --case S30k --reps 6 > gpurun_out/r3_prof/s30k.log 2>&1
f=$(find gpurun_out/r3_prof/s30k -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r3_prof/s30k_kernel_stats.csv
rm -rf gpurun_out/r3_prof/s30k
grep -v "rocprofv3\|output_stream" gpurun_out/r3_prof/s30k.log | tail -4
head -40 gpurun_out/r3_prof/s30k_kernel_stats.csv | cut -c1-200
