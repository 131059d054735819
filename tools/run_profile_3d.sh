# kernel split of the 3D factorisation (development aid): bash tools/run_profile_3d.sh C300k
set -e
cd /root/repo
export TMPDIR=/tmp
case=${1:-C300k}
mkdir -p gpurun_out/r3_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof/$case -- python3 tools/bench_ndlu.py --case $case --refactors ${2:-2} > gpurun_out/r3_prof/$case.log 2>&1
f=$(find gpurun_out/r3_prof/$case -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r3_prof/${case}_kernel_stats.csv
rm -rf gpurun_out/r3_prof/$case
grep -v "rocprofv3\|output_stream" gpurun_out/r3_prof/$case.log | tail -8
head -16 gpurun_out/r3_prof/${case}_kernel_stats.csv | cut -c1-220
