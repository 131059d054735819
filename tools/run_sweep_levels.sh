set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_base
for c in S30k S500k; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_base/$c -- python3 tools/bench_ndlu.py --case $c --refactors 2 > gpurun_out/r3_base/$c.log 2>&1
  f=$(find gpurun_out/r3_base/$c -name '*kernel_trace.csv' | head -1)
  n=$(grep -o "'apply_launches': [0-9]*" gpurun_out/r3_base/$c.log | grep -o '[0-9]*$' | head -1)
  python3 tools/sweep_levels.py $f $n > gpurun_out/r3_base/$c.levels.txt
  rm -rf gpurun_out/r3_base/$c
  tail -3 gpurun_out/r3_base/$c.log
done
