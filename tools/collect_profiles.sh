# Round evidence on one MI355X (run through gpurun): the default bench line, the rocprofv3 kernel statistics of the same
# command, the PMC passes (separate runs, no trace domains beside them) for the SpMV and for the LU sweeps, the per-level
# sweep tables.  Everything lands in gpurun_out/$R/; the summaries worth keeping are copied to profiles/ by hand.
R=${1:-r03}
cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/$R
mkdir -p $OUT
set -x
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 bench.py > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_rocprof.err || exit 1
cp $(find $OUT/prof_bench -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof_bench
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_spmv_$c -- python3 tools/spmv_only.py > $OUT/pmc_spmv_$c.log 2>&1 || exit 1
  cp $(find $OUT/pmc_spmv_$c -name '*counter_collection.csv' | head -1) $OUT/spmv_pmc_$c.csv
  rm -rf $OUT/pmc_spmv_$c
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_lu_$c -- python3 tools/bench_ndlu.py --case S500k --refactors 1 > $OUT/pmc_lu_$c.log 2>&1 || exit 1
  cp $(find $OUT/pmc_lu_$c -name '*counter_collection.csv' | head -1) $OUT/lu_pmc_$c.csv
  rm -rf $OUT/pmc_lu_$c
done
python3 tools/summarise_pmc.py $OUT > $OUT/pmc_summary.json
cat $OUT/pmc_summary.json
