cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04b
mkdir -p $OUT
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 bench.py > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_rocprof.err || exit 1
cp $(find $OUT/prof_bench -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof_bench
python3 -c "
import json
d=json.load(open('$OUT/bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])
c=d['config']
print({k:c[k] for k in ('seconds_expand','seconds_dense','seconds_restart','seconds_factor','prepare_ms','cold_first_solve_ms')})
for k in ('two_solves_in_flight','three_solves_in_flight','sptrsv_roofline','pattern_3d','other_pc'):
    print(k, json.dumps(c.get(k))[:400])
"
