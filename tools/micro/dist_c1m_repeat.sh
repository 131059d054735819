# repeats the C1M / 4-rank rehearsal on one GPU; usage: bash tools/micro/dist_c1m_repeat.sh "<env>" <runs>
cd $GRAFT_REPO_ROOT
env="$1"; runs=${2:-3}
for i in $(seq 1 $runs); do
  timeout -k 10 330 python tools/run_dist_one_gpu.py --case C1M --ranks 4 --env "$env" 2> gpurun_out/dv$i.err | grep '^{"case"' > gpurun_out/dv$i.json
  rc=$?
  python3 - $i $rc <<'PY'
import json, sys
i, rc = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/dv{i}.json").read())
    r = d["per_rank"]
    print(f"run {i}: bit_identical {d['ranks_bit_identical']} applies {[x['op_applies'] for x in r]} gathers {[x['allgather_calls'] for x in r]} refined {[x['refined_solves'] for x in r]} max_residual {r[0]['max_residual']:.2e}", flush=True)
except Exception as e:
    print(f"run {i}: no result (rc {rc}: hang or failure) {e!r}", flush=True)
PY
done
