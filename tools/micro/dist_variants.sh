cd $GRAFT_REPO_ROOT
for env in "LSA_ND_WORK_MB=1000" "LSA_ND_DIST_SPLIT=8000" "LSA_ND_WORK_MB=1000,LSA_ND_DIST_SPLIT=8000"; do
  echo "== $env"
  timeout -k 10 400 python tools/run_dist_one_gpu.py --case C300k --ranks 4 --env "$env" 2> gpurun_out/dv.err | grep '^{"case"' > gpurun_out/dv.json
  python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/dv.json").read())
r=d["rank0"]; s=r["stats"]
print("bit_identical", d["ranks_bit_identical"], "max_residual %.2e"%r["max_residual"], "applies", s["op_applies"], "restarts", s["krylov_restarts"], "max_rel_res %.1e"%s["max_rel_res"], "refined", s["refined_solves"], "dist nodes", r["top_nodes_distributed"], "factor s", round(s["seconds_factor"],1))
PY
done
