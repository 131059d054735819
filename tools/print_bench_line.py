"""Print the headline fields of a bench.py JSON line (development aid)."""
import json
import sys

for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c = d["config"]
    keys = ("layout", "allgather_calls_per_solve", "op_applies_per_solve", "gmres_iters_per_apply", "speedup_over_one_gpu_same_workload", "max_residual",
            "seconds_factor")
    print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "scaling")}, {k: c.get(k) for k in keys}, c.get("replicas"))
