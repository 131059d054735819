cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  timeout -k 10 330 python tools/run_dist_one_gpu.py --case C1M --ranks 4 --env "LSA_ND_WORK_MB=4000,LSA_TRACE_COMM=1" 2> gpurun_out/dc.err | grep '^{"case"' > gpurun_out/dc.json
  python3 - <<'PY'
import json, re, collections
seq = collections.defaultdict(list)
for ln in open("gpurun_out/dc.err", errors="replace"):
    m = re.match(r"\[comm rank (\d+)\] call (\d+) bytes (\d+)", ln)
    if m: seq[int(m.group(1))].append(int(m.group(3)))
hc = collections.defaultdict(list)
for ln in open("gpurun_out/dc.err", errors="replace"):
    m = re.match(r"\[hcol rank (\d+)\] (.*)", ln)
    if m: hc[int(m.group(1))].append(m.group(2).strip())
for r in sorted(hc):
    k = next((i for i in range(min(len(hc[r]), len(hc[0]))) if hc[r][i] != hc[0][i]), None)
    print(f"hcol rank {r}: {len(hc[r])} columns; first difference from rank 0 at index {k}" + (f": {hc[r][k]} | rank0 {hc[0][k]}" if k is not None else ""), flush=True)
ap = collections.defaultdict(list)
for ln in open("gpurun_out/dc.err", errors="replace"):
    m = re.match(r"\[trace rank (\d+)\] (.*) calls", ln)
    if m: ap[int(m.group(1))].append(m.group(2).strip())
for r in sorted(ap):
    k = next((i for i in range(min(len(ap[r]), len(ap[0]))) if ap[r][i] != ap[0][i]), None)
    print(f"apply rank {r}: {len(ap[r])} applies; first difference from rank 0 at index {k}" + (f": {ap[r][k]} | rank0 {ap[0][k]}" if k is not None else ""), flush=True)
n = {r: len(v) for r, v in seq.items()}
print("calls per rank", n, flush=True)
ref = seq[0]
for r in sorted(seq):
    v = seq[r]
    k = next((i for i in range(min(len(v), len(ref))) if v[i] != ref[i]), None)
    if k is not None or len(v) != len(ref):
        print(f"rank {r} differs from rank 0 at call {k}: {v[k-3:k+4] if k is not None else None} vs {ref[k-3:k+4] if k is not None else None}", flush=True)
try:
    d = json.loads(open("gpurun_out/dc.json").read())
    print("bit_identical", d["ranks_bit_identical"], [x["op_applies"] for x in d["per_rank"]], "max_residual %.2e" % d["per_rank"][0]["max_residual"], flush=True)
    ok = d["ranks_bit_identical"]
except Exception as e:
    print("no result", repr(e), flush=True); ok = False
open("gpurun_out/dc.ok", "w").write("1" if ok else "0")
PY
  if [ "$(cat gpurun_out/dc.ok)" = "0" ]; then echo "failure captured in run $i"; cp gpurun_out/dc.err gpurun_out/dc_failed.err; break; fi
done
