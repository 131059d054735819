# matrix-core busy fraction of the factorisation's product kernels (own PMC pass, no trace domains): bash tools/run_pmc_mfma.sh C300k
set -e
cd /root/repo
export TMPDIR=/tmp
case=${1:-C300k}
mkdir -p gpurun_out/r3_prof
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r3_prof/pmc_mfma_$case -- python3 tools/bench_ndlu.py --case $case --refactors 1 > gpurun_out/r3_prof/pmc_mfma_$case.log 2>&1
f=$(find gpurun_out/r3_prof/pmc_mfma_$case -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/r3_prof/mfma_util_$case.json
import csv, json, re, sys, collections
busy = collections.defaultdict(float); act = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    mt = re.search(r"(nd_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
    if not mt or ("gemm" not in mt.group(1) and "gj_update" not in mt.group(1)):
        continue
    k = mt.group(1)
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += v; calls[k] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[k] += v
out = {k: {"launches": calls[k], "mfma_busy_cycles": busy[k], "gui_active_cycles": act[k], "mfma_util_percent_of_1024_simds": 100.0 * busy[k] / (act[k] * 1024.0) if act[k] else None} for k in busy}
print(json.dumps(out, indent=1))
PY
head -3 "$f" > gpurun_out/r3_prof/pmc_mfma_head_$case.csv; rm -rf gpurun_out/r3_prof/pmc_mfma_$case
