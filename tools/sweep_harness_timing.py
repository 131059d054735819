"""End-to-end timing of the Reynolds-sweep harness (examples/eigenvalues.py) on synthetic S30k matrices: files on disk ->
MatrixMarket parse -> ordering -> upload -> shift-invert solve -> sigma file, for --jobs 1 and --jobs 2."""
import importlib.util
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the first HIP context
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
spec = importlib.util.spec_from_file_location("lsa_examples_eigenvalues", ROOT / "lsa-fw_amd" / "examples" / "eigenvalues.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
case = sys.argv[1] if len(sys.argv) > 1 else "S30k"
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    t0 = time.time()
    mod.synthesize(Path(tmp), case)
    print(f"synthesized {len(mod._REYNOLDS)} (A, M) pairs of {case} in {time.time() - t0:.1f} s", flush=True)
    for jobs in (1, 2, 1, 2):
        t0 = time.time()
        mod.main(["--save-dir", tmp, "--jobs", str(jobs)])
        dt = time.time() - t0
        print(f"--jobs {jobs}: {len(mod._REYNOLDS)} Reynolds numbers in {dt:.2f} s ({dt / len(mod._REYNOLDS) * 1e3:.0f} ms each, files to sigma)", flush=True)
