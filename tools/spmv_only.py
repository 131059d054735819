"""Launch the SROOF SpMV a few times (for rocprofv3 --pmc / --kernel-trace passes)."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--roof-case", default="S500k")
ap.add_argument("--roof-reps", type=int, default=10)
ap.add_argument("--roof-iters", type=int, default=10)
args = ap.parse_args()
out = bench.spmv_roofline(args, 0)
print({k: v for k, v in out.items()})
