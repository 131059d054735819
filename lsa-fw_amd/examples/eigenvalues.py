"""Cylinder flow: solve the EVP on pre-assembled (A, M) per Reynolds number and write sigma.

Counterpart of ``/root/reference/.examples/eigenvalues.py:61-108`` on the HIP path:
same Reynolds sweep and shift table (``:36-49``), same solver configuration (``num_eig = 5``, ``atol = 1e-3``,
shift-invert at the tabulated target, LU-class inner solves, ``:95-100``), same output file
``cases/cylinder/reynolds_<Re>/sigma_eig0.txt`` holding ``"{re} {im}\\n"`` (``:104-107``).

The reference loads ``A.mtx`` / ``M.mtx`` written by its dolfinx assembly stage.  ``--synthesize`` writes the oracle's
synthetic pair to the same place first, so the script runs end to end without dolfinx.
"""

from __future__ import annotations

import argparse
import logging
import sys
from pathlib import Path
from typing import Final

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "lsa-fw_amd")]

from FEM.utils import iPETScMatrix  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

_SAVE_DIR: Final[Path] = Path("cases") / "cylinder"
_NUM_EIG: Final[int] = 5
_EIG_INDEX: Final[int] = 0
_ATOL: Final[float] = 1e-3

_REYNOLDS: Final[tuple[float, ...]] = tuple(range(40, 91, 5))
_TARGETS: Final[tuple[complex, ...]] = (  # DOI:10.1115/1.4042737, as tabulated in the reference
    (-0.03 + 0.7197388769374216j),
    0.7316769290210628j,
    (0.018 + 0.7379601143282424j),
    (0.03 + 0.742986662573986j),
    (0.05 + 0.744243299635422j),
    (0.061 + 0.7461282552275759j),
    (0.072 + 0.7461282552275759j),
    (0.085 + 0.744557458900781j),
    (0.09 + 0.742986662573986j),
    (0.1 + 0.7398450699203962j),
    (0.115 + 0.7351326809400116j),
)

logger = logging.getLogger(__name__)


def synthesize(save_dir: Path, case: str) -> None:
    """Write the synthetic (A, M) of synthetic/fem.py for every Reynolds number of the sweep."""
    sys.path.insert(0, str(ROOT))
    from synthetic import fem

    for re in _REYNOLDS:
        mat_dir = save_dir / f"reynolds_{re:.1f}" / "matrices"
        mat_dir.mkdir(parents=True, exist_ok=True)
        es = fem.cylinder_case(case, re=float(re))
        iPETScMatrix(es.A).export(mat_dir / "A.mtx")
        iPETScMatrix(es.M).export(mat_dir / "M.mtx")


def solve_case(save_dir: Path, re: float, target: complex) -> Path | None:
    """One Reynolds number of the sweep: read the MatrixMarket pair, shift-invert at the tabulated target with the exact LU
    as inner solver, store the selected eigenvalue as ``"<real> <imag>"`` (the call sequence of the reference's loop body,
    ``.examples/eigenvalues.py:61-107``; same file names and output format, so its post-processing reads these results)."""
    root = save_dir / f"reynolds_{re:.1f}"
    files = {name: root / "matrices" / f"{name}.mtx" for name in ("A", "M")}
    absent = [str(path) for path in files.values() if not path.exists()]
    if absent:
        logger.warning("Re %.1f left out, no such file: %s", re, ", ".join(absent))
        return None
    pair = {}
    for name, path in files.items():
        mat = iPETScMatrix.from_path(path)
        mat.assemble()
        logger.info("Re %.1f: %s read from %s: %d x %d, %d stored entries, Frobenius norm %.3e", re, name, path, *mat.shape, mat.nonzero_entries, mat.norm)
        pair[name] = mat

    eigensolver = EigenSolver(pair["A"], pair["M"], cfg=EigensolverConfig(num_eig=_NUM_EIG, atol=_ATOL), check_hermitian=False)
    eps = eigensolver.solver
    eps.set_st_type(iSTType.SINVERT)
    eps.set_target(target)
    eps.set_st_pc_type(PreconditionerType.LU)
    eps.solve()
    value = eps.get_eigenvalue(_EIG_INDEX)
    result_file = root / f"sigma_eig{_EIG_INDEX}.txt"
    result_file.write_text(f"{value.real} {value.imag}\n", encoding="utf-8")
    logger.info("Re %.1f: eigenvalue %d nearest %s is %s -> %s", re, _EIG_INDEX, target, value, result_file)
    eps.release()
    return result_file


def main(argv: list[str] | None = None) -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--save-dir", type=Path, default=_SAVE_DIR)
    ap.add_argument("--synthesize", metavar="CASE", help="first write synthetic matrices (e.g. S5k)")
    ap.add_argument("--jobs", type=int, default=1,
                    help="Reynolds numbers in flight on the GPU at once (threads, one HIP context and stream set each). One "
                         "solve is a chain of dependent launches that leaves most of an MI355X idle: two in flight deliver "
                         "1.5x, three 2x the eigenpairs per second; four oversubscribe the hardware queues and fall back "
                         "below two (DESIGN.md section 6).  The host-side loading and ordering of one case hides behind the "
                         "GPU work of the others.")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    if args.synthesize:
        synthesize(args.save_dir, args.synthesize)
    cases = list(zip(_REYNOLDS, _TARGETS))
    import os

    os.environ.setdefault("LSA_HOST_BLAS_THREADS", "1")  # this script owns its process: keep spinning BLAS workers off the launch path
    if args.jobs <= 1:
        for re, target in cases:
            solve_case(args.save_dir, re, target)
    else:
        from concurrent.futures import ThreadPoolExecutor

        # every solve drives up to four streams; the runtime's default of four hardware queues per process would
        # serialise two solves on the same queues (read when the first HIP context is created)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (more than eight queues per GPU slow everything down: DESIGN.md section 6)
        with ThreadPoolExecutor(max_workers=args.jobs) as pool:
            for fut in [pool.submit(solve_case, args.save_dir, re, target) for re, target in cases]:
                fut.result()
    logger.info("All cases processed.")


if __name__ == "__main__":
    main()
