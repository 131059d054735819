"""GPU suite: the eigen-solve behind ONE library call (``lsa_eigs_sinvert`` / ``lsa_krylov_solve``: Krylov-Schur with the
library's own dense algebra) against the golden fixture and against the Python-over-LAPACK loop
(``lsa_hip/krylov_schur.py``, the test double)."""

import json
from pathlib import Path

import numpy as np
import pytest

import helpers  # noqa: F401

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def test_one_call_eigensolve_matches_the_s30k_golden_fixture(hip_ctx):
    """A consumer of include/lsa_hip.h: lsa_ctx_create -> lsa_csr_upload x 2 -> lsa_eigs_sinvert, nothing else (no ordering, no
    prepared analysis, no Python loop).  BASELINE config 2 at rtol 1e-8 against the oracle's eigenvalues."""
    import lsa_hip
    from oracle import shift_invert
    from synthetic import fem

    gold = json.loads((GOLDEN / "cylinder_s30k_k20.json").read_text())
    es = fem.cylinder_case("S30k")
    assert es.n == gold["n"] and es.A.nnz == gold["nnz"]
    sigma = complex(*gold["sigma"])
    ref = np.array([complex(a, b) for a, b in gold["eigenvalues"]])
    dA = lsa_hip.CsrMatrix.from_scipy(hip_ctx, es.A)
    dM = lsa_hip.CsrMatrix.from_scipy(hip_ctx, es.M)
    lam, X, est, res, st = lsa_hip.eigs_sinvert(hip_ctx, dA, dM, sigma, 20, ncv=80, tol=1e-10)
    assert res["nconv"] >= 20 and len(lam) >= 20
    d = helpers.match_nearest(lam, ref[:20])
    assert d.max() <= 1e-8, d
    assert np.all(np.diff(np.abs(lam - sigma)) >= -1e-12)  # nearest the target first
    r = shift_invert.compute_residuals(es.A.astype(complex), es.M.astype(complex), lam[:20], X[:, :20])
    assert r.max() <= 1e-8 and np.allclose(np.linalg.norm(X, axis=0), 1.0)
    assert st["gmres_iters"] == 0 and st["pc_fallback"] == 0 and st["op_applies"] == res["op_applies"]


@pytest.mark.parametrize("case,k,ncv", [("S5k", 20, 80), ("S2k", 6, 40)])
def test_library_loop_equals_the_python_loop(monkeypatch, case, k, ncv):
    """Same start vector, same device kernels; only the dense algebra on the projected matrix differs (in-tree QR algorithm
    against LAPACK): same eigenvalues to 1e-10, same number of operator applies, eigenvectors equal up to rounding."""
    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    es = fem.cylinder_case(case)
    out = {}
    for driver in ("python", "native"):
        monkeypatch.setenv("LSA_KS_DRIVER", driver)
        s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=k, atol=1e-10, ncv=ncv), check_hermitian=False)
        s.solver.set_st_type(iSTType.SINVERT)
        s.solver.set_st_pc_type(PreconditionerType.LU)
        s.solver.set_target(fem.SIGMA_RE50)
        pairs = s.solve()
        assert len(pairs) == k and s.solver.residuals()[:k].max() <= 1e-8
        out[driver] = (np.array([p[0] for p in pairs]), np.column_stack([s.solver.get_eigenvector_array(i) for i in range(k)]), s.solver.stats)
        s.solver.release()
    (l1, X1, s1), (l2, X2, s2) = out["python"], out["native"]
    assert np.abs(l1 - l2).max() <= 1e-10 * np.abs(l1).max()
    assert s1["op_applies"] == s2["op_applies"] and s1["krylov_restarts"] == s2["krylov_restarts"]
    assert np.abs(np.abs(np.einsum("ij,ij->j", X1.conj(), X2)) - 1.0).max() <= 1e-8


def test_which_policies_and_breakdown_through_the_library_loop(hip_ctx):
    """lsa_krylov_solve on small operators: every EPSWhich code, the plain shift transform, an exact breakdown (invariant
    subspace reached, fresh direction injected by the library's own generator), fewer pairs than asked for."""
    import scipy.sparse as sp

    import lsa_hip

    n = 60
    rng = np.random.default_rng(2)
    d = np.sort(rng.uniform(1.0, 9.0, n)) + 1j * rng.uniform(-1.0, 1.0, n)
    A = lsa_hip.CsrMatrix.from_scipy(hip_ctx, sp.diags(d).tocsr())
    I = lsa_hip.CsrMatrix.from_scipy(hip_ctx, sp.identity(n, format="csr"))
    sigma = 4.0 + 0.2j
    op = lsa_hip.ShiftInvertOperator(hip_ctx, A, I, sigma, pc_type=2)
    kb = lsa_hip.KrylovBasis(hip_ctx, op, 30)
    op2 = lsa_hip.ShiftInvertOperator(hip_ctx, A, None, 0.0, mode=1, pc_type=0)  # plain shift with sigma = 0: the operator is A
    kb2 = lsa_hip.KrylovBasis(hip_ctx, op2, 30)
    # the target policies through shift-invert, the extreme ones (largest / smallest ...) through the untransformed operator
    want = {7: np.abs(d - sigma), 8: np.abs(d.real - sigma.real), 9: np.abs(d.imag - sigma.imag), 1: -np.abs(d), 3: -d.real, 4: d.real, 5: -d.imag, 6: d.imag}
    for which, key in want.items():
        res = kb.solve(4, 1e-10, 200, which, 0, sigma, seed=1) if which >= 7 else kb2.solve(3, 1e-10, 500, which, 1, 0.0, seed=1)
        assert res.nconv >= 3, which
        found = np.array([int(np.argmin(np.abs(d - z))) for z in res.lam])
        assert np.abs(d[found] - res.lam).max() <= 1e-8 and np.all(np.diff(key[found]) >= -1e-9), which  # eigenvalues, in the requested order
        if which not in (8, 9):  # (one coordinate of the target only: a Krylov space built around sigma need not hold the global optimum)
            best = d[np.argsort(key, kind="stable")[:3]]
            assert helpers.match_nearest(res.lam, best).max() <= 1e-8, which
        X = res.vectors
        assert np.abs(d[:, None] * X - X * res.lam[None, :]).max() <= 1e-7
    # invariant subspace of dimension 3: breakdown, continued with fresh directions; every returned pair is exact
    v = np.zeros(n, dtype=np.complex128)
    v[[4, 9, 20]] = 1.0
    res = kb.solve(3, 1e-12, 50, 7, 0, sigma, v0=v)
    assert res.nconv >= 3 and helpers.match_nearest(d, res.lam).max() <= 1e-10  # every returned value is an eigenvalue


_CGS_CHILD = r"""
import json, sys
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/lsa-fw_amd", sys.argv[1] + "/tests"]
import numpy as np
from synthetic import fem
from Solver.eigen import EigenSolver, EigensolverConfig
from Solver.utils import PreconditionerType, iSTType
out = {}
es = fem.cylinder_case("S5k")
s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=8, atol=1e-10, ncv=40), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT); s.solver.set_st_pc_type(PreconditionerType.LU); s.solver.set_target(fem.SIGMA_RE50)
pairs = s.solve()
out["cylinder"] = {"lam": [[p[0].real, p[0].imag] for p in pairs], "res": float(s.solver.residuals()[:8].max()), "applies": s.solver.stats["op_applies"],
                   "max_rel_res": s.solver.stats["max_rel_res"], "spmv_calls": s.solver.stats["spmv_calls"]}
s.solver.release()
K, M, _bnd = fem.assemble_membrane(24, 24)  # real symmetric pair: the float64 kernels
s = EigenSolver(K, M, EigensolverConfig(num_eig=6, atol=1e-10, ncv=30), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT); s.solver.set_st_pc_type(PreconditionerType.LU); s.solver.set_target(0.0)
pairs = s.solve()
out["membrane"] = {"lam": [[p[0].real, p[0].imag] for p in pairs], "res": float(s.solver.residuals()[:6].max()), "applies": s.solver.stats["op_applies"]}
print(json.dumps(out))
"""


def test_cgs2_kernel_per_stage_forms_agree_with_the_five_launch_form(tmp_path):
    """The long-vector form of CGS2 (kernel per stage; from round 3 the first projection and the second dot product in one pass
    over the basis, ``cgs_axpy_dot_kernel``) only runs by itself beyond 262 k unknowns.  Forced onto a 5 k-unknown case
    (``LSA_KRYLOV_FUSED=0``; the switches are read once per process, hence child processes) with three and with four passes
    over the basis: the eigenvalues of the default five-launch form to 1e-10, complex and float64 kernels."""
    import os
    import subprocess
    import sys

    root = str(Path(__file__).resolve().parents[1])
    runs = {}
    for name, env in (("default", {}), ("three_passes", {"LSA_KRYLOV_FUSED": "0"}), ("four_passes", {"LSA_KRYLOV_FUSED": "0", "LSA_KRYLOV_PASSES": "4"}),
                      ("no_tail", {"LSA_KRYLOV_TAIL": "0"})):
        p = subprocess.run([sys.executable, "-c", _CGS_CHILD, root], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        runs[name] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    for prob in ("cylinder", "membrane"):
        ref = np.array([complex(a, b) for a, b in runs["default"][prob]["lam"]])
        for name in ("three_passes", "four_passes"):
            lam = np.array([complex(a, b) for a, b in runs[name][prob]["lam"]])
            assert runs[name][prob]["res"] <= 1e-8
            assert len(lam) == len(ref) and np.abs(lam - ref).max() <= 1e-10 * np.abs(ref).max(), (prob, name)
        # round 4: the tail form of a step (its last launch normalises, multiplies M v for the next step and checks the inner
        # solve: csrc/blas.hip::cgs_tail_kernel) against the five launches plus two products it replaces: the SAME bits, the same
        # operator applies; only the sums of the check are added in another order
        assert runs["no_tail"][prob]["lam"] == runs["default"][prob]["lam"], prob
        assert runs["no_tail"][prob]["applies"] == runs["default"][prob]["applies"]
    a, b = runs["default"]["cylinder"]["max_rel_res"], runs["no_tail"]["cylinder"]["max_rel_res"]
    assert 0.0 < a <= 1e-11 and abs(a - b) <= 1e-3 * max(a, b), (a, b)


_REPRO_CHILD = r"""
import json, sys
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/lsa-fw_amd", sys.argv[1] + "/tests"]
from synthetic import fem
from Solver.eigen import EigenSolver, EigensolverConfig
from Solver.utils import PreconditionerType, iSTType
es = fem.cylinder_case("S30k")
s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=8, atol=1e-10, ncv=40), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT); s.solver.set_st_pc_type(PreconditionerType.LU); s.solver.set_target(fem.SIGMA_RE50)
out = []
for _ in range(int(sys.argv[2])):
    pairs = s.solve()
    out.append([[p[0].real.hex(), p[0].imag.hex()] for p in pairs[:8]])
print(json.dumps(out))
"""


def test_long_vector_cgs2_gives_the_same_bits_in_processes_that_share_the_gpu():
    """The kernel-per-stage CGS2 of long vectors (``cgs_axpy_dot_kernel``: first projection and second dot product in one pass)
    forced onto the 30 k-unknown case, in four processes that time-share the GPU, several solves each: every solve of every
    process returns the same bits.  Round 4 met a read-after-write race in that kernel this way (all four wavefronts of a
    workgroup read ``w[i]`` behind the barrier, wavefront 0 wrote it): ranks of a sharded solve, whose replicated vectors must
    stay bit-identical, went out of step once in a few runs of a 1 M-unknown case."""
    import os
    import subprocess
    import sys

    root = str(Path(__file__).resolve().parents[1])
    env = {**os.environ, "LSA_KRYLOV_FUSED": "0"}
    procs = [subprocess.Popen([sys.executable, "-c", _REPRO_CHILD, root, "6"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(4)]
    results = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        results.append(json.loads([ln for ln in so.splitlines() if ln.startswith("[")][-1]))
    first = results[0][0]
    assert len(first) == 8
    for r in results:
        for solve in r:
            assert solve == first
