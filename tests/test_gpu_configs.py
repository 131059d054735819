"""GPU parity tests of the BASELINE.json configurations against committed golden fixtures (tests/golden/, written by
tests/golden/make_golden.py from the oracle) and, beyond the sizes the oracle answers in seconds, through
size-independent properties.

config 1 / 2: S30k pair at the Re = 50 target, k = 6 and k = 20;  config 3 (single-GPU part): S120k and S500k;
config 5: direct + adjoint pair at Re = 100.  (Row-sharded layouts: tests/test_gpu_sharded.py, tests/test_sharding_cpu.py;
3D: tests/test_gpu_3d.py.)
"""

import json
from pathlib import Path

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"


def _complex(pairs):
    return np.array([complex(a, b) for a, b in pairs])


def _solver(es, sigma, k, ncv, seed=0, atol=1e-10):
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=k, atol=atol, ncv=ncv, max_it=500), check_hermitian=False, seed=seed)
    s.solver.set_st_type(iSTType.SINVERT)
    s.solver.set_target(sigma)
    s.solver.set_st_pc_type(PreconditionerType.LU)
    return s


@pytest.fixture(scope="module")
def s30k_golden():
    from synthetic import fem

    gold = json.loads((GOLDEN / "cylinder_s30k_k20.json").read_text())
    es = fem.cylinder_case("S30k")
    assert es.n == gold["n"] and es.A.nnz == gold["nnz"]
    return es, complex(*gold["sigma"]), _complex(gold["eigenvalues"])


@pytest.mark.parametrize("k,ncv", [(20, 80), (6, 80)])
def test_s30k_eigenvalues_match_the_golden_fixture(s30k_golden, k, ncv):
    """BASELINE configs 2 (k = 20) and 1 (k = 6, the reference's CPU-runnable shape): rtol 1e-8 against the oracle's
    eigenvalues on the same assembled pair."""
    es, sigma, gold = s30k_golden
    s = _solver(es, sigma, k, ncv)
    pairs = s.solve()
    assert len(pairs) == k
    lam = np.array([p[0] for p in pairs])
    for r in gold[:k]:  # the fixture is sorted by distance to the target, like the solver's output
        assert np.min(np.abs(lam - r)) <= 1e-8 * abs(r)
    assert s.solver.residuals()[:k].max() <= 1e-8
    st = s.solver.stats
    assert st["pc_fallback"] == 0 and st["stagnated_solves"] == 0 and st["max_rel_res"] <= 1e-11 and st["analysis_reused"] == 1
    s.solver.release()


@pytest.mark.parametrize("case", ["S5k", "S30k"])
def test_direct_adjoint_pair_re100(case):
    """BASELINE config 5: the structural-sensitivity pair at Re = 100 (Sensitivity/__init__.py:158-311) against the
    oracle's direct and adjoint eigenvalues; a is a left eigenvector scaled so that a^H M v = 1."""
    from synthetic import fem
    from Sensitivity import EigenSensitivitySolver

    gold = json.loads((GOLDEN / "sensitivity_re100.json").read_text())
    target = complex(*gold["target"])
    rec = gold["cases"][case]
    es = fem.cylinder_case(case, re=gold["re"])
    assert es.n == rec["n"]
    sens = EigenSensitivitySolver(es.A, es.M, target=target, tol_direct=1e-10, tol_adjoint=1e-10)
    lam, v = sens.solve_direct_mode()
    assert abs(lam - complex(*rec["direct"])) <= 1e-8 * abs(lam)
    a = sens.solve_adjoint_mode()
    assert abs(sens._sigma_adj - complex(*rec["adjoint"])) <= 1e-8 * abs(lam)
    assert abs(sens._sigma_adj - np.conj(lam)) <= 1e-8 * abs(lam)
    assert np.vdot(a, es.M @ v) == pytest.approx(1.0, abs=1e-9)
    r = (es.A.conj().T @ a) - np.conj(lam) * (es.M.conj().T @ a)
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(es.A.conj().T @ a)
    rd = es.A @ v - lam * (es.M @ v)
    assert np.linalg.norm(rd) <= 1e-8 * (np.linalg.norm(es.A @ v) + abs(lam) * np.linalg.norm(es.M @ v))


@pytest.mark.parametrize("case", ["S120k", "S500k"])
def test_refined_meshes_residuals_and_start_vector_independence(case):
    """BASELINE config 3 on one GPU: k = 20 at the Re = 50 target on the refined meshes.  The oracle needs minutes there,
    so parity is checked through properties: true residuals, every inner solve direct and verified, and the same twenty
    eigenvalues from two different start vectors."""
    from synthetic import fem

    es = fem.cylinder_case(case)
    lams, vecs, ress = [], [], []
    for seed in (0, 11):
        s = _solver(es, fem.SIGMA_RE50, 20, 80, seed=seed, atol=1e-12)
        pairs = s.solve()
        assert len(pairs) == 20
        res = s.solver.residuals()[:20]
        assert res.max() <= 1e-8
        st = s.solver.stats
        assert st["gmres_iters"] == 0 and st["max_rel_res"] <= 1e-11 and st["pc_fallback"] == 0
        lams.append(np.array([p[0] for p in pairs]))
        vecs.append(np.column_stack([s.solver.get_eigenvector_array(i) for i in range(20)]))
        ress.append(res)
        s.solver.release()
    # How far two converged runs may differ is a property of each eigenvalue: its condition number kappa_i (left eigenvectors
    # by the adjoint path, helpers.eigenvalue_condition_numbers) times the residuals, |d lam| / |lam| <= 2 kappa_i (res_1 + res_2)
    # to first order; C_BOUND = 4 leaves a factor two for the second-order terms.  Measured: kappa from 8e5 (nearest the target)
    # to 1e11 (the dense outer branch of the refined meshes' spectra) -- no solver in double precision pins those to 1e-8.
    # The ten nearest agree to 1e-8 all the same, and that is asserted as well.
    C_BOUND = 4.0
    pick = np.array([int(np.argmin(np.abs(lams[1] - r))) for r in lams[0]])
    d = np.abs(lams[1][pick] - lams[0]) / np.abs(lams[0])
    kappa = helpers.eigenvalue_condition_numbers(es, lams[0], vecs[0])
    # (capped: for kappa ~ 1e11 the perturbation bound alone would allow differences of order one; measured 4e-15 ... 5e-8)
    bound = np.minimum(np.maximum(1e-8, C_BOUND * kappa * (ress[0] + ress[1][pick])), 1e-6)
    print(f"{case}: kappa {np.array2string(kappa, precision=1)}\n differences {np.array2string(d, precision=1)}\n bounds {np.array2string(bound, precision=1)}")
    assert np.all(np.isfinite(kappa)) and np.all(d <= bound), (d, bound)
    assert d[:10].max() <= 1e-8


def test_moving_the_target_keeps_what_was_prepared():
    """A new target on the same (A, M) re-uses the context, the uploaded matrices, the ordering and the cached analysis
    (only the pattern of A - sigma M and the scalar type of its factors matter to them); the answers are those of a solver
    built at the new target from scratch.  Real <-> complex shifts change the factors' type and prepare anew."""
    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    es = fem.cylinder_case("S5k")

    def make(target):
        s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=6, atol=1e-10, ncv=40), check_hermitian=False)
        s.solver.set_st_type(iSTType.SINVERT)
        s.solver.set_st_pc_type(PreconditionerType.LU)
        s.solver.set_target(target)
        return s

    s1 = make(fem.SIGMA_RE50)
    s1.solve()
    ctx = s1.solver._prepared["ctx"]
    sigma2 = fem.SIGMA_RE50 + (0.03 - 0.02j)
    s1.solver.set_target(sigma2)
    moved = s1.solve()
    assert s1.solver._prepared["ctx"] is ctx and s1.solver.stats["analysis_reused"] == 1
    fresh = make(sigma2).solve()
    assert len(moved) == len(fresh) == 6
    for (l1, v1), (l2, v2) in zip(moved, fresh):
        assert l1 == l2 and np.array_equal(v1.as_array(), v2.as_array())
    assert s1.solver.residuals().max() <= 1e-8
    s1.solver.set_target(0.05)  # a real shift: float64 factors, a different preparation
    real = s1.solve()
    assert s1.solver._prepared["ctx"] is not ctx and len(real) == 6 and s1.solver.residuals().max() <= 1e-8


def test_s500k_eigenvalues_match_the_golden_fixture():
    """BASELINE config 3's workload (the 500 k-unknown cylinder pair, k = 20) on one GPU against the oracle's eigenvalues on
    the same assembled pair (``tests/golden/make_golden_s500k.py``: ten minutes of ARPACK + SuperLU on one core)."""
    from synthetic import fem

    path = GOLDEN / "cylinder_s500k_k20.json"
    if not path.exists():
        pytest.skip("tests/golden/cylinder_s500k_k20.json has not been generated")
    gold = json.loads(path.read_text())
    es = fem.cylinder_case("S500k")
    assert es.n == gold["n"] and es.A.nnz == gold["nnz"]
    sigma, ref = complex(*gold["sigma"]), _complex(gold["eigenvalues"])
    s = _solver(es, sigma, 20, 80)
    pairs = s.solve()
    assert len(pairs) == 20
    lam = np.array([p[0] for p in pairs])
    pick = np.array([int(np.argmin(np.abs(lam - r))) for r in ref[:20]])
    diff = np.abs(lam[pick] - ref[:20]) / np.abs(ref[:20])
    res_gpu = s.solver.residuals()[:20][pick]
    # The fixture carries what the tolerance rests on (tests/golden/make_golden_s500k.py): ARPACK at tol = 0, the oracle's true
    # residuals and the condition number kappa_i of every eigenvalue (left eigenvectors from the adjoint problem on the same
    # SuperLU factors).  To first order two solvers agree to 2 kappa_i (res_gpu + res_oracle); C_BOUND = 4 leaves a factor two
    # for second-order terms.  1e-8 is asserted wherever that bound allows it, and on the ten nearest the target in any case.
    C_BOUND = 4.0
    kappa, res_ref = np.array(gold["kappa"]), np.array(gold["residuals"])
    # (capped at 1e-6: the six outermost modes have kappa up to 2.4e11; measured gaps 1e-7 ... 4e-7, the oracle's own, see below)
    bound = np.minimum(np.maximum(1e-8, C_BOUND * kappa * (res_gpu + res_ref)), 1e-6)
    print("relative differences to the oracle:", np.array2string(diff, precision=1), "\nbounds:", np.array2string(bound, precision=1),
          "\nkappa:", np.array2string(kappa, precision=1))
    assert np.all(diff <= bound), (diff, bound)
    assert diff[:10].max() <= 1e-8
    # the oracle's two-sided Rayleigh quotients a^H A v / a^H M v (error of second order in its residuals) tell which side an
    # outer eigenvalue's gap belongs to: the GPU's value is at least as close to them as ARPACK's own
    rq = _complex(gold["eigenvalues_two_sided_rq"])
    gpu_rq, ref_rq = np.abs(lam[pick] - rq) / np.abs(rq), np.abs(ref[:20] - rq) / np.abs(rq)
    print("GPU vs oracle RQ:", np.array2string(gpu_rq, precision=1), "\noracle vs its RQ:", np.array2string(ref_rq, precision=1))
    # measured: 2.5e-14 .. 9e-8 for the GPU against 9e-12 .. 4e-7 for ARPACK's own values -- the gap of the outer eigenvalues to
    # the oracle is the ORACLE's (kappa up to 2.4e11 times the rounding of its solves)
    assert np.all(gpu_rq <= np.maximum(1e-8, ref_rq)), (gpu_rq, ref_rq)
    assert s.solver.residuals()[:20].max() <= 1e-8
    st = s.solver.stats
    assert st["gmres_iters"] == 0 and st["pc_fallback"] == 0 and st["stagnated_solves"] == 0
    s.solver.release()
