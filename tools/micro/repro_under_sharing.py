"""Bitwise reproducibility of whole eigen-solves in processes that time-share the GPU (a race that is invisible on a GPU of one's
own shows up when wavefronts of other processes get in between): P processes, the same cases, every solve of every process must
return the same bits.   python tools/micro/repro_under_sharing.py [P] [solves] [more]"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CHILD = r"""
import json, sys
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/lsa-fw_amd"]
from synthetic import fem
from Solver.eigen import EigenSolver, EigensolverConfig
from Solver.utils import PreconditionerType, iSTType
out = {}
for case, k in (("S30k", 20), ("S120k", 20), ("S500k", 20), ("C40k", 10), ("C160k", 10)):
    cube = case.startswith("C")
    es = fem.cube_case(case) if cube else fem.cylinder_case(case)
    s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=k, atol=1e-10, ncv=4 * k), check_hermitian=False)
    s.solver.set_st_type(iSTType.SINVERT); s.solver.set_st_pc_type(PreconditionerType.LU)
    s.solver.set_target(fem.SIGMA_CUBE if cube else fem.SIGMA_RE50)
    out[case] = []
    for _ in range(int(sys.argv[2])):
        pairs = s.solve()
        out[case].append([[p[0].real.hex(), p[0].imag.hex()] for p in pairs[:k]])
    s.solver.release()
    print("done", case, file=sys.stderr, flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "more":   # the other inner solvers and arithmetic types of the path
    import numpy as np
    es = fem.cylinder_case("S5k")
    for name, kw, pc in (("S5k ilu-gmres", {}, PreconditionerType.ILU), ("S5k adjoint", {"adjoint": True}, PreconditionerType.LU)):
        s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=6, atol=1e-10, ncv=30), check_hermitian=False, **kw)
        s.solver.set_st_type(iSTType.SINVERT); s.solver.set_st_pc_type(pc); s.solver.set_target(fem.SIGMA_RE50)
        out[name] = []
        for _ in range(int(sys.argv[2])):
            pairs = s.solve()
            out[name].append([[p[0].real.hex(), p[0].imag.hex()] for p in pairs[:6]])
        s.solver.release()
        print("done", name, file=sys.stderr, flush=True)
    K, M, _b = fem.assemble_membrane(48, 48)
    s = EigenSolver(K, M, EigensolverConfig(num_eig=6, atol=1e-10, ncv=30), check_hermitian=False)
    s.solver.set_st_type(iSTType.SINVERT); s.solver.set_st_pc_type(PreconditionerType.LU); s.solver.set_target(0.0)
    out["membrane f64"] = []
    for _ in range(int(sys.argv[2])):
        pairs = s.solve()
        out["membrane f64"].append([[p[0].real.hex(), p[0].imag.hex()] for p in pairs[:6]])
    s.solver.release()
print(json.dumps(out))
"""

if __name__ == "__main__":
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    solves = sys.argv[2] if len(sys.argv) > 2 else "3"
    extra = sys.argv[3:4]
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, str(ROOT), solves] + extra, stdout=subprocess.PIPE, stderr=None, text=True) for _ in range(nproc)]
    outs = []
    for p in procs:
        so, _ = p.communicate(timeout=1500)
        assert p.returncode == 0
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    bad = 0
    for case in outs[0]:
        ref = outs[0][case][0]
        n_diff = sum(1 for o in outs for solve in o[case] if solve != ref)
        print(f"{case}: {len(outs) * len(outs[0][case])} solves in {len(outs)} processes, {n_diff} differ from the first", flush=True)
        bad += n_diff
    sys.exit(1 if bad else 0)
