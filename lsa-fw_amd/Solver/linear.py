"""Assembler-free linear solves on the HIP kernels of the eigen path.

Drop-in for the static ``LinearSolver.solve`` of ``/root/reference/Solver/linear.py:38-87``:
same signature and the same two supported ``ksp_type`` values,

* ``KSPType.PREONLY`` -- the reference pairs it with ``PreconditionerType.LU`` (``:60-64``): here the exact
  nested-dissection multifrontal LU on the device (``lsa_ndlu_*``; the dissection is internal, so no host ordering),
  with ILU(k)-preconditioned GMRES as fall-back only when the factors do not fit the device memory;
* ``KSPType.GMRES``   -- the reference runs it with ``PreconditionerType.NONE`` (``:60-64``, "< 200 iterations" on the
  cylinder Stokes system, ``doc/models/solver-linear.md:104``): here the device GMRES, unpreconditioned by default like
  the reference, or with ``pc=PreconditionerType.ILU`` to use the ILU(k) factors + blocked SpTRSV.

The instance methods of the reference's class need a dolfinx assembler and are out of scope.
"""

from __future__ import annotations

import logging
import time

import numpy as np
import scipy.sparse as sp

from FEM.utils import iPETScMatrix, iPETScVector

from .utils import KSPType, PreconditionerType, _permute, pivot_safe_rcm

logger = logging.getLogger(__name__)


class LinearSolver:
    """Container for the assembler-free linear solve (``Solver/linear.py:27-87``)."""

    @staticmethod
    def solve(A, b, *, ksp_type: KSPType, tol: float = 1e-12, rtol: float = 1e-8, max_it: int = 1_000,
              pc: PreconditionerType = PreconditionerType.NONE, ilu_levels: int = 2, restart: int = 1000, device: int = 0) -> iPETScVector:
        """Solve ``A x = b`` on the GPU.  ``tol`` (absolute) is accepted for signature compatibility; convergence is
        judged on the relative residual ``rtol`` like PETSc's default test with ``atol`` below it."""
        import lsa_hip

        if ksp_type not in (KSPType.PREONLY, KSPType.GMRES):
            raise ValueError("KSP type not supported.")
        A = A if isinstance(A, iPETScMatrix) else iPETScMatrix.from_matrix(A)
        mat = sp.csr_matrix(A.as_scipy_array())
        if mat.shape[0] != mat.shape[1]:
            raise ValueError(f"Operator A must be square, got shape {mat.shape}")
        rhs = b.as_array() if hasattr(b, "as_array") else np.asarray(b)
        if rhs.shape != (mat.shape[0],):
            raise ValueError(f"Right-hand side has shape {rhs.shape}, expected ({mat.shape[0]},)")
        cplx = np.iscomplexobj(mat.data) or np.iscomplexobj(rhs)
        vdt = np.complex128 if cplx else np.float64
        # ILU(k) wants a banded, pivot-safe order; the exact LU dissects the pattern itself
        perm = pivot_safe_rcm(mat) if (ksp_type is KSPType.GMRES and pc is not PreconditionerType.NONE and mat.shape[0] > 8) else np.arange(mat.shape[0])
        ctx = lsa_hip.Context(device)
        try:
            dA = lsa_hip.CsrMatrix.from_scipy(ctx, _permute(mat, perm))
            db = lsa_hip.DeviceVector.from_numpy(ctx, np.ascontiguousarray(rhs[perm], dtype=vdt))
            dx = lsa_hip.DeviceVector(ctx, mat.shape[0], vdt)
            t0 = time.time()
            its = 0
            if ksp_type is KSPType.PREONLY:
                try:
                    lsa_hip.NdLu(ctx, dA).solve(db, dx)
                except lsa_hip.LsaError as exc:
                    if exc.status != lsa_hip.LSA_ERR_OOM:  # only running out of device memory is answered by a leaner method
                        raise
                    logger.warning("exact LU does not fit the device memory (%s); using ILU(%d)-GMRES", exc, ilu_levels)
                    p2 = pivot_safe_rcm(mat)
                    dA2 = lsa_hip.CsrMatrix.from_scipy(ctx, _permute(mat, p2))
                    db2 = lsa_hip.DeviceVector.from_numpy(ctx, np.ascontiguousarray(rhs[p2], dtype=vdt))
                    its, _ = lsa_hip.gmres(ctx, dA2, lsa_hip.Ilu(ctx, dA2, levels=ilu_levels), db2, dx, rtol=min(rtol, 1e-12),
                                           restart=restart, maxit=max(max_it, 4000))
                    perm = p2
            else:
                pco = None if pc is PreconditionerType.NONE else lsa_hip.Ilu(ctx, dA, levels=ilu_levels)
                its, _ = lsa_hip.gmres(ctx, dA, pco, db, dx, rtol=rtol, restart=min(restart, max_it), maxit=max_it)
            logger.info("%s solve time: %.3f s (%d iterations)", ksp_type.name, time.time() - t0, its)
            x = np.empty(mat.shape[0], dtype=vdt)
            x[perm] = dx.numpy()
        finally:
            dA = db = dx = None
            import gc

            gc.collect()
            ctx.close()
        return iPETScVector(x)
