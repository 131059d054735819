/* ORACLE (test infrastructure, CPU only): body of oracle/kernels.c, included once per scalar type.
 *
 * SCALAR is `double` or `double _Complex`; FN(name) appends the type suffix (_f64 / _c128).
 * These are plain-C restatements of the third-party natives the reference's eigen path runs implicitly
 * through petsc4py (PETSc is not vendored in /root/reference; nominal version 3.22.0, README.md:86-87):
 *   MatMult (AIJ)            <- EPS/ST apply, explicit at Solver/eigen2.py:174
 *   MatAXPY same-pattern     <- C = A - sigma M, Solver/eigen2.py:110-111
 *   PCSetUp(ILU, 0 levels)   <- Solver/utils.py:261-266 with PreconditionerType.ILU
 *   MatSolve (L then U)      <- PC apply
 * CSR with sorted column indices and a structurally present diagonal is assumed (checked by callers).
 */

/* y = A x */
void FN(orc_spmv)(int n, const int *rp, const int *ci, const SCALAR *v, const SCALAR *x, SCALAR *y) {
    for (int i = 0; i < n; ++i) {
        SCALAR s = 0;
        for (int p = rp[i]; p < rp[i + 1]; ++p) s += v[p] * x[ci[p]];
        y[i] = s;
    }
}

/* ILU(0), IKJ order, in place on a copy of the values; returns 0 or -(row+1) of the first tiny pivot.
 * `diag[i]` receives the position of the diagonal entry of row i.  A pivot with |u_ii| < shift_tol is
 * replaced by shift_tol * (u_ii/|u_ii|) (PETSc: -pc_factor_shift_type nonzero). */
int FN(orc_ilu0)(int n, const int *rp, const int *ci, SCALAR *v, int *diag, double shift_tol, int *nshift) {
    int *pos = (int *)malloc(sizeof(int) * (size_t)n);
    if (!pos) return -1000000000;
    for (int i = 0; i < n; ++i) pos[i] = -1;
    int shifted = 0, rc = 0;
    for (int i = 0; i < n; ++i) {
        int d = -1;
        for (int p = rp[i]; p < rp[i + 1]; ++p) {
            pos[ci[p]] = p;
            if (ci[p] == i) d = p;
        }
        if (d < 0) { rc = -(i + 1); break; }
        diag[i] = d;
        for (int p = rp[i]; p < d; ++p) {
            int k = ci[p];
            SCALAR lik = v[p] / v[diag[k]];
            v[p] = lik;
            for (int q = diag[k] + 1; q < rp[k + 1]; ++q) {
                int pj = pos[ci[q]];
                if (pj >= 0) v[pj] -= lik * v[q];
            }
        }
        double mag = ABS(v[d]);
        if (!(mag >= shift_tol) || mag == 0.0) {
            if (shift_tol <= 0.0) { rc = -(i + 1); }
            else {
                v[d] = (mag > 0.0) ? v[d] / mag * shift_tol : (SCALAR)shift_tol;
                ++shifted;
            }
        }
        for (int p = rp[i]; p < rp[i + 1]; ++p) pos[ci[p]] = -1;
        if (rc) break;
    }
    free(pos);
    if (nshift) *nshift = shifted;
    return rc;
}

/* x = U^{-1} L^{-1} b with the factors stored in one CSR (unit lower L strictly below the diagonal). */
void FN(orc_ilu_solve)(int n, const int *rp, const int *ci, const SCALAR *v, const int *diag, const SCALAR *b,
                       SCALAR *x) {
    for (int i = 0; i < n; ++i) {
        SCALAR s = b[i];
        for (int p = rp[i]; p < diag[i]; ++p) s -= v[p] * x[ci[p]];
        x[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
        SCALAR s = x[i];
        for (int p = diag[i] + 1; p < rp[i + 1]; ++p) s -= v[p] * x[ci[p]];
        x[i] = s / v[diag[i]];
    }
}

/* lower-only and upper-only halves (for checking the two SpTRSV kernels separately) */
void FN(orc_sptrsv_lower_unit)(int n, const int *rp, const int *ci, const SCALAR *v, const int *diag, const SCALAR *b,
                               SCALAR *x) {
    for (int i = 0; i < n; ++i) {
        SCALAR s = b[i];
        for (int p = rp[i]; p < diag[i]; ++p) s -= v[p] * x[ci[p]];
        x[i] = s;
    }
}

void FN(orc_sptrsv_upper)(int n, const int *rp, const int *ci, const SCALAR *v, const int *diag, const SCALAR *b,
                          SCALAR *x) {
    for (int i = n - 1; i >= 0; --i) {
        SCALAR s = b[i];
        for (int p = diag[i] + 1; p < rp[i + 1]; ++p) s -= v[p] * x[ci[p]];
        x[i] = s / v[diag[i]];
    }
}
