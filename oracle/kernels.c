/* ORACLE (test infrastructure, CPU only) -- scalar C restatements of the sparse kernels on the
 * shift-invert eigen path.  Built by oracle/Makefile into oracle/_build/liblsa_oracle.so and loaded with
 * ctypes by oracle/kernels.py.  Never linked or loaded by the product (lsa-fw_amd/).
 * See kernels_impl.h for what each function restates.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>

#define SCALAR double
#define FN(name) name##_f64
#define ABS(x) fabs(x)
#include "kernels_impl.h"
#undef SCALAR
#undef FN
#undef ABS

#define SCALAR double _Complex
#define FN(name) name##_c128
#define ABS(x) cabs(x)
#include "kernels_impl.h"
#undef SCALAR
#undef FN
#undef ABS

/* y = A x with a real matrix and a complex vector (M x in the complex-shift path) */
void orc_spmv_rc(int n, const int *rp, const int *ci, const double *v, const double _Complex *x, double _Complex *y) {
    for (int i = 0; i < n; ++i) {
        double _Complex s = 0;
        for (int p = rp[i]; p < rp[i + 1]; ++p) s += v[p] * x[ci[p]];
        y[i] = s;
    }
}

/* C = alpha A + beta M on one shared pattern (MatAXPY SAME_NONZERO_PATTERN), real inputs, complex out */
void orc_axpby_rc(long nnz, const double *a, const double *m, double alpha_re, double alpha_im, double beta_re,
                  double beta_im, double _Complex *c) {
    const double _Complex alpha = alpha_re + alpha_im * I, beta = beta_re + beta_im * I;
    for (long p = 0; p < nnz; ++p) c[p] = alpha * a[p] + beta * m[p];
}

/* Symbolic ILU(k): level-of-fill pattern (PETSc MatILUFactorSymbolic with -pc_factor_levels k).
 * Input CSR pattern with sorted columns; output pattern written into caller buffers of capacity `cap`
 * entries (returns the needed nnz; call twice if the first capacity was too small).  lev(i,j) =
 * min over k of lev(i,k) + lev(k,j) + 1; entries with lev <= levels are kept. */
long orc_iluk_symbolic(int n, const int *rp, const int *ci, int levels, long cap, int *orp, int *oci) {
    int *olev = (int *)malloc(sizeof(int) * (size_t)(cap > 0 ? cap : 1));
    int *next = (int *)malloc(sizeof(int) * (size_t)(n + 1));   /* sorted linked list over columns */
    int *lev = (int *)malloc(sizeof(int) * (size_t)n);
    int *odiag = (int *)malloc(sizeof(int) * (size_t)n);
    long nnz = 0;
    int overflow = 0;
    orp[0] = 0;
    for (int i = 0; i < n; ++i) {
        /* load row i into the list */
        int head = n, prev = -1;
        for (int p = rp[i]; p < rp[i + 1]; ++p) {
            int c = ci[p];
            lev[c] = 0;
            if (prev < 0) head = c; else next[prev] = c;
            prev = c;
        }
        if (prev >= 0) next[prev] = n;
        /* eliminate with previous rows k < i in ascending order */
        for (int k = head; k < i; k = next[k]) {
            int lk = lev[k];
            if (overflow) break;
            int ins = k; /* insertion cursor: list is sorted, row k's upper part is sorted too */
            for (long q = odiag[k] + 1; q < orp[k + 1]; ++q) {
                int j = oci[q];
                int l = lk + olev[q] + 1;
                if (l > levels) continue;
                /* find position of j */
                while (next[ins] < j) ins = next[ins];
                if (next[ins] == j) { if (lev[j] > l) lev[j] = l; }
                else { next[j] = next[ins]; next[ins] = j; lev[j] = l; }
                ins = j;
            }
        }
        /* store row */
        for (int c = head; c < n; c = next[c]) {
            if (nnz < cap) { oci[nnz] = c; olev[nnz] = lev[c]; if (c == i) odiag[i] = (int)nnz; }
            else overflow = 1;
            ++nnz;
        }
        orp[i + 1] = (int)nnz;
    }
    free(olev); free(next); free(lev); free(odiag);
    return nnz;
}
