/* liblsa_hip.so -- C-ABI of the MI355X-native shift-invert eigen path for LSA-FW.
 *
 * The reference (ferdean/lsa-fw) has no native code of its own: the hot path is one Python call,
 *     Solver/eigen.py:136  ->  Solver/utils.py:270  ->  SLEPc.EPS.solve()
 * and every numeric step runs inside petsc4py/slepc4py.  This header is therefore the interface a
 * maintainer binds *instead of* slepc4py for that path (ctypes stub: INTEGRATION.md).  Each entry point
 * names the petsc4py/slepc4py call site in the reference it stands in for.
 *
 * Conventions
 *   - every function returns an int status: LSA_OK (0) or a negative lsa_status; no exception and no HIP
 *     error crosses the boundary; lsa_last_error(ctx) gives the text of the last failure;
 *   - host buffers are owned by the caller and only read/written during the call; device memory is owned
 *     by the library behind opaque handles and released by the matching *_destroy;
 *   - complex values are interleaved (re, im) doubles; CSR uses int32 row pointers / column indices,
 *     columns sorted inside each row, explicit zeros allowed, a structurally present diagonal is required
 *     for factorisation;
 *   - a context is bound to one GPU and one HIP stream and is not thread-safe.
 */
#ifndef LSA_HIP_H
#define LSA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { LSA_F64 = 0, LSA_C128 = 1 } lsa_dtype;

typedef enum {
    LSA_OK = 0,
    LSA_ERR_ARG = -1,        /* bad argument / shape / dtype            -> Python ValueError  (Solver/eigen.py:79-87) */
    LSA_ERR_HIP = -2,        /* HIP runtime failure (OOM, launch, ...)  -> RuntimeError                                 */
    LSA_ERR_ZERO_PIVOT = -3, /* factorisation hit a zero pivot          -> RuntimeError (tests/unit/Solver/test_eigen.py:272-281) */
    LSA_ERR_DIVERGED = -4,   /* inner solve did not reach rtol          -> RuntimeError (Solver/eigen2.py:179-181) */
    LSA_ERR_NONFINITE = -5,  /* NaN/Inf produced                        -> RuntimeError (Solver/eigen2.py:186-189) */
    LSA_ERR_TIMEOUT = -6,    /* a bounded device-side wait expired      -> RuntimeError                                 */
    LSA_ERR_COMM = -7,       /* RCCL failure                            -> RuntimeError                                 */
    LSA_ERR_OOM = -8         /* device memory exhausted (the only failure a factorisation may answer by a leaner method) */
} lsa_status;

typedef struct lsa_ctx lsa_ctx;
typedef struct lsa_mat lsa_mat;   /* CSR matrix on the device             (PETSc.Mat AIJ, FEM/utils.py:104)  */
typedef struct lsa_vec lsa_vec;   /* dense vector on the device           (PETSc.Vec,      FEM/utils.py:662)  */
typedef struct lsa_ilu lsa_ilu;   /* ILU(k) factors + triangular schedule (PETSc.PC ILU,   Solver/utils.py:261-266) */
typedef struct lsa_op lsa_op;     /* shift-invert operator (C^-1 M)       (SLEPc.ST SINVERT, Solver/utils.py:256-259) */
typedef struct lsa_krylov lsa_krylov; /* Arnoldi basis + recurrences      (SLEPc.BV + EPS Krylov-Schur, Solver/utils.py:270) */

/* counters filled by the solvers (all cumulative since creation of the object they belong to) */
typedef struct {
    int64_t op_applies;      /* shift-invert applies (outer Arnoldi steps)           */
    int64_t gmres_iters;     /* inner GMRES iterations                                */
    int64_t spmv_calls;      /* SpMV launches                                         */
    int64_t sptrsv_calls;    /* triangular-solve launches (L and U counted apart)     */
    double last_rel_res;     /* relative residual estimate of the last inner solve    */
    double max_rel_res;      /* worst one seen                                        */
    double seconds_factor;   /* wall seconds spent in symbolic + numeric factorisation */
    double seconds_solve;    /* wall seconds spent inside lsa_op_apply / lsa_krylov_extend */
    int32_t stagnated_solves; /* inner solves accepted at a stagnated true residual in (10 rtol, 1000 rtol]: the caller warns */
    int32_t pc_fallback;     /* 1 if the exact LU did not fit the device memory and ILU(k) + GMRES took its place */
    int32_t backward_accepted; /* direct solves whose ||b - C x|| / ||b|| missed rtol but whose backward error
                                  ||b - C x|| / (||C||_F ||x||) is <= 1e-12: shifts next to an eigenvalue */
    int32_t analysis_reused; /* 1 if the exact LU found its pattern-only analysis prepared (lsa_ndlu_prepare) or cached in the context */
    int32_t refined_solves;  /* direct solves that took a step of iterative refinement x += C^-1 (b - C x) because the first
                                answer missed rtol (Solver/eigen2.py:178-189 checks the same residual) */
    int32_t reserved0;
} lsa_stats;

/* ---- context ------------------------------------------------------------------------------------ */
int lsa_ctx_create(int device, lsa_ctx **out);
void lsa_ctx_destroy(lsa_ctx *ctx);
const char *lsa_last_error(const lsa_ctx *ctx);
int lsa_ctx_synchronize(lsa_ctx *ctx);
/* name of the GPU architecture the context runs on, e.g. "gfx950" */
const char *lsa_ctx_arch(const lsa_ctx *ctx);

/* ---- vectors ------------------------------------------------------------------------------------ */
int lsa_vec_create(lsa_ctx *ctx, int64_t n, int dtype, lsa_vec **out);
void lsa_vec_destroy(lsa_vec *v);
int lsa_vec_upload(lsa_ctx *ctx, lsa_vec *v, const void *host);
int lsa_vec_download(lsa_ctx *ctx, const lsa_vec *v, void *host);

/* ---- CSR matrices: iPETScMatrix.from_matrix / as_scipy_array (FEM/utils.py:183-220,585-588) -------- */
int lsa_csr_upload(lsa_ctx *ctx, int32_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                   const void *val, int dtype, lsa_mat **out);
void lsa_mat_destroy(lsa_mat *m);
int lsa_mat_download_values(lsa_ctx *ctx, const lsa_mat *m, void *host_val);
/* C = alpha*A + beta*B on one shared sparsity pattern: MatDuplicate + MatAXPY of ST sinvert
 * (explicit in Solver/eigen2.py:110-111).  alpha/beta are (re, im); out_dtype LSA_F64 needs real inputs
 * and zero imaginary parts.  LIFETIME: the result BORROWS A's device index arrays (row pointers, columns) and only owns
 * its values: A must outlive it (destroy the result first).  The Python binding keeps A alive for that reason. */
int lsa_csr_axpby(lsa_ctx *ctx, const lsa_mat *A, const lsa_mat *B, const double alpha[2], const double beta[2],
                  int out_dtype, lsa_mat **out);
/* y = A x: MatMult (Solver/eigen2.py:174).  Real matrix with complex vectors is supported. */
int lsa_spmv(lsa_ctx *ctx, const lsa_mat *A, const lsa_vec *x, lsa_vec *y);
/* y = A^T x or A^H x (conj != 0) without a second copy of the values: a transposed INDEX (row pointers, rows, positions of the
 * values) is built once per pattern and shared by the matrices of that pattern; additions in a fixed order.  The adjoint
 * eigenproblem of Sensitivity/__init__.py:47-57,247-248 */
int lsa_spmv_transpose(lsa_ctx *ctx, const lsa_mat *A, int conj, const lsa_vec *x, lsa_vec *y);
/* Which kernel lsa_spmv launches for this matrix and vector type (template arguments included) and the bytes that
 * kernel moves per launch: values + column indices (2-byte offsets from the row's first column when the compressed form
 * is in use) + row pointers + one read of x and one write of y.  The numerator of an HBM-roofline fraction. */
int lsa_spmv_info(lsa_ctx *ctx, const lsa_mat *A, int xdtype, char *kernel, int32_t kernel_len, int64_t *bytes_moved);
/* Launch y = A x `iters` times back to back on the context's stream, bracketed by HIP events on that
 * stream; *avg_ms = mean duration of one launch.  This is the measurement bench.py's roofline uses. */
int lsa_spmv_time(lsa_ctx *ctx, const lsa_mat *A, const lsa_vec *x, lsa_vec *y, int iters, double *avg_ms);

/* ---- ILU(k): PC ILU of the ST's KSP (Solver/utils.py:261-266, PreconditionerType.ILU) ------------ */
/* levels = level of fill (0 = ILU(0)); pivots with |u_ii| < shift_tol are replaced by shift_tol*sign
 * (PETSc -pc_factor_shift_type nonzero); shift_tol = 0 makes a zero pivot an error.  Symbolic analysis
 * and the dependency schedule are computed on the host, the numeric factorisation on the device. */
int lsa_ilu_create(lsa_ctx *ctx, const lsa_mat *C, int levels, double shift_tol, lsa_ilu **out);
void lsa_ilu_destroy(lsa_ilu *pc);
/* Triangular-solve algorithm: 0 = one launch per dependency level, 1 = sync-free (one launch, row hand-offs through
 * the solution vector), 2 = blocked (diagonal blocks of block_size rows inverted into dense triangles on the device,
 * 2 launches per block replayed from a hipGraph).  lsa_ilu_create picks 2 for narrow dependency DAGs (2D FEM) when
 * the 2 * n * block_size scalars fit, else 1.  All three give the same solve up to summation order. */
int lsa_ilu_set_algorithm(lsa_ctx *ctx, lsa_ilu *pc, int algo, int32_t block_size);
/* which: 0 = x = L^-1 b (unit lower), 1 = x = U^-1 b, 2 = x = U^-1 L^-1 b (MatSolve) */
int lsa_ilu_solve(lsa_ctx *ctx, lsa_ilu *pc, int which, const lsa_vec *b, lsa_vec *x);
/* `iters` back-to-back solves bracketed by HIP events on the context's stream; *avg_ms = mean per solve */
int lsa_ilu_solve_time(lsa_ctx *ctx, lsa_ilu *pc, int which, const lsa_vec *b, lsa_vec *x, int iters, double *avg_ms);
/* introspection for tests: nnz of the factor pattern, number of dependency levels (lower, upper),
 * number of shifted pivots */
int lsa_ilu_info(const lsa_ilu *pc, int64_t *nnz, int32_t *levels_lower, int32_t *levels_upper, int32_t *nshift);
/* copy the factor out (CSR, L strictly below the diagonal with unit diagonal implied, U on and above) */
int lsa_ilu_download(lsa_ctx *ctx, const lsa_ilu *pc, int32_t *rowptr, int32_t *col, void *val);

/* ---- nested-dissection multifrontal LU: PC LU of the ST's KSP (.examples/eigenvalues.py:100;
 * Sensitivity/__init__.py:182,260) ---------------------------------------------------------------------------------------
 * The sparse direct solver PETSc's PC LU stands for, rebuilt for the device: elimination forest of dense fronts from a
 * nested dissection of the pattern's graph, every front's pivot block inverted explicitly, so that a solve is two sweeps
 * over the tree levels with one dense mat-vec per node and sweep.  C may be in any order (the dissection is internal). */
typedef struct lsa_nd_sym lsa_nd_sym;  /* analysis of a pattern (host only)  */
typedef struct lsa_ndlu lsa_ndlu;      /* factorisation resident in HBM      */
/* Host-only analysis of a square CSR pattern (no GPU needed): ordering, elimination forest, front index lists.
 * leaf_size <= 0 picks 128 unknowns per leaf subdomain.  constraint: NULL, or n flags marking the unknowns whose diagonal
 * is numerically zero (the pressure rows of a saddle-point matrix): they are eliminated after all their neighbours, which
 * keeps every pivot block non-singular.  The handle is returned on failure too (for lsa_nd_sym_error). */
int lsa_nd_analyse(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t leaf_size, const int8_t *constraint, lsa_nd_sym **out);
/* Ordering only: the dissection's elimination order and forest, without the index tables.  lsa_nd_sym_export then fills
 * perm, node_start, parent and level (front_size = own size, idx untouched).  A caller that permutes its matrices by perm
 * and hands the forest back (first = node_start[t], size = node_start[t + 1] - node_start[t]: lsa_ndlu_prepare_tree) runs the
 * whole iteration in the elimination order: a node's own unknowns are contiguous in every vector, and the sweeps of the
 * factorisation address them without index lists. */
int lsa_nd_order(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t leaf_size, const int8_t *constraint, lsa_nd_sym **out);
/* The same analysis for a tree the caller provides, optionally localised for one rank of a subtree-parallel
 * factorisation.  Node t owns the matrix indices [first[t], first[t] + size[t]) (nodes in any order in which parent[] can be
 * looked up; -1 = root); owner[t] (NULL on one rank) = the rank whose subtree the node belongs to, -1 = the replicated top
 * of the tree.  Rows that belong to no node must be empty (padding of the sharded block layout).  The tables then hold
 * this rank's own nodes, the top, and the other ranks' subtree roots as childless "ghost" nodes. */
int lsa_nd_analyse_tree(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t ntree, const int32_t *first, const int32_t *size,
                        const int32_t *parent, const int32_t *owner, int32_t rank, int32_t nranks, lsa_nd_sym **out);
const char *lsa_nd_sym_error(const lsa_nd_sym *h);
void lsa_nd_sym_destroy(lsa_nd_sym *h);
/* tree nodes, tree levels, largest front, total length of the front index lists, scalars one solve reads (sum of
 * m^2 + 2 m b over the nodes), scalars of all fronts (sum of f^2), multiply-adds of the numeric factorisation */
int lsa_nd_sym_info(const lsa_nd_sym *h, int32_t *ntree, int32_t *nlevels, int32_t *max_front, int64_t *index_entries,
                    int64_t *factor_entries, int64_t *front_entries, double *flops);
/* Device memory of a factorisation of this analysis, in bytes, before any GPU is touched (what lsa_ndlu_create will allocate;
 * scalar_bytes 8 or 16; work_budget_bytes: what the working fronts of one chunk may take, <= 0: a whole level): out[0] packed
 * factors sum(m^2 + 2 m b), out[1] working arena, out[2] update arena (out[3] of it: the exchange region of a forest cut over
 * ranks), out[4] sweep buffers, out[5] number of chunks, out[6] the largest front, out[7] index tables.  Optional per-node
 * copies of the plan (tests): upd_off[ntree], work_off[ntree] (scalars), chunk_of[ntree] (-1: not factored here). */
int lsa_nd_sym_memory(const lsa_nd_sym *h, int32_t scalar_bytes, int64_t work_budget_bytes, int64_t *out, int64_t *upd_off, int64_t *work_off,
                      int32_t *chunk_of);
/* copies of the analysis (any pointer may be NULL): perm[n] (elimination order -> original index), node_start[ntree + 1],
 * parent[ntree] (-1 = root), level[ntree], front_size[ntree], idx[index_entries] (front index lists, original numbering) */
int lsa_nd_sym_export(const lsa_nd_sym *h, int32_t *perm, int32_t *node_start, int32_t *parent, int32_t *level, int32_t *front_size,
                      int32_t *idx);
/* the index tables the device kernels walk, for tests: cmap[sum b] (position of each boundary unknown in the parent's
 * front), gptr[sum (f + 1)] / gidx[sum b] (gather lists of the upward sweep), asm_dst[nnz] (front-buffer slot of every
 * matrix entry), lvl_ptr[nlevels + 1] / lvl_nodes[ntree] (nodes by level) */
int lsa_nd_sym_export_tables(const lsa_nd_sym *h, int32_t *cmap, int32_t *gptr, int32_t *gidx, int64_t *asm_dst, int32_t *lvl_ptr,
                             int32_t *lvl_nodes);
/* per kept node: kind[ntree] (1 = factored on this rank, 2 = replicated top, 3 = another rank's subtree root, 4 = distributed
 * top node, see lsa_nd_sym_export_top), front_off /
 * u_off[ntree + 1] (offsets into the front / update-vector buffers; the subtree roots lie in per-rank slots at the start),
 * asm_src[scalars[3]] (matrix entry of every assembly slot), children_ptr[ntree + 1] / children_idx; scalars[6] = front
 * slot, update-vector slot, first replicated work level, assembly entries, ranks, rank */
int lsa_nd_sym_export_dist(const lsa_nd_sym *h, int32_t *kind, int64_t *front_off, int64_t *u_off, int32_t *asm_src, int32_t *children_ptr,
                           int32_t *children_idx, int64_t *scalars);
/* DISTRIBUTED top nodes (owner -2 in the forest handed to lsa_nd_analyse_tree; kind 4): every rank keeps the pivot block whole
 * and its own equal slice of the boundary rows (working front (m + brow) x f, update matrix brow x b, packed L (m + brow) x m)
 * and of the own rows of U (orows x b).  Per kept node: owner[ntree] (rank, -1 replicated top, -2 distributed top), rows[4 ntree]
 * = (brow0, brow, orow0, orows), exch[4 ntree] = (ux_base, ux_stride, xg_base, xg_stride): where the node's update entries and
 * finished own rows lie in the sweeps' per-level exchange regions (entry k of rank k / s's slot: base + (k / s) stride + k % s,
 * s = ceil(count / ranks)); totals[2] = entries of the update-vector buffer, of the own-row exchange buffer */
int lsa_nd_sym_export_top(const lsa_nd_sym *h, int32_t *owner, int32_t *rows, int64_t *exch, int64_t *totals);
/* Analysis (from C's host copy of the pattern; reused from the context when the last destroyed or prepared factorisation
 * had the same pattern) + numeric factorisation on the device.  If a pivot block comes out singular and C has zero
 * diagonal entries, the analysis is redone once with those unknowns as constraints.  LSA_ERR_ZERO_PIVOT when a pivot block
 * is singular to 1e-15 max|C| after that, LSA_ERR_OOM when the fronts do not fit. */
int lsa_ndlu_create(lsa_ctx *ctx, const lsa_mat *C, int32_t leaf_size, lsa_ndlu **out);
/* Analysis only, parked in the context: the pattern of P (any matrix with C's pattern, e.g. A) and the scalar type the
 * factors will have.  The next lsa_ndlu_create on that pattern then runs the numeric phase alone.  Lets a caller keep
 * the pattern-only work out of a timed solve (the Python layer calls it from prepare()).  constraint: as for
 * lsa_nd_analyse (NULL: none). */
int lsa_ndlu_prepare(lsa_ctx *ctx, const lsa_mat *P, int dtype, int32_t leaf_size, const int8_t *constraint);
/* The same with the caller's forest (arguments as for lsa_nd_analyse_tree on one rank; normally the forest of lsa_nd_order on
 * the matrix permuted by its perm). */
int lsa_ndlu_prepare_tree(lsa_ctx *ctx, const lsa_mat *P, int dtype, int32_t ntree, const int32_t *first, const int32_t *size,
                          const int32_t *parent);
/* Subtree-parallel form (one process per GPU, after lsa_comm_init*): every rank holds the whole matrix C and calls this with
 * the same forest (arguments as for lsa_nd_analyse_tree; the context's rank selects the localisation).  A rank factors
 * the subtrees it owns; the subtree roots' fronts are exchanged by one in-place all-gather; the top of the forest
 * (owner -1) is then factored redundantly by every rank.  A solve exchanges the subtree roots' update vectors (a few KB)
 * between the upward sweep over the own subtrees and the replicated top; it returns the entries of x that belong to this
 * rank's subtrees and to the top (the caller's all-gather of x completes it).  Failures are agreed on collectively. */
int lsa_ndlu_create_tree(lsa_ctx *ctx, const lsa_mat *C, int32_t ntree, const int32_t *first, const int32_t *size, const int32_t *parent,
                         const int32_t *owner, lsa_ndlu **out);
/* new values on the analysed pattern (a shift sweep: .examples/eigenvalues.py:97-108) */
int lsa_ndlu_refactor(lsa_ctx *ctx, lsa_ndlu *f, const lsa_mat *C);
void lsa_ndlu_destroy(lsa_ndlu *f);
/* x = C^-1 b */
int lsa_ndlu_solve(lsa_ctx *ctx, lsa_ndlu *f, const lsa_vec *b, lsa_vec *x);
/* x = C^-T b (conj = 0) or C^-H b (conj != 0) on the same factors: the sweeps of the transposed forest.  The adjoint
 * eigenproblem (Sensitivity/__init__.py:47-57,247-287 forms A^H, M^H explicitly and factorises again) needs no second
 * factorisation and no transposed matrix. */
int lsa_ndlu_solve_adjoint(lsa_ctx *ctx, lsa_ndlu *f, int conj, const lsa_vec *b, lsa_vec *x);
int lsa_ndlu_solve_time(lsa_ctx *ctx, lsa_ndlu *f, const lsa_vec *b, lsa_vec *x, int iters, double *avg_ms);
/* Inertia (numbers of negative, zero, positive eigenvalues) of a REAL SYMMETRIC C from its factorisation: what SLEPc's spectrum
 * slicing takes from the symmetric-indefinite factorisation behind EPS.setInterval / EPS_ALL (Solver/utils.py:248-254: the
 * number of eigenvalues of a definite pencil below sigma is the number of negative eigenvalues of A - sigma M).  Sum over the
 * tree nodes of the inertia of their pivot blocks, each evaluated on the host (O(m^3)): for the Hermitian problems of the
 * interval sweep.  Real factors on one rank only; the symmetry of C is the caller's statement. */
int lsa_ndlu_inertia(lsa_ctx *ctx, lsa_ndlu *f, int64_t *negative, int64_t *zero, int64_t *positive);
/* front_entries: scalars of the device buffers (packed factors + the working fronts of one chunk + the update arena);
 * apply_bytes: algorithmic bytes of one solve (every factor scalar once + the vectors); apply_launches: dependent
 * launches of one solve (two per tree level, less one: the roots have no downward step) */
int lsa_ndlu_info(const lsa_ndlu *f, int32_t *ntree, int32_t *nlevels, int32_t *max_front, int64_t *factor_entries,
                  int64_t *front_entries, int64_t *apply_bytes, int32_t *apply_launches, double *seconds_analyse,
                  double *seconds_numeric);

/* ---- GMRES: KSPSolve of the ST (reference default PREONLY+LU; north star: GMRES+ILU) ----------------- */
/* right-preconditioned restarted GMRES with CGS2; pc may be NULL.  x holds the initial guess on entry
 * when use_x0 != 0.  Returns LSA_ERR_DIVERGED if rtol is not reached within maxit iterations. */
int lsa_gmres(lsa_ctx *ctx, const lsa_mat *C, lsa_ilu *pc, const lsa_vec *b, lsa_vec *x, int use_x0, double rtol,
              int restart, int maxit, int32_t *iters, double *rel_res);

/* ---- shift-invert operator: ST SINVERT (Solver/utils.py:244-266; Solver/eigen2.py:109-201) ---------- */
typedef struct {
    int32_t ilu_levels;   /* level of fill of the preconditioner (default 0)                  */
    double ilu_shift;     /* pivot shift tolerance (0 = zero pivot is an error)              */
    double ksp_rtol;      /* inner GMRES relative tolerance                                  */
    int32_t ksp_restart;  /* GMRES restart length                                            */
    int32_t ksp_maxit;    /* GMRES iteration cap                                             */
    int32_t pc_type;      /* 0 = none, 1 = ILU(k), 2 = exact LU (nested-dissection multifrontal; falls back to ILU(k) + GMRES only
                             when it runs out of device memory).  (3, round 1's banded block LU, is a cross-check library now.) */
    double antishift[2];  /* mode 2 only: nu of the Cayley transform (re, im); SLEPc's default is nu = sigma */
} lsa_op_options;

/* Builds C = A - sigma*M (complex if sigma has an imaginary part or A/M are complex), factors it, and
 * allocates the inner-solver workspace.  M may be NULL (standard problem, M = I).
 * mode: 0 = shift-invert  y = (A - sigma M)^-1 M x     (iSTType.SINVERT)
 *       1 = shift         y = M^-1 (A - sigma M) x     (iSTType.SHIFT; M = NULL gives y = (A - sigma I) x)
 *       2 = Cayley        y = (A - sigma M)^-1 (A + nu M) x   (iSTType.CAYLEY, nu = opts->antishift) */
int lsa_op_create(lsa_ctx *ctx, const lsa_mat *A, const lsa_mat *M, const double sigma[2], int mode,
                  const lsa_op_options *opts, lsa_op **out);
void lsa_op_destroy(lsa_op *op);
int lsa_op_apply(lsa_ctx *ctx, lsa_op *op, const lsa_vec *x, lsa_vec *y);
int lsa_op_stats(const lsa_op *op, lsa_stats *out);
/* Adjoint form of a shift-invert operator on the SAME factors: y = Kfac^-H Kmul^H x = (A - sigma M)^-H M^H x, the operator of
 * the adjoint eigenproblem (A^H, M^H) at the target conj(sigma) (Sensitivity/__init__.py:230-311, which forms the two
 * transposes and factorises again).  Transposed sweeps of the nested-dissection LU, transposed SpMV; one rank, or the
 * subtree-parallel layout of lsa_op_create_dist (same exchanges as the forward solve; the transposed products use the whole
 * matrices every rank holds). */
int lsa_op_set_adjoint(lsa_ctx *ctx, lsa_op *op, int on);
/* Projected operator  y = P Kfac^-1 Kmul x  with P = diag(keep): keep[i] in {0, 1}, host array of n doubles (NULL
 * removes the projection).  Stands in for the velocity-subspace projection of ArpackEigenSolver's matvec
 * (Solver/eigen2.py:164-201: pressure dofs zeroed before and after the inner solve); the input side of the
 * projection is the caller's: Krylov vectors are outputs of this operator, the start vector is masked on the host. */
int lsa_op_set_projection(lsa_ctx *ctx, lsa_op *op, const double *keep);

/* ---- Krylov basis: BV + Arnoldi recurrences of EPS Krylov-Schur (SLEPc.EPS.solve, Solver/utils.py:270) -- */
/* Basis of up to ncv+1 complex vectors of length n, resident in HBM, column-major. */
int lsa_krylov_create(lsa_ctx *ctx, lsa_op *op, int32_t ncv, lsa_krylov **out);
void lsa_krylov_destroy(lsa_krylov *k);
/* perm[i] = the caller's index of row i of the basis (the solve runs in a permuted numbering): lsa_krylov_ritz_vectors then
 * returns its vectors in the caller's numbering, rows scattered on the device.  NULL removes it. */
int lsa_krylov_set_row_permutation(lsa_ctx *ctx, lsa_krylov *k, const int32_t *perm);
/* v_0 = v / ||v||  (host complex vector of length n) */
int lsa_krylov_set_start(lsa_ctx *ctx, lsa_krylov *k, const void *host_v);
/* v_j = host vector orthonormalised (CGS2) against v_0..v_{j-1}: used to continue after an exact breakdown
 * (invariant subspace found, e.g. repeated eigenvalues) with a fresh direction; j = 0 equals set_start. */
int lsa_krylov_inject(lsa_ctx *ctx, lsa_krylov *k, int32_t j, const void *host_v);
/* Arnoldi steps j = j0 .. j1-1:  w = OP v_j;  CGS2 against v_0..v_j;  v_{j+1} = w/||w||.
 * H is the caller's (ncv+1) x ncv column-major complex Hessenberg; columns j0..j1-1 are written.
 * Returns LSA_OK and *breakdown = step index if ||w|| underflowed (invariant subspace), else -1. */
int lsa_krylov_extend(lsa_ctx *ctx, lsa_krylov *k, int32_t j0, int32_t j1, void *H, int32_t ldh, int32_t *breakdown);
/* Krylov-Schur truncation: V[:, 0:knew] = V[:, 0:m] Q (Q is m x knew column-major complex on the host)
 * and V[:, knew] = V[:, m]. */
int lsa_krylov_restart(lsa_ctx *ctx, lsa_krylov *k, int32_t m, int32_t knew, const void *Q, int32_t ldq);
/* X = V[:, 0:m] Y, Y m x nvec on the host; X (n x nvec column-major complex) is written to the host.  normalise: 0 = as they
 * are; bit 0 = each column scaled to unit 2-norm (SLEPc convention, Solver/utils.py:309); bit 1 = and rotated to a canonical
 * phase (its entry of largest magnitude real and positive: a real eigenvector comes out real, the test the reference's real
 * build makes at Solver/utils.py:280-291), with the 2-norms of the columns' imaginary parts kept for lsa_krylov_imag_norms. */
int lsa_krylov_ritz_vectors(lsa_ctx *ctx, lsa_krylov *k, int32_t m, int32_t nvec, const void *Y, int32_t ldy,
                            int normalise, void *X);
/* 2-norms of the imaginary parts of the columns of the last lsa_krylov_ritz_vectors call with normalise bit 1 (nvec <= its nvec) */
int lsa_krylov_imag_norms(const lsa_krylov *k, int32_t nvec, double *out);
/* residual check of Solver/eigen2.py:48-56 on the device:
 * res[i] = ||A x_i - lam_i M x_i|| / (||A x_i|| + |lam_i| ||M x_i|| + 1e-16), X on the host (n x nvec). */
int lsa_eig_residuals(lsa_ctx *ctx, const lsa_mat *A, const lsa_mat *M, int32_t nvec, const void *lam, const void *X,
                      double *res);

/* ---- the whole eigen-solve behind one call: SLEPc.EPS.solve() (Solver/utils.py:268-270) ------------------------------------------
 * Krylov-Schur outer iteration (SLEPc's default EPS: expand to ncv vectors, Schur form of the projected matrix with the wanted
 * Ritz values first, residual estimates |b^H y|, relative convergence test EPS_CONV_REL, restart with nconv + (m - nconv)/2
 * vectors) with the dense ncv x ncv algebra done by the library itself on the host (in-tree complex QR algorithm: no LAPACK
 * is needed by a consumer of this header).  lsa_hip/krylov_schur.py is the same loop in Python over LAPACK, kept as the test
 * double and for `which` policies that need callbacks. */
typedef enum {  /* EPSWhich on the eigenvalues lambda of the pencil (iEpsWhich, Solver/utils.py:152-187) */
    LSA_WHICH_LARGEST_MAGNITUDE = 1,
    LSA_WHICH_LARGEST_REAL = 3,
    LSA_WHICH_SMALLEST_REAL = 4,
    LSA_WHICH_LARGEST_IMAGINARY = 5,
    LSA_WHICH_SMALLEST_IMAGINARY = 6,
    LSA_WHICH_TARGET_MAGNITUDE = 7,
    LSA_WHICH_TARGET_REAL = 8,
    LSA_WHICH_TARGET_IMAGINARY = 9
} lsa_which;
typedef struct {
    int32_t nev;            /* eigenpairs wanted                                                                       */
    int32_t max_restarts;   /* EPS max_it                                                                              */
    double tol;             /* relative tolerance on the residual estimate of a Ritz pair                             */
    int32_t which;          /* lsa_which                                                                               */
    int32_t transform;      /* how a Ritz value theta of the operator maps to lambda: 0 = sigma + 1/theta (shift-invert),
                               1 = theta + sigma (shift), 2 = (sigma theta + nu) / (theta - 1) (Cayley)               */
    double sigma[2];        /* the operator's shift                                                                    */
    double antishift[2];    /* nu of the Cayley transform                                                              */
    double target[2];       /* target of the TARGET_* policies                                                         */
    uint64_t seed;          /* of the start vector (when v0 is NULL) and of the fresh directions after a breakdown     */
    double keep_fraction;   /* share of the unconverged part of the basis kept at a restart (0.5)                      */
} lsa_ks_options;
typedef struct {
    int32_t nconv;          /* converged pairs (may exceed nev, like EPS.getConverged())                               */
    int32_t nout;           /* pairs written: min(nconv, max_out)                                                      */
    int32_t restarts;
    int32_t pad;
    int64_t op_applies;
    double next_unconverged; /* relative residual estimate of the first pair that missed the tolerance                 */
    double seconds_expand;   /* wall time inside the Arnoldi expansions (device work + waiting for it)                 */
    double seconds_dense;    /* ... in the host's dense algebra on the projected matrix (Schur forms, reordering)      */
    double seconds_restart;  /* ... in the basis updates V <- V Q and the final Ritz vectors                           */
} lsa_ks_result;
/* n and ncv of a basis */
int lsa_krylov_shape(const lsa_krylov *k, int64_t *n, int32_t *ncv);
/* rows of a matrix handle (local rows of a shard) */
int64_t lsa_mat_rows(const lsa_mat *m);
/* The outer iteration on an existing basis (any operator: shift-invert, shift, Cayley, projected, adjoint, sharded).
 * v0: host start vector (n complex) or NULL (random from opts->seed); mask: NULL or n 0/1 doubles applied to every injected
 * vector (projected operators, padding of the sharded layout).  Outputs, wanted first: theta_out / lambda_out[max_out]
 * (complex), X_out (n x max_out complex column-major, unit 2-norm, canonical phase; NULL: no vectors), est_out[max_out]
 * relative residual estimates.  Returns LSA_OK also when fewer than nev pairs converged in max_restarts (see result). */
int lsa_krylov_solve(lsa_ctx *ctx, lsa_krylov *k, const lsa_ks_options *opts, const void *v0, const double *mask, int32_t max_out,
                     void *theta_out, void *lambda_out, void *X_out, double *est_out, lsa_ks_result *result);
/* Eigenpairs of A x = lambda M x nearest sigma in ONE call: builds and factors A - sigma M (lsa_op_create, mode 0), allocates
 * the basis, iterates, writes the pairs and releases everything.  ncv <= 0: max(2 nev, nev + 15) (SLEPc's default).  row_perm:
 * NULL, or perm[i] = the caller's index of row i (the matrices were uploaded in a permuted order, e.g. that of lsa_nd_order):
 * the vectors come back in the caller's numbering.  stats: NULL or the operator's counters. */
int lsa_eigs_sinvert(lsa_ctx *ctx, const lsa_mat *A, const lsa_mat *M, const double sigma[2], int32_t nev, int32_t ncv, double tol,
                     int32_t max_restarts, const lsa_op_options *opts, const void *v0, const int32_t *row_perm, int32_t max_out,
                     void *lambda_out, void *X_out, double *est_out, lsa_ks_result *result, lsa_stats *stats);
/* The dense kernels of that iteration, for tests (host only; complex column-major):
 *   lsa_dense_schur: A = Q T Q^H, T (upper triangular) over A, Q written; LSA_ERR_DIVERGED if the QR algorithm stalls
 *   lsa_dense_schur_reorder: the diagonal entries with select[k] != 0 move to the leading block (orders kept), T and Q updated
 *   lsa_dense_tri_eigenvectors: right eigenvectors of the upper triangular T, unit 2-norm columns of S */
int lsa_dense_schur(int32_t n, void *A, int32_t lda, void *Q, int32_t ldq);
/* Inertia of a dense real symmetric matrix (column-major, both triangles read and symmetrised; host only): Bunch-Kaufman diagonal
 * pivoting; pivots below tol_rel * max|A| count as zero.  The building block of lsa_ndlu_inertia. */
int lsa_dense_sym_inertia(int32_t n, const double *A, int32_t lda, double tol_rel, int64_t *negative, int64_t *zero, int64_t *positive);
int lsa_dense_schur_reorder(int32_t n, void *T, int32_t ldt, void *Q, int32_t ldq, const int32_t *select, int32_t *nselected);
int lsa_dense_tri_eigenvectors(int32_t n, const void *T, int32_t ldt, void *S, int32_t lds);

/* ---- MatrixMarket reader (host only): the A.mtx / M.mtx stage boundary --------------------------------------------
 * Stands in for scipy.io.mmread + the per-entry setValue loop of iPETScMatrix.from_path / from_matrix
 * (FEM/utils.py:143-147,208-215).  Coordinate format; general / symmetric / hermitian / skew-symmetric; real / integer /
 * complex / pattern; explicit zeros kept, duplicates summed, columns sorted.  Needs no GPU. */
typedef struct lsa_mm lsa_mm;
int lsa_mm_open(const char *path, lsa_mm **out, int32_t *nrows, int32_t *ncols, int64_t *nnz, int *is_complex);
int lsa_mm_read_csr(const lsa_mm *h, int32_t *rowptr, int32_t *col, void *val);
const char *lsa_mm_error(const lsa_mm *h);
void lsa_mm_close(lsa_mm *h);

/* ---- multi-GPU (one process per GPU; rows of C, M and the factors are sharded, the basis replicated) ---- */
/* RCCL bootstrap: rank 0 calls lsa_comm_unique_id, the launcher broadcasts the 128 bytes (e.g. with
 * torch.distributed), every rank then calls lsa_comm_init. */
int lsa_comm_unique_id(void *id128);
int lsa_comm_init(lsa_ctx *ctx, int nranks, int rank, const void *id128);
/* The same layout with a host-staged exchange instead of RCCL: the library copies this rank's block to a host buffer of
 * nranks * bytes_per_rank bytes (block r at r * bytes_per_rank), calls fn, which must fill the other ranks' blocks
 * (returning 0), and copies the buffer back to the device.  For tests and rehearsals with several ranks on ONE GPU (RCCL
 * refuses two ranks on a device) and for launchers whose only transport is a host one (torch.distributed over gloo). */
typedef int (*lsa_host_allgather_fn)(void *host_buf, int64_t bytes_per_rank, void *user);
int lsa_comm_init_host(lsa_ctx *ctx, int nranks, int rank, lsa_host_allgather_fn fn, void *user);
/* all-gathers issued on this context and the bytes this rank received through them */
int lsa_comm_stats(const lsa_ctx *ctx, int64_t *calls, int64_t *bytes_received);
/* The RCCL plumbing exercised with ONE rank (what a one-GPU box can run): open the library, unique id, a communicator of
 * one rank on the context's device, one in-place ncclAllGather of `bytes` bytes on the context's stream, compare, destroy.
 * Leaves the context's own communicator untouched.  LSA_OK, or LSA_ERR_COMM with the failing step in lsa_last_error. */
int lsa_comm_selftest(lsa_ctx *ctx, int64_t bytes);
/* Row-block shard of a global CSR: this rank owns rows [row0, row1); x and y of lsa_spmv stay global-length
 * and replicated, each rank computes its rows and the blocks are exchanged with ncclAllGather.
 * Padded block layout: the global index space is nranks equal blocks of B_pad = n_global / nranks slots, rank r owns
 * [r*B_pad, r*B_pad + rows_r); the unused tail of a block is padding (zero in every vector, referenced by no column),
 * so the all-gather moves equal counts.  Column indices are expressed in this padded space. */
int lsa_csr_upload_shard(lsa_ctx *ctx, int32_t n_global, int32_t row0, int32_t row1, int64_t nnz_local,
                         const int32_t *rowptr_local, const int32_t *col, const void *val, int dtype, lsa_mat **out);
/* Shift-invert operator on row shards (PETSc's parallel default bjacobi + ilu): A_rows / M_rows are this
 * rank's shards (lsa_csr_upload_shard), A_diag / M_diag its square diagonal blocks in local numbering
 * (lsa_csr_upload).  C's rows and the ILU(k) of C's diagonal block stay on the rank; every SpMV and every
 * preconditioner apply ends with one in-place all-gather; the Krylov bases are replicated, so dot products need no
 * collective and the Hessenberg matrices are bit-identical on every rank. */
int lsa_op_create_sharded(lsa_ctx *ctx, const lsa_mat *A_rows, const lsa_mat *M_rows, const lsa_mat *A_diag,
                          const lsa_mat *M_diag, const double sigma[2], int mode, const lsa_op_options *opts, lsa_op **out);

/* Shift-invert operator of the subtree-parallel layout: every rank holds the WHOLE matrices A, M in the padded block layout
 * (n_pad x n_pad, empty padding rows) and the forest of lsa_ndlu_create_tree, and solves with the subtree-parallel exact LU
 * (own subtrees, one small all-gather, replicated top, all-gather of the solution): no inner iteration.  The sparse products
 * run on the whole matrix on every rank when the pattern has at most 60 entries per row (cheaper than an exchange: two
 * collectives per operator apply), on the rank's rows [row0, row1) followed by an all-gather otherwise (four).  mode 0 or
 * 2, opts->pc_type 2. */
int lsa_op_create_dist(lsa_ctx *ctx, const lsa_mat *A, const lsa_mat *M, int32_t row0, int32_t row1, int32_t ntree, const int32_t *first,
                       const int32_t *size, const int32_t *parent, const int32_t *owner, const double sigma[2], int mode,
                       const lsa_op_options *opts, lsa_op **out);

#ifdef __cplusplus
}
#endif
#endif /* LSA_HIP_H */
