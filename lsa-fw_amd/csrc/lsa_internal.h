// Internal declarations shared by the translation units of liblsa_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lsa_hip.h"

// ---- scalar helpers ---------------------------------------------------------------------------------
struct cplx {
    double re, im;
};
static_assert(sizeof(cplx) == 16, "complex128 layout");

__host__ __device__ inline cplx make_cplx(double r, double i) { return cplx{r, i}; }

template <typename T>
struct scalar_traits;
template <>
struct scalar_traits<double> {
    static constexpr int dtype = LSA_F64;
    __host__ __device__ static double zero() { return 0.0; }
};
template <>
struct scalar_traits<cplx> {
    static constexpr int dtype = LSA_C128;
    __host__ __device__ static cplx zero() { return cplx{0.0, 0.0}; }
};

// acc += a * b for every (matrix scalar, vector scalar) pair the path needs
__host__ __device__ inline void fma_acc(double& acc, double a, double b) { acc = fma(a, b, acc); }
__host__ __device__ inline void fma_acc(cplx& acc, double a, cplx b) {
    acc.re = fma(a, b.re, acc.re);
    acc.im = fma(a, b.im, acc.im);
}
__host__ __device__ inline void fma_acc(cplx& acc, cplx a, cplx b) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
}
// acc += conj(a) * b
__host__ __device__ inline void fma_conj_acc(double& acc, double a, double b) { acc = fma(a, b, acc); }
__host__ __device__ inline void fma_conj_acc(cplx& acc, cplx a, cplx b) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(-a.im, b.re, acc.im);
}
__host__ __device__ inline double s_mul(double a, double b) { return a * b; }
__host__ __device__ inline cplx s_mul(double a, cplx b) { return cplx{a * b.re, a * b.im}; }
__host__ __device__ inline cplx s_mul(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__host__ __device__ inline double s_add(double a, double b) { return a + b; }
__host__ __device__ inline cplx s_add(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
__host__ __device__ inline double s_sub(double a, double b) { return a - b; }
__host__ __device__ inline cplx s_sub(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
__host__ __device__ inline double s_conj(double a) { return a; }
__host__ __device__ inline cplx s_conj(cplx a) { return cplx{a.re, -a.im}; }
__host__ __device__ inline double s_abs2(double a) { return a * a; }
__host__ __device__ inline double s_abs2(cplx a) { return a.re * a.re + a.im * a.im; }
__host__ __device__ inline double s_inv(double a) { return 1.0 / a; }
__host__ __device__ inline cplx s_inv(cplx a) {
    // Smith's algorithm: no overflow for large / tiny pivots
    if (fabs(a.re) >= fabs(a.im)) {
        double r = a.im / a.re, d = a.re + a.im * r;
        return cplx{1.0 / d, -r / d};
    }
    double r = a.re / a.im, d = a.re * r + a.im;
    return cplx{r / d, -1.0 / d};
}
__host__ __device__ inline cplx to_cplx(double a) { return cplx{a, 0.0}; }
__host__ __device__ inline cplx to_cplx(cplx a) { return a; }
__host__ __device__ inline void s_from(double& out, double re, double) { out = re; }
__host__ __device__ inline void s_from(cplx& out, double re, double im) { out = cplx{re, im}; }

// ---- objects behind the opaque handles ----------------------------------------------------------------
struct lsa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    std::string arch;
    int num_cu = 256;
    // pinned host scratch for small device->host reads (Hessenberg columns, flags)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // device scratch for reductions
    void* dscratch = nullptr;
    size_t dscratch_bytes = 0;
    // RCCL (multi-GPU); null when single-process
    void* comm = nullptr;
    int nranks = 1, rank = 0;
    // host-staged transport (tests and rehearsals on one GPU; RCCL refuses two ranks on one device)
    lsa_host_allgather_fn host_gather = nullptr;
    void* host_gather_user = nullptr;
    void* comm_stage = nullptr;
    size_t comm_stage_bytes = 0;
    int64_t comm_calls = 0, comm_bytes = 0;  // all-gathers issued, bytes received by this rank
    // the last destroyed nested-dissection LU (analysis + tables + buffers), reused when the next one has the same pattern and
    // shape: a shift sweep refactorises the same pattern once per sigma (.examples/eigenvalues.py:97-108)
    struct lsa_ndlu* nd_cache = nullptr;
    struct lsa_krylov* krylov_cache = nullptr;  // the last destroyed Krylov workspace (two bases, work vectors), reused by the next of the same shape
};
extern "C" void lsa_ndlu_drop_cache(lsa_ctx* ctx);  // ndlu.hip
extern "C" void lsa_krylov_drop_cache(lsa_ctx* ctx);  // solver.hip

struct lsa_vec {
    lsa_ctx* ctx;
    int64_t n;
    int dtype;
    void* d;
};

// immutable host array of int32 shared by reference count
struct SharedInts {
    std::shared_ptr<const std::vector<int32_t>> p;
    const int32_t* data() const { return p ? p->data() : nullptr; }
    size_t size() const { return p ? p->size() : 0; }
    const int32_t& operator[](size_t i) const { return (*p)[i]; }
    const std::vector<int32_t>& vec() const {
        static const std::vector<int32_t> none;
        return p ? *p : none;
    }
    void assign(const int32_t* b, const int32_t* e) { p = std::make_shared<const std::vector<int32_t>>(b, e); }
    bool same_object(const SharedInts& o) const { return p == o.p; }
    bool operator==(const SharedInts& o) const { return p == o.p || vec() == o.vec(); }
    bool operator!=(const SharedInts& o) const { return !(*this == o); }
};

// The transpose of a pattern in pull form (built on first use of the transposed product, shared by the matrices of one pattern):
// output c sums the entries q in [rp[c], rp[c + 1]): the value at position src[q] of the CSR arrays times x[ci[q]].
struct TransposeIndex {
    int32_t *rp = nullptr, *ci = nullptr, *src = nullptr;  // device: ncols + 1, nnz, nnz
    int32_t ncols = 0;
    int64_t nnz = 0;
    ~TransposeIndex() {
        for (void* p : {(void*)rp, (void*)ci, (void*)src})
            if (p) (void)hipFree(p);
    }
};

struct PatternShared {  // what the matrices of one pattern build once and share (a slot, so that whoever builds it first fills it for all)
    std::shared_ptr<TransposeIndex> tr;
};

struct lsa_mat {
    lsa_ctx* ctx;
    int32_t n;        // rows held locally
    int32_t ncols;    // global column count
    int32_t row0;     // first global row of this shard (0 when unsharded)
    int64_t nnz;
    int dtype;
    int32_t* rp;      // device, n+1
    int32_t* ci;      // device, nnz
    void* val;        // device, nnz
    bool owns_index;  // false when the index arrays are shared with another matrix
    bool owns_values = true;  // false for a row view (mat_row_view): every device array belongs to the viewed matrix
    // host copy of the pattern (analysis phases): shared, not copied, between the matrices that have one pattern (A, M and the
    // C = A - sigma M built for every solve), together with its hash (nd_pattern_hash, computed on first use: the cache of the
    // nested-dissection LU is matched by it once per solve -- hashing 15 M indices anew took 18 ms of a 0.36 s solve)
    mutable SharedInts h_rp, h_ci;
    mutable std::shared_ptr<uint64_t> h_hash;
    mutable std::shared_ptr<PatternShared> h_shared;  // (shared like the hash: A, M and every C = A - sigma M built on their pattern)
    // compressed column indices for the SpMV (built on first use): col = cbase[row] + ci16[p] when every row spans
    // fewer than 65 536 columns (banded FEM matrices do); ci16_state: 0 = not tried, 1 = available, -1 = does not fit
    mutable uint16_t* ci16 = nullptr;
    mutable int32_t* cbase = nullptr;
    mutable int ci16_state = 0;
    // row groups for the SpMV (built on first use): consecutive rows with one and the same column pattern (the unknowns of
    // a mesh node) share their column indices and their gathers of x; grp_start holds (first row, rows, first entry, entries per row) of
    // group g (at most 4).  grp_state: 0 = not tried, 1 = available, -1 = not worth it (mean group size < 1.5)
    mutable int32_t* grp_start = nullptr;
    mutable int32_t ngroups = 0;
    mutable int grp_state = 0;
    bool owns_extras = true;  // false when ci16 / cbase / grp_start are borrowed from the matrix whose pattern this one shares
};

int lsa_set_error(lsa_ctx* ctx, int code, const char* fmt, ...);
// rows [r0, r1) of a square matrix as a shard (n = r1 - r0, row0 = r0): borrows every array, must not outlive `full`
lsa_mat* mat_row_view(const lsa_mat* full, int32_t r0, int32_t r1);
void comm_release(lsa_ctx* ctx);  // comm.hip
int k_agree_status(lsa_ctx* ctx, int rc);  // comm.hip: collective agreement on a status (returns rc on one rank)
int k_agree_min_i64(lsa_ctx* ctx, int64_t* value);  // comm.hip: the smallest of the ranks' values (collective)
int k_agree_in_step(lsa_ctx* ctx, const char* where);  // comm.hip: error on every rank unless all ranks have made the same number of exchanges (collective)
int lsa_ensure_scratch(lsa_ctx* ctx, size_t dbytes, size_t hbytes);

#define LSA_HIP_CHECK(ctx, expr)                                                                         \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return lsa_set_error((ctx), LSA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                                    \
    } while (0)

// allocations: out of memory is its own status (callers may fall back to a leaner method; nothing else may)
#define LSA_HIP_ALLOC(ctx, expr)                                                                                    \
    do {                                                                                                            \
        hipError_t _e = (expr);                                                                                     \
        if (_e != hipSuccess) {                                                                                     \
            (void)hipGetLastError();                                                                                \
            return lsa_set_error((ctx), _e == hipErrorOutOfMemory ? LSA_ERR_OOM : LSA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                                        \
        }                                                                                                           \
    } while (0)

#define LSA_CHECK(expr)            \
    do {                           \
        int _rc = (expr);          \
        if (_rc != LSA_OK) return _rc; \
    } while (0)

// ---- kernels launched across translation units --------------------------------------------------------
// (all launch on ctx->stream and return LSA_OK or a status)

// y = A x                       (spmv.hip)
int k_spmv(lsa_ctx* ctx, const lsa_mat* A, int xdtype, const void* x, void* y);
int k_spmv_transpose(lsa_ctx* ctx, const lsa_mat* A, int conj, int xdtype, const void* x, void* y);

// BLAS-1 / tall-skinny kernels (blas.hip); T selected by dtype
int k_copy(lsa_ctx* ctx, int dtype, int64_t n, const void* x, void* y);
int k_set_zero(lsa_ctx* ctx, int dtype, int64_t n, void* x);
// y += alpha x
int k_axpy(lsa_ctx* ctx, int dtype, int64_t n, const double alpha[2], const void* x, void* y);
// h[c] = sum_i conj(V[i,c]) w[i], c < j   -> device array h (same dtype); V column-major with leading dim ldv
int k_multi_dot(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, const void* w, void* h_dev);
// w -= V h, and nrm2_dev[0] = ||w||^2 afterwards when nrm2_dev != null
int k_multi_axpy_dot(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, const void* h1_dev, void* w, void* h2_dev);
int k_multi_axpy(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, const void* h_dev, void* w,
                 double* nrm2_dev);
// columns of X: canonical phase (largest entry real positive), unit norm when `unit`; imag2_dev[c] = sum Im(X[:, c])^2
int k_columns_canonical(lsa_ctx* ctx, int64_t n, int ncols, void* X, int64_t ldx, int unit, double* imag2_dev);
// nrm2_dev[0] = ||x||^2
int k_nrm2(lsa_ctx* ctx, int dtype, int64_t n, const void* x, double* nrm2_dev);
int k_mask(lsa_ctx* ctx, int dtype, int64_t n, const double* keep_dev, void* y);
int k_residual_norms(lsa_ctx* ctx, int dtype, int64_t n, const void* b, const void* z, void* w, double* nrm2_dev);
// y = x / sqrt(nrm2_dev[0])   (no host round trip)
int k_scale_by_inv_norm(lsa_ctx* ctx, int dtype, int64_t n, const void* x, const double* nrm2_dev, void* y);
// hsum[c] += hadd[c] (c < j) and hsum[j] = sqrt(nrm2[0])  -- assembles one Hessenberg column on the device
int k_hess_column(lsa_ctx* ctx, int dtype, int j, const void* h1, const void* h2, const double* nrm2_dev, void* hout);
// CGS2 + normalisation of one Arnoldi step in five launches, the inner solve's residual check folded in (blas.hip); returns 1
// (nothing launched) for shapes it does not handle
size_t k_cgs2_fused_work_bytes(lsa_ctx* ctx, int64_t n, int jmax);
int k_cgs2_fused(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, void* w, void* vnext, void* hcol_dev, void* work,
                 const void* chk_b, const void* chk_z, double* chk_out);
// The same CGS2 for an Arnoldi step of OP = C^-1 M whose matrices share their index arrays, in five launches that leave y alone
// and end with ONE kernel that normalises (v_next = w / ||w||, Hessenberg column), multiplies t <- M v_next for the next step
// and checks this step's inner solve (|t - C y|^2, |t|^2 per workgroup into tail_part, summed by k_cgs2_tail_checks): 18
// launches per step instead of 20.  The two products are bit for bit those of k_spmv.  Complex vectors only.
bool k_cgs2_tail_fits(lsa_ctx* ctx, int64_t n, int jmax, const lsa_mat* M, const lsa_mat* C);
int k_cgs2_tail_parts(int64_t n);  // workgroups of the tail kernel = (|t - C y|^2, |t|^2) pairs per step
int k_cgs2_fused_tail(lsa_ctx* ctx, int64_t n, int j, const void* V, int64_t ldv, const void* y, void* w, void* vnext, void* hcol_dev, void* work,
                      const lsa_mat* M, const lsa_mat* C, void* t, double* tail_part);
int k_cgs2_tail_checks(lsa_ctx* ctx, int nslots, int nparts, const double* tail_parts, double* checks);
int k_spmv_plain_subwave_lanes(const lsa_mat* A);  // spmv.hip
// Out[:, 0:k] = V[:, 0:m] Q   (Q m x k column-major on the device, ldq)
int k_basis_gemm(lsa_ctx* ctx, int dtype, int64_t n, int m, int k, const void* V, int64_t ldv, const void* Q, int ldq,
                 void* Out, int64_t ldo);

// ---- ILU (ilu.hip) -------------------------------------------------------------------------------------
struct lsa_ilu {
    lsa_ctx* ctx;
    int32_t n;
    int64_t nnz;
    int dtype;  // scalar type of the factors
    int32_t *rp, *ci, *diag;   // device pattern of the factors + position of the diagonal
    void* val;                 // device factor values (L strictly lower, unit diagonal implied; U upper)
    void* dinv;                // device 1/u_ii
    int32_t *order_l, *order_u;  // device row lists sorted by dependency level
    std::vector<int32_t> lvl_ptr_l, lvl_ptr_u;  // host level pointers into the row lists
    std::vector<int32_t> h_rp, h_ci, h_diag;
    int32_t nshift;
    int32_t* flag;   // device int[4]: abort / error / shift count / spare
    int sptrsv_blocks_l, sptrsv_blocks_u;  // workgroups used by the sync-free solves
    int algo;                               // 0 = level-by-level launches, 1 = sync-free, 2 = blocked (inverted diagonal blocks)
    void* tmp;                              // device vector (intermediate of L then U), sized on demand
    int tmp_dtype;
    // blocked form (sptrsv_block.hip)
    int32_t blk_B = 0, blk_nb = 0;
    int32_t *lsplit = nullptr, *usplit = nullptr;  // device: first in-block L entry / first out-of-block U entry per row
    void *linv = nullptr, *uinv = nullptr;         // device: n x B row-major dense inverses of the diagonal blocks
    void *blk_t[2] = {nullptr, nullptr}, *blk_y[2] = {nullptr, nullptr};      // per vector dtype: work vectors
    void *blk_in[2] = {nullptr, nullptr}, *blk_out[2] = {nullptr, nullptr};  // fixed graph input / output
    void* blk_graph[2] = {nullptr, nullptr};                                 // hipGraphExec_t of the full apply
};

int ilu_solve_dev(lsa_ctx* ctx, lsa_ilu* pc, int which, int vdtype, const void* b, void* x);
int blk_setup(lsa_ctx* ctx, lsa_ilu* pc, int32_t B);
int blk_solve(lsa_ctx* ctx, lsa_ilu* pc, int which, int vdtype, const void* b, void* x);
void blk_release(lsa_ilu* pc);

// ---- nested-dissection multifrontal LU (ndlu.hip) -------------------------------------------------------
struct lsa_ndlu;
int ndlu_solve_dev(lsa_ctx* ctx, lsa_ndlu* f, int vdtype, const void* b, void* x);
int ndlu_solve_adjoint_dev(lsa_ctx* ctx, lsa_ndlu* f, int conj, int vdtype, const void* b, void* x);
