// Context, device vectors and CSR matrices of liblsa_hip.so.
#include "lsa_internal.h"

int lsa_set_error(lsa_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int lsa_ensure_scratch(lsa_ctx* ctx, size_t dbytes, size_t hbytes) {
    if (dbytes > ctx->dscratch_bytes) {
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->dscratch) (void)hipFree(ctx->dscratch);
        ctx->dscratch = nullptr;
        ctx->dscratch_bytes = 0;
        size_t want = dbytes * 2;
        LSA_HIP_CHECK(ctx, hipMalloc(&ctx->dscratch, want));
        ctx->dscratch_bytes = want;
    }
    if (hbytes > ctx->pinned_bytes) {
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        size_t want = hbytes * 2;
        LSA_HIP_CHECK(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    return LSA_OK;
}

lsa_mat* mat_row_view(const lsa_mat* full, int32_t r0, int32_t r1) {
    lsa_mat* v = new lsa_mat();
    v->ctx = full->ctx;
    v->n = r1 - r0;
    v->ncols = full->ncols;
    v->row0 = r0;
    v->nnz = (int64_t)full->h_rp[(size_t)r1] - full->h_rp[(size_t)r0];
    v->dtype = full->dtype;
    v->rp = full->rp + r0;  // row pointers keep their global offsets: ci / val are the viewed matrix's arrays
    v->ci = full->ci;
    v->val = full->val;
    v->owns_index = false;
    v->owns_values = false;
    v->ci16_state = -1;  // no compressed-index form for views
    return v;
}

extern "C" {

int lsa_ctx_create(int device, lsa_ctx** out) {
    if (!out) return LSA_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LSA_ERR_HIP;  // no GPU: fail loudly
    if (device < 0 || device >= ndev) return LSA_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return LSA_ERR_HIP;
    lsa_ctx* ctx = new lsa_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete ctx;
        return LSA_ERR_HIP;
    }
    ctx->arch = prop.gcnArchName;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return LSA_ERR_HIP;
    }
    if (lsa_ensure_scratch(ctx, 1 << 20, 1 << 16) != LSA_OK) {
        delete ctx;
        return LSA_ERR_HIP;
    }
    *out = ctx;
    return LSA_OK;
}

void lsa_ctx_destroy(lsa_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    lsa_ndlu_drop_cache(ctx);
    lsa_krylov_drop_cache(ctx);
    comm_release(ctx);  // before the stream the communicator is bound to goes away
    if (ctx->dscratch) (void)hipFree(ctx->dscratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* lsa_last_error(const lsa_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
const char* lsa_ctx_arch(const lsa_ctx* ctx) { return ctx ? ctx->arch.c_str() : ""; }

int lsa_ctx_synchronize(lsa_ctx* ctx) {
    if (!ctx) return LSA_ERR_ARG;
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

// ---- vectors ---------------------------------------------------------------------------------------------
static size_t dtype_size(int dtype) { return dtype == LSA_C128 ? 16 : 8; }

int lsa_vec_create(lsa_ctx* ctx, int64_t n, int dtype, lsa_vec** out) {
    if (!ctx || !out || n < 0 || (dtype != LSA_F64 && dtype != LSA_C128))
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_vec_create: bad argument");
    lsa_vec* v = new lsa_vec{ctx, n, dtype, nullptr};
    hipError_t e = hipMalloc(&v->d, (size_t)(n > 0 ? n : 1) * dtype_size(dtype));
    if (e != hipSuccess) {
        delete v;
        return lsa_set_error(ctx, LSA_ERR_HIP, "hipMalloc(vector of %lld) failed: %s", (long long)n, hipGetErrorString(e));
    }
    e = hipMemsetAsync(v->d, 0, (size_t)(n > 0 ? n : 1) * dtype_size(dtype), ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(v->d);
        delete v;
        return lsa_set_error(ctx, LSA_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    *out = v;
    return LSA_OK;
}

void lsa_vec_destroy(lsa_vec* v) {
    if (!v) return;
    if (v->ctx && v->ctx->stream) (void)hipStreamSynchronize(v->ctx->stream);
    if (v->d) (void)hipFree(v->d);
    delete v;
}

int lsa_vec_upload(lsa_ctx* ctx, lsa_vec* v, const void* host) {
    if (!ctx || !v || !host) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_vec_upload: null argument");
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(v->d, host, (size_t)v->n * dtype_size(v->dtype), hipMemcpyHostToDevice, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

int lsa_vec_download(lsa_ctx* ctx, const lsa_vec* v, void* host) {
    if (!ctx || !v || !host) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_vec_download: null argument");
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(host, v->d, (size_t)v->n * dtype_size(v->dtype), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

// ---- CSR ---------------------------------------------------------------------------------------------------
static int validate_csr(lsa_ctx* ctx, int32_t n, int32_t ncols, int64_t nnz, const int32_t* rp, const int32_t* ci) {
    if (rp[0] != 0 || rp[n] != nnz) return lsa_set_error(ctx, LSA_ERR_ARG, "CSR: rowptr[0] must be 0 and rowptr[n] == nnz");
    for (int32_t i = 0; i < n; ++i) {
        if (rp[i + 1] < rp[i]) return lsa_set_error(ctx, LSA_ERR_ARG, "CSR: rowptr not monotone at row %d", i);
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
            if (ci[p] < 0 || ci[p] >= ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "CSR: column %d out of range in row %d", ci[p], i);
            if (p > rp[i] && ci[p] <= ci[p - 1]) return lsa_set_error(ctx, LSA_ERR_ARG, "CSR: columns of row %d are not strictly increasing", i);
        }
    }
    return LSA_OK;
}

static int csr_upload_impl(lsa_ctx* ctx, int32_t n_local, int32_t ncols, int32_t row0, int64_t nnz, const int32_t* rowptr,
                           const int32_t* col, const void* val, int dtype, lsa_mat** out) {
    if (!ctx || !out || !rowptr || (nnz > 0 && (!col || !val)) || n_local < 0 || nnz < 0 ||
        (dtype != LSA_F64 && dtype != LSA_C128))
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_csr_upload: bad argument");
    if (nnz >= (int64_t)INT32_MAX) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_csr_upload: nnz >= 2^31 needs 64-bit row pointers");
    LSA_CHECK(validate_csr(ctx, n_local, ncols, nnz, rowptr, col));
    lsa_mat* m = new lsa_mat();
    m->ctx = ctx;
    m->n = n_local;
    m->ncols = ncols;
    m->row0 = row0;
    m->nnz = nnz;
    m->dtype = dtype;
    m->owns_index = true;
    m->rp = nullptr;
    m->ci = nullptr;
    m->val = nullptr;
    m->h_rp.assign(rowptr, rowptr + n_local + 1);
    m->h_ci.assign(col, col + nnz);
    size_t vb = (size_t)(nnz > 0 ? nnz : 1) * dtype_size(dtype);
    hipError_t e1 = hipMalloc(&m->rp, sizeof(int32_t) * (size_t)(n_local + 1));
    hipError_t e2 = hipMalloc(&m->ci, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    hipError_t e3 = hipMalloc(&m->val, vb);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        lsa_mat_destroy(m);
        return lsa_set_error(ctx, LSA_ERR_HIP, "hipMalloc(CSR n=%d nnz=%lld) failed", n_local, (long long)nnz);
    }
    hipError_t e = hipMemcpyAsync(m->rp, rowptr, sizeof(int32_t) * (size_t)(n_local + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(m->ci, col, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(m->val, val, (size_t)nnz * dtype_size(dtype), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {  // the half-built matrix and its device buffers go with the error
        lsa_mat_destroy(m);
        return lsa_set_error(ctx, LSA_ERR_HIP, "CSR upload failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return LSA_OK;
}

int lsa_csr_upload(lsa_ctx* ctx, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const void* val,
                   int dtype, lsa_mat** out) {
    return csr_upload_impl(ctx, n, n, 0, nnz, rowptr, col, val, dtype, out);
}

int lsa_csr_upload_shard(lsa_ctx* ctx, int32_t n_global, int32_t row0, int32_t row1, int64_t nnz_local,
                         const int32_t* rowptr_local, const int32_t* col, const void* val, int dtype, lsa_mat** out) {
    if (row0 < 0 || row1 < row0 || row1 > n_global) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_csr_upload_shard: bad row range");
    return csr_upload_impl(ctx, row1 - row0, n_global, row0, nnz_local, rowptr_local, col, val, dtype, out);
}

int64_t lsa_mat_rows(const lsa_mat* m) { return m ? (int64_t)m->n : 0; }

void lsa_mat_destroy(lsa_mat* m) {
    if (!m) return;
    if (m->ctx && m->ctx->stream) (void)hipStreamSynchronize(m->ctx->stream);
    if (m->owns_index) {
        if (m->rp) (void)hipFree(m->rp);
        if (m->ci) (void)hipFree(m->ci);
    }
    if (m->val && m->owns_values) (void)hipFree(m->val);
    if (!m->owns_extras) m->ci16 = nullptr, m->cbase = nullptr, m->grp_start = nullptr;
    if (m->grp_start) (void)hipFree(m->grp_start);
    if (m->ci16) (void)hipFree(m->ci16);
    if (m->cbase) (void)hipFree(m->cbase);
    delete m;
}

int lsa_mat_download_values(lsa_ctx* ctx, const lsa_mat* m, void* host_val) {
    if (!ctx || !m || !host_val) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_mat_download_values: null argument");
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(host_val, m->val, (size_t)m->nnz * dtype_size(m->dtype), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

}  // extern "C"

// ---- C = alpha A + beta B on a shared pattern ---------------------------------------------------------------
template <typename TA, typename TB, typename TC>
__global__ void axpby_kernel(int64_t nnz, const TA* __restrict__ a, const TB* __restrict__ b, cplx alpha, cplx beta,
                             TC* __restrict__ c) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) {
        cplx acc{0.0, 0.0};
        fma_acc(acc, alpha, to_cplx(a[p]));
        fma_acc(acc, beta, to_cplx(b[p]));
        s_from(c[p], acc.re, acc.im);
    }
}

template <typename TA, typename TB, typename TC>
static void launch_axpby(lsa_ctx* ctx, int64_t nnz, const void* a, const void* b, cplx alpha, cplx beta, void* c) {
    int threads = 256;
    int64_t want = (nnz + threads - 1) / threads;
    int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    hipLaunchKernelGGL((axpby_kernel<TA, TB, TC>), dim3(blocks), dim3(threads), 0, ctx->stream, nnz, (const TA*)a, (const TB*)b,
                       alpha, beta, (TC*)c);
}

extern "C" int lsa_csr_axpby(lsa_ctx* ctx, const lsa_mat* A, const lsa_mat* B, const double alpha[2], const double beta[2],
                             int out_dtype, lsa_mat** out) {
    if (!ctx || !A || !B || !alpha || !beta || !out) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_csr_axpby: null argument");
    if (A->n != B->n || A->nnz != B->nnz || A->ncols != B->ncols || A->row0 != B->row0)
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_csr_axpby: operands must share one sparsity pattern (n %d vs %d, nnz %lld vs %lld)",
                             A->n, B->n, (long long)A->nnz, (long long)B->nnz);
    if (!A->h_rp.same_object(B->h_rp) || !A->h_ci.same_object(B->h_ci)) {
        // compared entry by entry once; from then on B shares A's host copy (and its hash), and the next call -- one per
        // eigen-solve -- sees one object
        if (A->h_rp != B->h_rp || A->h_ci != B->h_ci)
            return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_csr_axpby: operands must share one sparsity pattern (index arrays differ)");
        B->h_rp = A->h_rp;
        B->h_ci = A->h_ci;
        if (!A->h_hash) A->h_hash = B->h_hash ? B->h_hash : std::make_shared<uint64_t>(0);
        B->h_hash = A->h_hash;
        if (!A->h_shared) A->h_shared = B->h_shared ? B->h_shared : std::make_shared<PatternShared>();
        B->h_shared = A->h_shared;
    }
    if (out_dtype == LSA_F64 && (A->dtype != LSA_F64 || B->dtype != LSA_F64 || alpha[1] != 0.0 || beta[1] != 0.0))
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_csr_axpby: a real result needs real operands and real coefficients");
    lsa_mat* C = new lsa_mat();
    C->ctx = ctx;
    C->n = A->n;
    C->ncols = A->ncols;
    C->row0 = A->row0;
    C->nnz = A->nnz;
    C->dtype = out_dtype;
    C->rp = A->rp;
    C->ci = A->ci;
    C->owns_index = false;
    C->h_rp = A->h_rp;
    C->h_ci = A->h_ci;
    if (!A->h_hash) A->h_hash = std::make_shared<uint64_t>(0);
    C->h_hash = A->h_hash;
    if (!A->h_shared) A->h_shared = std::make_shared<PatternShared>();
    C->h_shared = A->h_shared;
    C->val = nullptr;
    hipError_t e = hipMalloc(&C->val, (size_t)(C->nnz > 0 ? C->nnz : 1) * dtype_size(out_dtype));
    if (e != hipSuccess) {
        delete C;
        return lsa_set_error(ctx, LSA_ERR_HIP, "hipMalloc(axpby result) failed: %s", hipGetErrorString(e));
    }
    cplx al{alpha[0], alpha[1]}, be{beta[0], beta[1]};
    const bool ca = A->dtype == LSA_C128, cb = B->dtype == LSA_C128;
    if (out_dtype == LSA_F64) launch_axpby<double, double, double>(ctx, C->nnz, A->val, B->val, al, be, C->val);
    else if (!ca && !cb) launch_axpby<double, double, cplx>(ctx, C->nnz, A->val, B->val, al, be, C->val);
    else if (ca && !cb) launch_axpby<cplx, double, cplx>(ctx, C->nnz, A->val, B->val, al, be, C->val);
    else if (!ca && cb) launch_axpby<double, cplx, cplx>(ctx, C->nnz, A->val, B->val, al, be, C->val);
    else launch_axpby<cplx, cplx, cplx>(ctx, C->nnz, A->val, B->val, al, be, C->val);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        lsa_mat_destroy(C);
        return lsa_set_error(ctx, LSA_ERR_HIP, "axpby launch failed: %s", hipGetErrorString(le));
    }
    *out = C;
    return LSA_OK;
}
