// RCCL plumbing for the row-sharded path (one process per GPU).  RCCL is opened with dlopen so that a
// single-GPU user never needs it and so that the copy bundled with PyTorch (used only by bench.py's barrier)
// cannot clash with the one the data path uses.
#include <dlfcn.h>

#include <algorithm>

#include "lsa_internal.h"

namespace {

// the slice of the NCCL API the data path needs
typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
typedef int (*fn_comm_destroy)(nccl_comm);
typedef const char* (*fn_err_string)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_err_string err_string = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle || !r.why.empty()) return r;
    const char* env = getenv("LSA_RCCL_PATH");
    if (env && *env) r.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    // RCCL must sit on the HIP / HSA runtime this library is bound to.  A process can hold two ROCm trees (PyTorch wheels
    // bundle their own next to /opt/rocm): the RCCL of the other tree finds an HSA runtime nobody initialised ("no
    // ROCm-capable device is detected" in ncclCommInitRank).  So: the librccl next to the libamdhip64 that resolved hipStreamSynchronize
    // for us -- PyTorch's when the launcher imported torch first (bench.py: one runtime per process, a second communicator
    // inside it), ROCm's otherwise --
    if (!r.handle) {
        Dl_info info;
        if (dladdr((void*)&hipStreamSynchronize, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char* leaf : {"librccl.so.1", "librccl.so"}) {
                    if (r.handle) break;
                    r.handle = dlopen((dir + leaf).c_str(), RTLD_NOW | RTLD_LOCAL);
                }
            }
        }
    }
    // ... then whatever RCCL the process has loaded already, then ROCm's own copy
    const char* loaded[] = {"librccl.so", "librccl.so.1"};
    for (const char* c : loaded)
        if (!r.handle) r.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    const char* candidates[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char* c : candidates) {
        if (r.handle) break;
        r.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) {
        r.why = "librccl.so not found (set LSA_RCCL_PATH)";
        return r;
    }
    r.get_uid = (fn_get_uid)dlsym(r.handle, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.err_string = (fn_err_string)dlsym(r.handle, "ncclGetErrorString");
    if (!r.get_uid || !r.init_rank || !r.all_gather || !r.comm_destroy) {
        r.why = "librccl.so lacks the expected symbols";
        dlclose(r.handle);
        r.handle = nullptr;
    }
    return r;
}

constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar

}  // namespace

// in-place all-gather of equal-sized blocks of a replicated vector: block r (bytes_per_rank bytes at offset
// r * bytes_per_rank) is this rank's contribution when r == ctx->rank
int k_allgather_inplace(lsa_ctx* ctx, void* vec, size_t bytes_per_rank) {
    if (ctx->nranks <= 1) return LSA_OK;
    ++ctx->comm_calls;
    ctx->comm_bytes += (int64_t)bytes_per_rank * (ctx->nranks - 1);  // received per rank
    {
        // LSA_TRACE_COMM: one line per exchange on standard error.  Ranks that take different turns show up as sequences that
        // differ (tools/micro/dist_c1m_comm.sh compares them); this is how round 4's out-of-step ranks were traced.
        static const bool trace = getenv("LSA_TRACE_COMM") != nullptr;
        if (trace) fprintf(stderr, "[comm rank %d] call %lld bytes %zu\n", ctx->rank, (long long)ctx->comm_calls, bytes_per_rank);
    }
    if (ctx->host_gather) {
        // host-staged transport (lsa_comm_init_host): own block to pinned memory, the caller's exchange (e.g.
        // torch.distributed over gloo), everything back.  Same layout and call sites as the RCCL path.
        const size_t total = bytes_per_rank * (size_t)ctx->nranks;
        if (total > ctx->comm_stage_bytes) {
            LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->comm_stage) (void)hipHostFree(ctx->comm_stage);
            ctx->comm_stage = nullptr;
            ctx->comm_stage_bytes = 0;
            LSA_HIP_CHECK(ctx, hipHostMalloc(&ctx->comm_stage, total * 2, hipHostMallocDefault));
            ctx->comm_stage_bytes = total * 2;
        }
        char* stage = (char*)ctx->comm_stage;
        const size_t off = (size_t)ctx->rank * bytes_per_rank;
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(stage + off, (const char*)vec + off, bytes_per_rank, hipMemcpyDeviceToHost, ctx->stream));
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const int hrc = ctx->host_gather(stage, (int64_t)bytes_per_rank, ctx->host_gather_user);
        if (hrc != 0) return lsa_set_error(ctx, LSA_ERR_COMM, "host all-gather callback failed (%d)", hrc);
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(vec, stage, total, hipMemcpyHostToDevice, ctx->stream));
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next exchange
        return LSA_OK;
    }
    Rccl& r = rccl();
    if (!r.handle || !ctx->comm) return lsa_set_error(ctx, LSA_ERR_COMM, "all-gather requested without an initialised communicator");
    const char* send = (const char*)vec + (size_t)ctx->rank * bytes_per_rank;
    int rc = r.all_gather(send, vec, bytes_per_rank, kNcclChar, (nccl_comm)ctx->comm, ctx->stream);
    if (rc != 0) return lsa_set_error(ctx, LSA_ERR_COMM, "ncclAllGather failed: %s", r.err_string ? r.err_string(rc) : "?");
    return LSA_OK;
}

// The ranks of a sharded layout agree on a status before any of them enters a data collective or returns from a collective
// set-up: every rank passes what it has (LSA_OK or its own failure) and gets back its own failure, or else the status of the
// first rank that failed.  A rank that failed locally -- out of device memory for buffers whose size differs from rank to
// rank, a limit of the kernels, non-finite input -- thus never leaves the others waiting in an exchange that has no time-out.
// One 16-byte all-gather through the context's scratch (allocated with the context: nothing can fail before the exchange).
int k_agree_status(lsa_ctx* ctx, int rc) {
    if (ctx->nranks <= 1) return rc;
    const size_t slot = 4 * sizeof(int32_t);
    if (ctx->dscratch_bytes < slot * (size_t)ctx->nranks || ctx->pinned_bytes < slot * (size_t)ctx->nranks)
        return rc != LSA_OK ? rc : lsa_set_error(ctx, LSA_ERR_ARG, "k_agree_status: context scratch too small for %d ranks", ctx->nranks);
    int32_t* h = (int32_t*)ctx->pinned;
    char* d = (char*)ctx->dscratch;
    h[0] = rc;
    h[1] = h[2] = h[3] = 0;
    bool ok = hipMemcpyAsync(d + slot * (size_t)ctx->rank, h, slot, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
              hipStreamSynchronize(ctx->stream) == hipSuccess;
    const std::string mine = ctx->err;
    const int xrc = k_allgather_inplace(ctx, d, slot);  // (entered even after a local copy failure: the others are in it)
    ok = ok && xrc == LSA_OK && hipMemcpyAsync(h, d, slot * (size_t)ctx->nranks, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
         hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (rc != LSA_OK) {
        ctx->err = mine;
        return rc;
    }
    if (!ok) return lsa_set_error(ctx, xrc != LSA_OK ? xrc : LSA_ERR_HIP, "k_agree_status: the status exchange failed");
    for (int r = 0; r < ctx->nranks; ++r)
        if (h[4 * r] != LSA_OK) return lsa_set_error(ctx, h[4 * r], "rank %d failed with status %d before a collective step; every rank gives up with it", r, h[4 * r]);
    return LSA_OK;
}

// the smallest of the ranks' values (one 16-byte all-gather through the context's scratch, like the status agreement): figures
// every rank must compute alike -- the device memory the chunks of a distributed factorisation are planned against
int k_agree_min_i64(lsa_ctx* ctx, int64_t* value) {
    if (ctx->nranks <= 1) return LSA_OK;
    const size_t slot = 2 * sizeof(int64_t);
    if (ctx->dscratch_bytes < slot * (size_t)ctx->nranks || ctx->pinned_bytes < slot * (size_t)ctx->nranks)
        return lsa_set_error(ctx, LSA_ERR_ARG, "k_agree_min_i64: context scratch too small for %d ranks", ctx->nranks);
    int64_t* h = (int64_t*)ctx->pinned;
    char* d = (char*)ctx->dscratch;
    h[0] = *value;
    h[1] = 0;
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(d + slot * (size_t)ctx->rank, h, slot, hipMemcpyHostToDevice, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    LSA_CHECK(k_allgather_inplace(ctx, d, slot));
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(h, d, slot * (size_t)ctx->nranks, hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < ctx->nranks; ++r) *value = std::min(*value, h[2 * r]);
    return LSA_OK;
}

// Are the ranks in step?  Every rank passes the number of exchanges it has made so far; a rank that took a different turn
// somewhere (a decision taken from replicated data that was not bit-identical after all) has made more or fewer.  Exchanges of
// equal size pair up silently whatever they were meant to carry, so an out-of-step run can end with wrong numbers on every rank
// instead of a hang: this check turns that into an error.  Collective; one 16-byte all-gather.
int k_agree_in_step(lsa_ctx* ctx, const char* where) {
    if (ctx->nranks <= 1) return LSA_OK;
    const int64_t mine = ctx->comm_calls;
    int64_t lo = mine, hi = -mine;
    LSA_CHECK(k_agree_min_i64(ctx, &lo));
    LSA_CHECK(k_agree_min_i64(ctx, &hi));  // min of the negatives = -max; (both calls count alike on every rank)
    if (lo != -hi)
        return lsa_set_error(ctx, LSA_ERR_COMM, "%s: the ranks are out of step (between %lld and %lld exchanges so far; this rank %lld): a decision was "
                             "taken from replicated data that differs between ranks; the results of this solve cannot be trusted",
                             where, (long long)lo, (long long)-hi, (long long)mine);
    return LSA_OK;
}

void comm_release(lsa_ctx* ctx) {
    if (ctx->comm) {
        Rccl& r = rccl();
        if (r.handle) r.comm_destroy((nccl_comm)ctx->comm);
        ctx->comm = nullptr;
    }
    if (ctx->comm_stage) (void)hipHostFree(ctx->comm_stage);
    ctx->comm_stage = nullptr;
    ctx->comm_stage_bytes = 0;
    ctx->host_gather = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
}

extern "C" {

int lsa_comm_unique_id(void* id128) {
    if (!id128) return LSA_ERR_ARG;
    Rccl& r = rccl();
    if (!r.handle) return LSA_ERR_COMM;
    nccl_uid uid;
    if (r.get_uid(&uid) != 0) return LSA_ERR_COMM;
    memcpy(id128, &uid, sizeof uid);
    return LSA_OK;
}

int lsa_comm_init(lsa_ctx* ctx, int nranks, int rank, const void* id128) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_comm_init: bad argument");
    if (nranks == 1) {
        ctx->nranks = 1;
        ctx->rank = 0;
        return LSA_OK;
    }
    comm_release(ctx);  // a second initialisation replaces the first
    Rccl& r = rccl();
    if (!r.handle) return lsa_set_error(ctx, LSA_ERR_COMM, "RCCL unavailable: %s", r.why.c_str());
    LSA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    (void)hipGetLastError();  // RCCL reads the runtime's sticky last-error state: a handled failure of earlier work must not fail its setup
    nccl_uid uid;
    memcpy(&uid, id128, sizeof uid);
    nccl_comm comm = nullptr;
    int rc = r.init_rank(&comm, nranks, uid, rank);
    if (rc != 0) return lsa_set_error(ctx, LSA_ERR_COMM, "ncclCommInitRank failed: %s", r.err_string ? r.err_string(rc) : "?");
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return LSA_OK;
}

int lsa_comm_init_host(lsa_ctx* ctx, int nranks, int rank, lsa_host_allgather_fn fn, void* user) {
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_comm_init_host: bad argument");
    comm_release(ctx);
    ctx->nranks = nranks;
    ctx->rank = rank;
    if (nranks > 1) {
        ctx->host_gather = fn;
        ctx->host_gather_user = user;
    }
    return LSA_OK;
}

int lsa_comm_selftest(lsa_ctx* ctx, int64_t bytes) {
    if (!ctx || bytes < 1 || bytes > (1 << 28)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_comm_selftest: bad argument");
    Rccl& r = rccl();
    if (!r.handle) return lsa_set_error(ctx, LSA_ERR_COMM, "RCCL unavailable: %s", r.why.c_str());
    LSA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipGetLastError();  // (see lsa_comm_init)
    nccl_uid uid;
    int rc = r.get_uid(&uid);
    if (rc != 0) return lsa_set_error(ctx, LSA_ERR_COMM, "ncclGetUniqueId failed: %s", r.err_string ? r.err_string(rc) : "?");
    nccl_comm comm = nullptr;
    rc = r.init_rank(&comm, 1, uid, 0);
    if (rc != 0) return lsa_set_error(ctx, LSA_ERR_COMM, "ncclCommInitRank (one rank) failed: %s", r.err_string ? r.err_string(rc) : "?");
    std::vector<unsigned char> h((size_t)bytes), back((size_t)bytes, 0);
    for (int64_t i = 0; i < bytes; ++i) h[(size_t)i] = (unsigned char)(i * 131 + 7);
    void* d = nullptr;
    int out = LSA_OK;
    if (hipMalloc(&d, (size_t)bytes) != hipSuccess) {
        r.comm_destroy(comm);
        return lsa_set_error(ctx, LSA_ERR_OOM, "lsa_comm_selftest: out of device memory");
    }
    if (hipMemcpyAsync(d, h.data(), (size_t)bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) out = LSA_ERR_HIP;
    if (out == LSA_OK) {
        rc = r.all_gather(d, d, (size_t)bytes, kNcclChar, comm, ctx->stream);  // in place: block 0 of one
        if (rc != 0) out = lsa_set_error(ctx, LSA_ERR_COMM, "ncclAllGather failed: %s", r.err_string ? r.err_string(rc) : "?");
    }
    if (out == LSA_OK && (hipMemcpyAsync(back.data(), d, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                          hipStreamSynchronize(ctx->stream) != hipSuccess))
        out = lsa_set_error(ctx, LSA_ERR_HIP, "lsa_comm_selftest: copy back failed");
    if (out == LSA_OK && back != h) out = lsa_set_error(ctx, LSA_ERR_COMM, "lsa_comm_selftest: the gathered block differs from what was sent");
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    r.comm_destroy(comm);
    return out;
}

int lsa_comm_stats(const lsa_ctx* ctx, int64_t* calls, int64_t* bytes_received) {
    if (!ctx) return LSA_ERR_ARG;
    if (calls) *calls = ctx->comm_calls;
    if (bytes_received) *bytes_received = ctx->comm_bytes;
    return LSA_OK;
}

}  // extern "C"
