// What a grid-wide barrier inside one persistent kernel costs on this part, against a kernel boundary.
// Every round each workgroup writes a stamp (and optionally streams `touch` bytes through a buffer), all workgroups
// meet at an agent-scope barrier, then each checks the stamp of a workgroup half a grid away (another XCD).
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                          \
        }                                                                      \
    } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* abort_flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}

__global__ void __launch_bounds__(256) persistent(int rounds, unsigned* counter, unsigned* abort_flag, unsigned* stamps, double* buf, long touch_elems,
                                                  unsigned* errors) {
    const unsigned G = gridDim.x, w = blockIdx.x;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned* cur = stamps + (size_t)(r & 1) * G;
        if (threadIdx.x == 0) cur[w] = (unsigned)r * 7919u + w;
        for (long i = (long)w * 256 + threadIdx.x; i < touch_elems; i += (long)G * 256) buf[i] = buf[i] * 1.0000001 + 1.0;
        if (!grid_barrier(counter, G * (unsigned)(r + 1), abort_flag)) return;
        const unsigned other = (w + G / 2 + 1) % G;
        if (threadIdx.x == 0 && cur[other] != (unsigned)r * 7919u + other) ++bad;
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errors, bad);
}

__global__ void __launch_bounds__(256) one_round(int r, unsigned* stamps, double* buf, long touch_elems, unsigned* errors) {
    const unsigned G = gridDim.x, w = blockIdx.x;
    unsigned* cur = stamps + (size_t)(r & 1) * G;
    unsigned* prev = stamps + (size_t)((r + 1) & 1) * G;
    const unsigned other = (w + G / 2 + 1) % G;
    if (threadIdx.x == 0 && r > 0 && prev[other] != (unsigned)(r - 1) * 7919u + other) atomicAdd(errors, 1u);
    if (threadIdx.x == 0) cur[w] = (unsigned)r * 7919u + w;
    for (long i = (long)w * 256 + threadIdx.x; i < touch_elems; i += (long)G * 256) buf[i] = buf[i] * 1.0000001 + 1.0;
}

int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent, 256, 0));
    printf("%s: %d CUs, %d resident workgroups of 256 per CU\n", prop.name, prop.multiProcessorCount, per_cu);
    const int rounds = 2000;
    unsigned *counter, *abort_flag, *stamps, *errors;
    double* buf;
    const long max_touch = 1 << 22;  // 32 MB of doubles
    CK(hipMalloc(&counter, 4));
    CK(hipMalloc(&abort_flag, 4));
    CK(hipMalloc(&errors, 4));
    CK(hipMalloc(&stamps, 2 * 8192 * 4));
    CK(hipMalloc(&buf, max_touch * 8));
    CK(hipMemset(buf, 0, max_touch * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (long touch : {0L, 1L << 16, 1L << 19, 1L << 22}) {
        for (int G : {256, 512, 1024, 2048}) {
            if (G > prop.multiProcessorCount * per_cu) continue;
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipMemsetAsync(abort_flag, 0, 4, s));
            CK(hipMemsetAsync(errors, 0, 4, s));
            CK(hipEventRecord(e0, s));
            persistent<<<G, 256, 0, s>>>(rounds, counter, abort_flag, stamps, buf, touch, errors);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h_err = 0, h_abort = 0;
            CK(hipMemcpy(&h_err, errors, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&h_abort, abort_flag, 4, hipMemcpyDeviceToHost));
            printf("persistent  G=%5d touch=%8ld B: %7.2f us per round, stale reads %u, aborted %u\n", G, touch * 8, 1e3 * ms / rounds, h_err, h_abort);
            if (h_abort) return 2;
            CK(hipMemsetAsync(errors, 0, 4, s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < rounds; ++r) one_round<<<G, 256, 0, s>>>(r, stamps, buf, touch, errors);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&h_err, errors, 4, hipMemcpyDeviceToHost));
            printf("launches    G=%5d touch=%8ld B: %7.2f us per round, stale reads %u\n", G, touch * 8, 1e3 * ms / rounds, h_err);
        }
    }
    return 0;
}
