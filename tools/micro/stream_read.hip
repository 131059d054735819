// Development microbenchmark: what does a pure streaming read / copy reach on this MI355X (the practical HBM ceiling the
// SpMV is compared with)?   hipcc --offload-arch=gfx950 -O3 stream_read.hip -o stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n, double* out) {
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        s += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    }
    for (; i < n; i += stride) s += a[i].x + a[i].y;
    if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
int main() {
    const size_t bytes = 3200ull << 20, n = bytes / 16;
    double2 *a, *b; double* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 8);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        for (int which = 0; which < 2; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0, 0);
                if (which == 0) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, o);
                else hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%s blocks %5d: %.3f ms -> %.2f TB/s\n", which == 0 ? "read 3.36 GB" : "copy 3.36+3.36 GB", blocks, best,
                   (which == 0 ? 1.0 : 2.0) * bytes / best / 1e9);
        }
    }
    return 0;
}
