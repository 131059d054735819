// Development microbenchmark: cost of straight-line (executed-once) code vs the same work in a loop, one workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Unroll { template <typename F> __device__ static void run(F f) { f(); Unroll<N - 1>::run(f); } };
template <> struct Unroll<0> { template <typename F> __device__ static void run(F) {} };

__global__ __launch_bounds__(1024) void k_straight(double* o, double s) {
    const unsigned long long t0 = wall_clock64();
    double a = threadIdx.x + (t0 == 1234567ull ? 1.0 : 0.0), b = a + 1, c = a + 2, d = a + 3;
#pragma unroll
    for (int i = 0; i < 1024; ++i) { a = fma(a, s, 1.0 + i); b = fma(b, s, 2.0 + i); c = fma(c, s, 3.0 + i); d = fma(d, s, 4.0 + i); }
    const double sum = a + b + c + d;
    const unsigned long long t1 = wall_clock64() + (sum == 12345.678 ? 1ull : 0ull);
    o[threadIdx.x] = sum;
    atomicMax((unsigned long long*)(o + 2048), t1 - t0);
}
__global__ __launch_bounds__(1024) void k_loop(double* o, double s, int n) {
    const unsigned long long t0 = wall_clock64();
    double a = threadIdx.x + (t0 == 1234567ull ? 1.0 : 0.0), b = a + 1, c = a + 2, d = a + 3;
#pragma unroll 1
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { a = fma(a, s, 1.0 + i); b = fma(b, s, 2.0 + i); c = fma(c, s, 3.0 + i); d = fma(d, s, 4.0 + i); }
    }
    const double sum = a + b + c + d;
    const unsigned long long t1 = wall_clock64() + (sum == 12345.678 ? 1ull : 0ull);
    o[threadIdx.x] = sum;
    atomicMax((unsigned long long*)(o + 2048), t1 - t0);
}
__global__ void k_flush(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }

int main() {
    double* o; hipMalloc(&o, 8 * 4096); double* big; hipMalloc(&big, 8 << 20); hipMemset(big, 0, 8 << 20);
    for (int threads : {64, 256, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            unsigned long long h;
            hipLaunchKernelGGL(k_flush, dim3(4096), dim3(256), 0, 0, big, 1 << 20);
            hipMemset(o + 2048, 0, 8); hipLaunchKernelGGL(k_straight, dim3(1), dim3(threads), 0, 0, o, 0.999); hipDeviceSynchronize();
            hipMemcpy(&h, o + 2048, 8, hipMemcpyDeviceToHost); printf("threads %4d straight-line 4096 FMA: %6.2f us", threads, (double)h * 0.01);
            hipLaunchKernelGGL(k_flush, dim3(4096), dim3(256), 0, 0, big, 1 << 20);
            hipMemset(o + 2048, 0, 8); hipLaunchKernelGGL(k_loop, dim3(1), dim3(threads), 0, 0, o, 0.999, 64); hipDeviceSynchronize();
            hipMemcpy(&h, o + 2048, 8, hipMemcpyDeviceToHost); printf("   looped 64x64 FMA: %6.2f us\n", (double)h * 0.01);
        }
    }
    return 0;
}
