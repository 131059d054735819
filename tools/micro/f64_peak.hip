// Register-only FP64 rates on this part: v_fma_f64 against v_mfma_f64_16x16x4_f64 (both advertised at 78.6 TFLOP/s).
// Build: hipcc --offload-arch=gfx950 -O3 f64_peak.hip -o f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fma_loop(int iters, double* out, double x, double y) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], x, y);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(int iters, double* out, double x, double y) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    const double a = x + threadIdx.x * 1e-9, b = y;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* out;
    hipMalloc(&out, 8);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    for (int wg_per_cu : {1, 2, 4}) {
        const int grid = cus * wg_per_cu;
        double ms = time_ms([&] { fma_loop<<<grid, 256>>>(iters, out, 1.0000001, 1e-9); });
        printf("v_fma_f64      %d WG/CU: %7.2f TFLOP/s\n", wg_per_cu, 2.0 * 16 * 256.0 * grid * iters / ms / 1e9);
        ms = time_ms([&] { mfma_loop<4><<<grid, 256>>>(iters, out, 1.0000001, 1e-9); });
        printf("mfma f64 x4acc %d WG/CU: %7.2f TFLOP/s\n", wg_per_cu, 2.0 * 1024 * 4 * 4.0 * grid * iters / ms / 1e9);
        ms = time_ms([&] { mfma_loop<16><<<grid, 256>>>(iters, out, 1.0000001, 1e-9); });
        printf("mfma f64 x16   %d WG/CU: %7.2f TFLOP/s\n", wg_per_cu, 2.0 * 1024 * 16 * 4.0 * grid * iters / ms / 1e9);
    }
    return 0;
}
