// Development microbenchmark: what does a kernel boundary cost on gfx950 in the shapes the block-LU factorisation uses?
//   hipcc --offload-arch=gfx950 -O3 launch_cost.hip -o launch_cost && ./launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_empty(int* p) { if (p == nullptr) p[0] = 1; }
// one workgroup touches `rows` cache lines, stride ld doubles2
__global__ __launch_bounds__(1024) void k_panel(double2* a, int ld, int m, int k0) {
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        double2* r = a + (size_t)i * ld + k0;
        double2 s = {0, 0};
        for (int j = 0; j < 8; ++j) { s.x += r[j].x; s.y += r[j].y; }
        r[0] = s;
    }
}
// grid-wide read-modify-write of the m x m tile
__global__ __launch_bounds__(256) void k_update(double2* a, int ld, int m) {
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < m; c += 256) { double2 v = a[(size_t)i * ld + c]; v.x += 1.0; a[(size_t)i * ld + c] = v; }
}
// column-tile shaped update: workgroup = 8 columns x all rows, 8 lanes per row
__global__ __launch_bounds__(1024) void k_update_tile(double2* a, int ld, int m) {
    const int c0 = blockIdx.x * 8, c = threadIdx.x & 7;
    for (int i = threadIdx.x >> 3; i < m; i += 128) { double2 v = a[(size_t)i * ld + c0 + c]; v.x += 1.0; a[(size_t)i * ld + c0 + c] = v; }
}

template <typename F> double time_us(hipStream_t st, int reps, F body) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    body(); hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) body();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}

int main() {
    const int m = 1024, ld = 1040;
    double2* a; CK(hipMalloc(&a, sizeof(double2) * (size_t)ld * m * 32));
    CK(hipMemset(a, 0, sizeof(double2) * (size_t)ld * m * 32));
    hipStream_t st; CK(hipStreamCreate(&st));
    int* d; CK(hipMalloc(&d, 64));
    printf("empty <<<1,64>>>            %7.2f us/launch\n", time_us(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d); }));
    printf("empty <<<1,1024>>>          %7.2f us/launch\n", time_us(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(1024), 0, st, d); }));
    printf("empty <<<1024,256>>>        %7.2f us/launch\n", time_us(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, d); }));
    printf("panel-touch <<<1,1024>>>    %7.2f us/launch\n", time_us(st, 2000, [&] { hipLaunchKernelGGL(k_panel, dim3(1), dim3(1024), 0, st, a, ld, m, 8); }));
    printf("update 16 MB rmw            %7.2f us/launch\n", time_us(st, 1000, [&] { hipLaunchKernelGGL(k_update, dim3(m), dim3(256), 0, st, a, ld, m); }));
    printf("update-tile 16 MB rmw       %7.2f us/launch\n", time_us(st, 1000, [&] { hipLaunchKernelGGL(k_update_tile, dim3(m / 8), dim3(1024), 0, st, a, ld, m); }));
    printf("update + panel-touch        %7.2f us/pair\n", time_us(st, 1000, [&] { hipLaunchKernelGGL(k_update, dim3(m), dim3(256), 0, st, a, ld, m);
                                                                                   hipLaunchKernelGGL(k_panel, dim3(1), dim3(1024), 0, st, a, ld, m, 8); }));
    printf("update + empty<<<1,64>>>    %7.2f us/pair\n", time_us(st, 1000, [&] { hipLaunchKernelGGL(k_update, dim3(m), dim3(256), 0, st, a, ld, m);
                                                                                   hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d); }));
    printf("update-tile + panel-touch   %7.2f us/pair\n", time_us(st, 1000, [&] { hipLaunchKernelGGL(k_update_tile, dim3(m / 8), dim3(1024), 0, st, a, ld, m);
                                                                                   hipLaunchKernelGGL(k_panel, dim3(1), dim3(1024), 0, st, a, ld, m, 8); }));
    // the same pair replayed from a graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < 100; ++r) { hipLaunchKernelGGL(k_update, dim3(m), dim3(256), 0, st, a, ld, m); hipLaunchKernelGGL(k_panel, dim3(1), dim3(1024), 0, st, a, ld, m, 8); }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("graph(update + panel-touch) %7.2f us/pair\n", time_us(st, 10, [&] { hipGraphLaunch(ge, st); }) / 100);
    // a different tile each time (cold L2) as in the factorisation of consecutive blocks
    int blk = 0;
    printf("update (32 tiles round robin) %7.2f us/launch\n", time_us(st, 1000, [&] { hipLaunchKernelGGL(k_update, dim3(m), dim3(256), 0, st, a + (size_t)(blk++ % 32) * ld * m, ld, m); }));
    return 0;
}
