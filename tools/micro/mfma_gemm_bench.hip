// Development aid: FP64 matrix-core product C += A B (row-major), tile-shape variants of the kernel in csrc/ndlu.hip, and the
// bare instruction rate.   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_gemm_bench.hip -o /tmp/mfma_bench && /tmp/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void peak_kernel(double* out, int iters) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// WG tile (32 WI) x (32 WJ), 2 x 2 wavefronts, each WI x WJ instruction tiles; BK = 16; NBUF LDS buffers
template <int WI, int WJ, int NBUF, bool SWZ>
__global__ __launch_bounds__(256) void gemm_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int M, int N, int K) {
    constexpr int TM = 32 * WI, TN = 32 * WJ, BK = 16, LDAS = BK + 1, LDBS = TN + 16;
    __shared__ double As[NBUF][TM * LDAS];
    __shared__ double Bs[NBUF][BK * LDBS];
    const int tilesN = (N + TN - 1) / TN, tilesM = (M + TM - 1) / TM;
    int bid = blockIdx.x;
    int tm, tn;
    if (SWZ) {  // label bid % 8 -> band of tile rows, super-tiles of 8 x 16 inside
        const int nwg = tilesM * tilesN, x = bid % 8, i = bid / 8;
        const int r0 = (int)((long long)tilesM * x / 8), r1 = (int)((long long)tilesM * (x + 1) / 8);
        const int h = r1 - r0, sh = h < 8 ? h : 8, sw = 128 / sh;
        (void)nwg;
        // i-th tile of band x in super-tile order (bands assumed equal: tilesM % 8 == 0)
        const int per_rb = sh * tilesN;  // tiles per row-block of sh rows
        const int rb = i / per_rb, rem = i % per_rb;
        const int cbw = sh * sw, cb = rem / cbw, rem2 = rem % cbw;
        const int wcur = (cb * sw + sw <= tilesN) ? sw : tilesN - cb * sw;
        tm = r0 + rb * sh + rem2 / wcur;
        tn = cb * sw + rem2 % wcur;
    } else {
        tm = bid / tilesN, tn = bid % tilesN;
    }
    const int row0 = tm * TM, col0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = 16 * WI * (wave >> 1), wc = 16 * WJ * (wave & 1), l15 = lane & 15, l4 = lane >> 4;
    d4 acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = d4{0, 0, 0, 0};
    constexpr int NA = TM * BK / 256, NB = BK * TN / 256;
    double pa[NA], pb[NB];
    auto gload = [&](int kk) {
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            const int e = tid + 256 * s;
            const int gr = row0 + (e >> 4), gk = kk + (e & 15);
            pa[s] = (gr < M && gk < K) ? A[(size_t)gr * K + gk] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const int e = tid + 256 * s;
            const int gk = kk + e / TN, gc = col0 + e % TN;
            pb[s] = (gk < K && gc < N) ? B[(size_t)gk * N + gc] : 0.0;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            const int e = tid + 256 * s;
            As[buf][(e >> 4) * LDAS + (e & 15)] = pa[s];
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const int e = tid + 256 * s;
            Bs[buf][(e / TN) * LDBS + e % TN] = pb[s];
        }
    };
    gload(0);
    int buf = 0;
    if (NBUF == 2) { sstore(0); __syncthreads(); if (BK < K) gload(BK); }
    for (int kk = 0; kk < K; kk += BK) {
        if (NBUF == 1) {
            sstore(0);
            __syncthreads();
            if (kk + BK < K) gload(kk + BK);
        }
#pragma unroll
        for (int k4 = 0; k4 < BK; k4 += 4) {
            double a[WI], b[WJ];
#pragma unroll
            for (int i = 0; i < WI; ++i) a[i] = As[buf][(wr + 16 * i + l15) * LDAS + k4 + l4];
#pragma unroll
            for (int j = 0; j < WJ; ++j) b[j] = Bs[buf][(k4 + l4) * LDBS + wc + 16 * j + l15];
#pragma unroll
            for (int i = 0; i < WI; ++i)
#pragma unroll
                for (int j = 0; j < WJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (NBUF == 1) {
            __syncthreads();
        } else {
            if (kk + BK < K) {  // the other buffer was last read one iteration ago, behind the barrier below
                sstore(buf ^ 1);
                __syncthreads();
                if (kk + 2 * BK < K) gload(kk + 2 * BK);
                buf ^= 1;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = row0 + wr + 16 * i + l4 + 4 * r;
            if (gr >= M) continue;
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int gc = col0 + wc + 16 * j + l15;
                if (gc < N) C[(size_t)gr * N + gc] += acc[i][j][r];
            }
        }
}

template <int WI, int WJ, int NBUF, bool SWZ>
void run(const char* name, const double* A, const double* B, double* C, int M, int N, int K, const std::vector<double>& hA, const std::vector<double>& hB) {
    constexpr int TM = 32 * WI, TN = 32 * WJ;
    const int grid = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    CK(hipMemset(C, 0, (size_t)M * N * 8));
    hipLaunchKernelGGL((gemm_kernel<WI, WJ, NBUF, SWZ>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    // spot check
    std::vector<double> row(N);
    double err = 0;
    for (int r : {0, 77, M - 1}) {
        CK(hipMemcpy(row.data(), C + (size_t)r * N, N * 8, hipMemcpyDeviceToHost));
        for (int c : {0, 1, 65, N - 1}) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += hA[(size_t)r * K + k] * hB[(size_t)k * N + c];
            err = fmax(err, fabs(ref - row[c]) / (fabs(ref) + 1e-300));
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_kernel<WI, WJ, NBUF, SWZ>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s M=%d N=%d K=%d  %.2f ms  %.1f TFLOP/s  rel.err %.1e\n", name, M, N, K, ms / reps, 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12, err);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 4096;
    {
        double* out;
        CK(hipMalloc(&out, 256 * 4 * 256 * 8 * 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int wgs : {256 * 1, 256 * 2, 256 * 4}) {
            const int iters = 20000;
            hipLaunchKernelGGL(peak_kernel, dim3(wgs), dim3(256), 0, 0, out, 100);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(peak_kernel, dim3(wgs), dim3(256), 0, 0, out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("bare v_mfma_f64_16x16x4: %d workgroups of 4 wavefronts: %.1f TFLOP/s\n", wgs, (double)wgs * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12);
        }
    }
    std::vector<double> hA((size_t)M * K), hB((size_t)K * N);
    srand(1);
    for (auto& v : hA) v = rand() / (double)RAND_MAX - 0.5;
    for (auto& v : hB) v = rand() / (double)RAND_MAX - 0.5;
    double *A, *B, *C;
    CK(hipMalloc(&A, hA.size() * 8));
    CK(hipMalloc(&B, hB.size() * 8));
    CK(hipMalloc(&C, (size_t)M * N * 8));
    CK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
    run<2, 2, 1, false>("64x64 one buffer (csrc)", A, B, C, M, N, K, hA, hB);
    run<2, 2, 1, true>("64x64 one buffer, XCD bands", A, B, C, M, N, K, hA, hB);
    run<2, 2, 2, false>("64x64 two buffers", A, B, C, M, N, K, hA, hB);
    run<4, 2, 1, false>("128x64 one buffer", A, B, C, M, N, K, hA, hB);
    run<4, 2, 1, true>("128x64 one buffer, XCD bands", A, B, C, M, N, K, hA, hB);
    run<2, 4, 1, true>("64x128 one buffer, XCD bands", A, B, C, M, N, K, hA, hB);
    run<4, 2, 2, true>("128x64 two buffers, XCD bands", A, B, C, M, N, K, hA, hB);
    run<4, 4, 1, false>("128x128 one buffer", A, B, C, M, N, K, hA, hB);
    run<4, 4, 1, true>("128x128 one buffer, XCD bands", A, B, C, M, N, K, hA, hB);
    run<4, 4, 2, false>("128x128 two buffers", A, B, C, M, N, K, hA, hB);
    run<4, 4, 2, true>("128x128 two buffers, XCD bands", A, B, C, M, N, K, hA, hB);
    return 0;
}
