// Development check: operand and result lane maps of v_mfma_f64_16x16x4_f64 on gfx950, with exact integer data.
//   documented (cdna_hip_programming.md): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
//   D: 4 results per lane, col = l & 15, row = (l >> 4) + 4 * reg
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/, double* D /*16x16 row-major*/) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k2 = 0; k2 < 4; ++k2) hA[i * 4 + k2] = 1 + i + 17 * k2;
    for (int k2 = 0; k2 < 4; ++k2) for (int j = 0; j < 16; ++j) hB[k2 * 16 + j] = 2 + 3 * j - 5 * k2;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k2 = 0; k2 < 4; ++k2) s += hA[i * 4 + k2] * hB[k2 * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("f64 16x16x4 MFMA with the documented lane maps: %d of 256 results differ\n", bad);
    return bad != 0;
}
