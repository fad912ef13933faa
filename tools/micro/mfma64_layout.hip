// v_mfma_f64_16x16x4_f64 on gfx950: which (row, column) of D sits in register r of lane L, in which order the four k
// steps are accumulated, and whether each step is a fused multiply-add (run once on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4v __attribute__((ext_vector_type(4)));
__global__ void k(double *out) {
    const int L = threadIdx.x;
    // A[i][k]: lane holds i = L % 16, k = L / 16;  B[k][j]: lane holds j = L % 16, k = L / 16
    const double a = (L / 16 == 0) ? (double)(L % 16 + 1) : 0.0;         // A[i][0] = i + 1
    const double b = (L / 16 == 0) ? (double)(100 * (L % 16 + 1)) : 0.0; // B[0][j] = 100 (j + 1)
    d4v d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[L * 4 + r] = d[r];
    // order: C = 1; products p0 = 2^-60, p1 = -1, p2 = 2^-60, p3 = 0 -> sequential k = 0..3: ((1 + 2^-60) - 1) + 2^-60 = 2^-60
    const int kk = L / 16;
    const double a2 = 1.0, b2 = kk == 0 ? 0x1p-60 : (kk == 1 ? -1.0 : (kk == 2 ? 0x1p-60 : 0.0));
    d4v c = {1, 1, 1, 1};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, c, 0, 0, 0);
    if (L == 0) out[256] = c[0];
    // fused: C = -(1 + 2^-29), a = b = 1 + 2^-30 (k = 0 only): exact a*b = 1 + 2^-29 + 2^-60 -> fused: 2^-60; unfused: 0
    const double a3 = kk == 0 ? 1.0 + 0x1p-30 : 0.0, b3 = a3;
    d4v e;
    for (int r = 0; r < 4; ++r) e[r] = -(1.0 + 0x1p-29);
    e = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, e, 0, 0, 0);
    if (L == 0) out[257] = e[0];
    // denormal
    const double a4 = kk == 0 ? 1.0e-310 : 0.0, b4 = kk == 0 ? 1.0 : 0.0;
    d4v g = {0, 0, 0, 0};
    g = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b4, g, 0, 0, 0);
    if (L == 0) out[258] = g[0];
}
int main() {
    double *d, h[260];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int L : {0, 1, 15, 16, 17, 32, 48, 63}) {
        printf("lane %2d:", L);
        for (int r = 0; r < 4; ++r) {
            const long v = (long)(h[L * 4 + r] + 0.5);
            const int j = L % 16;
            const long q = v / (100 * (j + 1));
            printf(" r%d=(i%ld,j%d)%s", r, q - 1, j, (q * 100 * (j + 1) == v) ? "" : "?");
        }
        printf("\n");
    }
    printf("k order: -> %g (8.67e-19 = 2^-60: k = 0,1,2,3 in sequence; other values: another order)\n", h[256]);
    printf("fused?   -> %g (8.67e-19: fused; 0: product rounded first)\n", h[257]);
    printf("denormal 1e-310 * 1 -> %g\n", h[258]);
    return 0;
}
