// Which (row, column) of D does register r of lane L hold for v_mfma_f32_32x32x2_f32 on gfx950? (run once on the GPU box)
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/mfma_layout.bin tools/micro/mfma_layout.hip && tools/micro/mfma_layout.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float *out) {
    const int L = threadIdx.x;
    // A[i][k]: lane holds i = L % 32, k = L / 32;  B[k][j]: lane holds j = L % 32, k = L / 32
    const float a = (L / 32 == 0) ? (float)(L % 32 + 1) : 0.0f;         // A[i][0] = i + 1, A[i][1] = 0
    const float b = (L / 32 == 0) ? (float)(100 * (L % 32 + 1)) : 0.0f; // B[0][j] = 100 (j + 1)
    f16v d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[L * 16 + r] = d[r];
    // order of accumulation inside one instruction: k = 0 first? D = (C + a0 b0) + a1 b1 vs (C + a1 b1) + a0 b0
    const float a2 = (L / 32 == 0) ? 1.0f : 1.0f, b2 = (L / 32 == 0) ? 1.0e-8f : -1.0f; // with C = 1: (1 + 1e-8) - 1 = 0 ; (1 - 1) + 1e-8 = 1e-8
    f16v c = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, c, 0, 0, 0);
    if (L == 0) out[64 * 16] = c[0];
    // fused or not: C + a*b with a*b needing more than 24 bits: a = 1 + 2^-12, b = 1 + 2^-12, C = -(1 + 2^-11): exact = 2^-24
    const float a3 = (L / 32 == 0) ? 1.0f + 1.0f / 4096 : 0.0f, b3 = (L / 32 == 0) ? 1.0f + 1.0f / 4096 : 0.0f;
    f16v e;
    for (int r = 0; r < 16; ++r) e[r] = -(1.0f + 1.0f / 2048);
    e = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, e, 0, 0, 0);
    if (L == 0) out[64 * 16 + 1] = e[0];
    // denormal input
    const float a4 = (L / 32 == 0) ? 1.0e-40f : 0.0f, b4 = (L / 32 == 0) ? 1.0f : 0.0f;
    f16v g = {0};
    g = __builtin_amdgcn_mfma_f32_32x32x2f32(a4, b4, g, 0, 0, 0);
    if (L == 0) out[64 * 16 + 2] = g[0];
}
int main() {
    float *d, h[64 * 16 + 4];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int L : {0, 1, 31, 32, 33, 63}) {
        printf("lane %2d:", L);
        for (int r = 0; r < 16; ++r) {
            const int v = (int)(h[L * 16 + r] + 0.5f); // = 100 (j+1) (i+1)
            // find i, j
            int fi = -1, fj = -1;
            for (int j = 0; j < 32 && fi < 0; ++j)
                for (int i = 0; i < 32; ++i)
                    if (100 * (j + 1) * (i + 1) == v && j == L % 32) { fi = i; fj = j; break; }
            printf(" r%d=(i%d,j%d)", r, fi, fj);
        }
        printf("\n");
    }
    printf("k order: C=1, a0 b0 = 1e-8, a1 b1 = -1 -> %g   (0: k = 0 first; 1e-8: k = 1 first)\n", h[64 * 16]);
    printf("fused?  -(1+2^-11) + (1+2^-12)^2 -> %g   (5.96e-08 = 2^-24: fused; 0: product rounded first)\n", h[64 * 16 + 1]);
    printf("denormal input 1e-40 * 1 -> %g\n", h[64 * 16 + 2]);
    return 0;
}
