// v_permlane32_swap_b32 on gfx950: which halves does it exchange?   hipcc --offload-arch=gfx950 -O2 -o permlane_swap.bin permlane_swap.hip
// Expected (what OpsPair relies on): r[0] = (a lanes 0-31 | b lanes 0-31), r[1] = (a lanes 32-63 | b lanes 32-63).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o) {
    const unsigned a = 1000u + threadIdx.x, b = 2000u + threadIdx.x;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
    printf("r1: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
    const bool ok = h[0] == 1000 && h[31] == 1031 && h[32] == 2000 && h[63] == 2031 && h[64] == 1032 && h[95] == 1063 && h[96] == 2032 && h[127] == 2063;
    printf("%s\n", ok ? "as OpsPair expects" : "DIFFERENT");
    return ok ? 0 : 1;
}
