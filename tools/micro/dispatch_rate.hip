// Workgroup dispatch rate: empty workgroups (one store per workgroup so that nothing is optimised away).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_rate tools/micro/dispatch_rate.hip && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(unsigned *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0xffffffffu) out[0] = 1;
}
__global__ void lds_kernel(unsigned *out) { // 16 KB of LDS per workgroup, a barrier
    extern __shared__ unsigned sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (sm[(threadIdx.x + 1) % blockDim.x] == 0xffffffffu) out[0] = 1;
}
int main() {
    unsigned *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int shapes[][2] = {{32768, 512}, {32768, 128}, {32768, 64}, {131072, 512}, {131072, 128}, {8192, 512}};
    for (auto &sh : shapes) {
        for (int lds = 0; lds < 2; ++lds) {
            for (int rep = 0; rep < 3; ++rep) {
                if (lds) hipLaunchKernelGGL(lds_kernel, dim3(sh[0]), dim3(sh[1]), 16384, 0, d);
                else hipLaunchKernelGGL(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, 0, d);
            }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 20; ++rep) {
                if (lds) hipLaunchKernelGGL(lds_kernel, dim3(sh[0]), dim3(sh[1]), 16384, 0, d);
                else hipLaunchKernelGGL(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, 0, d);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%6d workgroups of %3d threads%s: %.1f us per launch = %.0f workgroups / us\n", sh[0], sh[1], lds ? " (16 KB LDS + barrier)" : "",
                   ms / 20 * 1e3, sh[0] / (ms / 20 * 1e3));
        }
    }
    return 0;
}
