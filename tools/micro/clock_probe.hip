// What does the shader clock do after an idle period? s_memtime counts shader cycles, s_memrealtime a fixed 100 MHz: their ratio
// inside a kernel is the clock the wave ran at. Probe launched (a) after 0.5 s of idle, repeatedly, and (b) during / after a
// stream of large fills (the load a voxelizer batch puts on the chip).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/clock_probe.hip -o tools/micro/clock_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void probe(unsigned long long *out, int spin) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
        out[2] = (unsigned long long)x;
    }
}
__global__ void fill(float4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(0, 0, 0, 0);
}
static double mhz(unsigned long long *h) { return 100.0 * (double)h[0] / (double)h[1]; }

int main() {
    unsigned long long *d, h[3];
    float4 *buf;
    const size_t n = (size_t)1 << 28; // 4 GiB of float4
    hipMalloc(&d, 64);
    hipMalloc(&buf, n * sizeof(float4));
    auto run_probe = [&]() {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 20000);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        return mhz(h);
    };
    run_probe();
    for (int rep = 0; rep < 3; ++rep) {
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        printf("after 0.5 s idle: ");
        for (int i = 0; i < 6; ++i) printf("%.0f ", run_probe());
        printf("MHz (consecutive probes, ~20 us each)\n");
    }
    for (int rep = 0; rep < 2; ++rep) {
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        printf("after 0.5 s idle, fills of 4 GiB (0.7 ms each) with a probe after every 5:");
        for (int g = 0; g < 12; ++g) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill, dim3(256 * 16), dim3(256), 0, 0, buf, n);
            hipDeviceSynchronize();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 5;
            printf(" [%.3f ms/fill, %.0f MHz]", ms, run_probe());
        }
        printf("\n");
    }
    return 0;
}
