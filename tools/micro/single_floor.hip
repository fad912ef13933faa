// What can ONE per-molecule call's store stream reach? A cfg-2 grid is 32 x 64^3 floats = 33.5 MB; the voxelize launch of a
// single molecule (512 workgroups of 512 threads, 64 KB each) takes 10.6 us even with no atoms at all. This fills the same
// grid with other launch shapes and reports (a) the dispatch-packet time (hipExtLaunchKernelGGL events) and (b) host time per
// launch over 200 back-to-back launches (what tools/single_calls.py reports for a call).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/single_floor.hip -o tools/micro/single_floor.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

// slab = 2 x PY x 64 voxels of all CT channels of chunk blockIdx.y: per (channel, x) PY rows of 256 B contiguous
template <int PY, int CT, bool NT>
__global__ void __launch_bounds__(512) fill(float *out, int nsy) {
    const int D = 64;
    const int t = blockIdx.x, cc = blockIdx.y;
    const int sx = t / nsy, sy = t % nsy;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const f4 v = {0.f, 0.f, 0.f, 0.f};
    const int f4_per_piece = PY * 16, pieces = CT * 2;
    for (int i = tid; i < pieces * f4_per_piece; i += nthr) {
        const int piece = i / f4_per_piece, q = i % f4_per_piece;
        const int c = cc * CT + piece / 2, x = piece % 2;
        float *dst = out + (size_t)c * D * D * D + (size_t)(sx * 2 + x) * D * D + (size_t)(sy * PY) * D + 4 * q;
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(dst));
        else *reinterpret_cast<f4 *>(dst) = v;
    }
}

// flat fill: grid-stride 16-B stores
template <bool NT>
__global__ void __launch_bounds__(256) flat(float *out, size_t n4) {
    const f4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out) + i);
        else reinterpret_cast<f4 *>(out)[i] = v;
    }
}

template <typename F>
static void measure(const char *name, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch(nullptr, nullptr);
    hipDeviceSynchronize();
    std::vector<float> ms;
    for (int it = 0; it < 40; ++it) {
        launch(e0, e1);
        hipEventSynchronize(e1);
        float m;
        hipEventElapsedTime(&m, e0, e1);
        ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 400; ++i) launch(nullptr, nullptr);
    hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / 400.0;
    printf("%-58s packet p50 %6.2f us   back-to-back %6.2f us per launch (%.2f TB/s)\n", name, 1e3 * ms[ms.size() / 2], us,
           33.554432 / us);
}

int main() {
    const size_t bytes = (size_t)32 * 64 * 64 * 64 * 4;
    float *out;
    hipMalloc(&out, bytes);
    hipMemset(out, 0, bytes);
#define SLAB(PY, CT, NT, THREADS)                                                                                          \
    measure("slab 2x" #PY "x64, " #CT " channels per workgroup, " #THREADS " threads" #NT, [&](hipEvent_t a, hipEvent_t b) { \
        const dim3 g(32 * (64 / PY), 32 / CT), blk(THREADS);                                                               \
        if (a) hipExtLaunchKernelGGL((fill<PY, CT, true>), g, blk, 0, 0, a, b, 0, out, 64 / PY);                           \
        else hipLaunchKernelGGL((fill<PY, CT, true>), g, blk, 0, 0, out, 64 / PY);                                         \
    })
    SLAB(4, 32, , 512);  // the kernel's shape: 512 workgroups x 64 KB
    SLAB(4, 32, , 256);
    SLAB(4, 16, , 512);  // 1024 workgroups x 32 KB
    SLAB(4, 16, , 256);
    SLAB(4, 8, , 256);   // 2048 x 16 KB
    SLAB(4, 8, , 512);
    SLAB(2, 32, , 512);  // 1024 workgroups x 32 KB, 512-B pieces
    SLAB(2, 16, , 256);  // 2048
    SLAB(4, 4, , 256);   // 4096 x 8 KB
    SLAB(1, 32, , 256);  // 2048 x 16 KB in 256-B pieces
    measure("slab 2x4x64 32 ch 512 thr, plain stores", [&](hipEvent_t a, hipEvent_t b) {
        const dim3 g(512, 1), blk(512);
        if (a) hipExtLaunchKernelGGL((fill<4, 32, false>), g, blk, 0, 0, a, b, 0, out, 16);
        else hipLaunchKernelGGL((fill<4, 32, false>), g, blk, 0, 0, out, 16);
    });
    for (int wgs : {256, 512, 1024, 2048, 4096, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "flat fill, %d workgroups of 256 threads, nt", wgs);
        measure(nm, [&](hipEvent_t a, hipEvent_t b) {
            if (a) hipExtLaunchKernelGGL((flat<true>), dim3(wgs), dim3(256), 0, 0, a, b, 0, out, bytes / 16);
            else hipLaunchKernelGGL((flat<true>), dim3(wgs), dim3(256), 0, 0, out, bytes / 16);
        });
        snprintf(nm, sizeof nm, "flat fill, %d workgroups of 256 threads, plain", wgs);
        measure(nm, [&](hipEvent_t a, hipEvent_t b) {
            if (a) hipExtLaunchKernelGGL((flat<false>), dim3(wgs), dim3(256), 0, 0, a, b, 0, out, bytes / 16);
            else hipLaunchKernelGGL((flat<false>), dim3(wgs), dim3(256), 0, 0, out, bytes / 16);
        });
    }
    {   // hipMemsetAsync
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 400; ++i) hipMemsetAsync(out, 0, bytes, 0);
        hipDeviceSynchronize();
        const auto t1 = std::chrono::steady_clock::now();
        printf("hipMemsetAsync: back-to-back %.2f us per call\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 400.0);
    }
    {   // an empty kernel: the launch floor
        measure("empty launch (flat fill of 0 bytes, 512 workgroups)", [&](hipEvent_t a, hipEvent_t b) {
            if (a) hipExtLaunchKernelGGL((flat<true>), dim3(512), dim3(256), 0, 0, a, b, 0, out, (size_t)0);
            else hipLaunchKernelGGL((flat<true>), dim3(512), dim3(256), 0, 0, out, (size_t)0);
        });
    }
    hipFree(out);
    return 0;
}
