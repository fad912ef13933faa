// Store-stream microbenchmark: what the grid write-out can reach for different piece sizes per workgroup.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o tools/micro/store_pattern.bin (built here, run on the GPU box)
// A "molecule" is C x D^3 floats (C = 32, D = 64). A workgroup writes, for each of its CT channels, `px` x-planes x
// `py` y-rows x 256 B (one z row), i.e. pieces of py*256 contiguous bytes at 16-KB (x) and 1-MB (channel) strides -
// exactly the voxelize kernel's pattern for py = 4, px = 2 (64 KB per workgroup in 64 pieces of 1 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int PX, int PY, bool NT>
__global__ void __launch_bounds__(1024) fill(float *out, int nsx, int nsy, int sleep) {
    // grid.x = slab (sy fastest), grid.y = molecule
    const int D = 64, C = 32;
    const int t = blockIdx.x, b = blockIdx.y;
    const int sx = t / nsy, sy = t % nsy;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (sleep) __builtin_amdgcn_s_sleep(64);
    const f4 v = {1.f, 2.f, 3.f, (float)tid};
    // pieces: (c, x) -> PY*256 B contiguous = PY*16 float4
    const int f4_per_piece = PY * 16;
    const int pieces = C * PX;
    for (int i = tid; i < pieces * f4_per_piece; i += nthr) {
        const int piece = i / f4_per_piece, q = i % f4_per_piece;
        const int c = piece / PX, x = piece % PX;
        float *dst = out + ((size_t)b * C + c) * D * D * D + (size_t)(sx * PX + x) * D * D + (size_t)(sy * PY) * D + 4 * q;
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(dst));
        else *reinterpret_cast<f4 *>(dst) = v;
    }
}

template <int PX, int PY, bool NT>
float run(float *out, int B, int threads, int sleep) {
    const int nsx = 64 / PX, nsy = 64 / PY;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill<PX, PY, NT>), dim3(nsx * nsy, B), dim3(threads), 0, 0, out, nsx, nsy, sleep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float m;
        hipEventElapsedTime(&m, e0, e1);
        if (it >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

// D = 128 (cfg-5): a z row is 512 B. MODE 0: the kernel's slab today, 2 x 4 x 64 voxels: per channel and x-plane four HALF
// rows (256 B each, 512 B apart); the other halves belong to the next workgroup. MODE 1: 2 x 2 x 128: two whole rows =
// 1 KB contiguous. MODE 2: 2 x 4 x 128 with 1024 threads (2 KB contiguous). MODE 3: 4 x 1 x 128 (512-B pieces).
template <int MODE>
__global__ void __launch_bounds__(1024) fill128(float *out) {
    const int D = 128, C = 32;
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    const f4 v = {1.f, 2.f, 3.f, (float)tid};
    constexpr int PX = MODE == 3 ? 4 : 2, PY = MODE == 1 ? 2 : (MODE == 3 ? 1 : 4), ZB = MODE == 0 ? 256 : 512; // bytes per row piece
    constexpr int NZ = 512 / ZB;                                                   // workgroups along a row
    const int nsy = D / PY;
    const int zc = t % NZ, sy = (t / NZ) % nsy, sx = t / (NZ * nsy);
    const int f4_row = ZB / 16;
    const int total = C * PX * PY * f4_row;
    for (int i = tid; i < total; i += nthr) {
        const int q = i % f4_row, r = i / f4_row;
        const int y = r % PY, x = (r / PY) % PX, c = r / (PY * PX);
        float *dst = out + ((size_t)b * C + c) * D * D * D + (size_t)(sx * PX + x) * D * D + (size_t)(sy * PY + y) * D + zc * (ZB / 4) + 4 * q;
        __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(dst));
    }
}

template <int MODE>
float run128(float *out, int B, int threads) {
    constexpr int PX = MODE == 3 ? 4 : 2, PY = MODE == 1 ? 2 : (MODE == 3 ? 1 : 4), NZ = MODE == 0 ? 2 : 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill128<MODE>), dim3((128 / PX) * (128 / PY) * NZ, B), dim3(threads), 0, 0, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float m;
        hipEventElapsedTime(&m, e0, e1);
        if (it >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main() {
    const int B = 256;
    const size_t bytes = (size_t)B * 32 * 64 * 64 * 64 * 4;
    float *out;
    if (hipMalloc(&out, bytes) != hipSuccess) return 1;
    auto rep = [&](const char *name, float ms) { printf("%-44s %.3f ms  %.0f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000); };
    rep("px2 py4 (1 KB pieces, 64 KB/WG) 512thr nt", run<2, 4, true>(out, B, 512, 0));
    rep("px2 py4 512thr nt paced", run<2, 4, true>(out, B, 512, 1));
    rep("px2 py4 512thr plain", run<2, 4, false>(out, B, 512, 0));
    rep("px2 py8 (2 KB pieces, 128 KB/WG) 1024thr nt", run<2, 8, true>(out, B, 1024, 0));
    rep("px2 py8 1024thr nt paced", run<2, 8, true>(out, B, 1024, 1));
    rep("px2 py8 512thr nt", run<2, 8, true>(out, B, 512, 0));
    rep("px2 py16 (4 KB pieces, 256 KB/WG) 1024thr nt", run<2, 16, true>(out, B, 1024, 0));
    rep("px1 py8 (2 KB pieces, 64 KB/WG) 512thr nt", run<1, 8, true>(out, B, 512, 0));
    rep("px1 py8 512thr nt paced", run<1, 8, true>(out, B, 512, 1));
    rep("px1 py16 (4 KB pieces, 128 KB/WG) 512thr nt", run<1, 16, true>(out, B, 512, 0));
    rep("px4 py4 (1 KB pieces, 128 KB/WG) 1024thr nt", run<4, 4, true>(out, B, 1024, 0));
    rep("px1 py64 (16 KB pieces, 512 KB/WG) 1024thr nt", run<1, 64, true>(out, B, 1024, 0));
    rep("px4 py4 (1 KB pieces, 128 KB/WG) 512thr nt", run<4, 4, true>(out, B, 512, 0));
    rep("px4 py4 1024thr plain", run<4, 4, false>(out, B, 1024, 0));
    rep("px4 py4 1024thr nt paced", run<4, 4, true>(out, B, 1024, 1));
    rep("px8 py4 (1 KB pieces, 256 KB/WG) 1024thr nt", run<8, 4, true>(out, B, 1024, 0));
    rep("px2 py4 256thr nt", run<2, 4, true>(out, B, 256, 0));
    rep("px2 py4 1024thr nt", run<2, 4, true>(out, B, 1024, 0));
    rep("px2 py2 (512 B pieces, 32 KB/WG) 256thr nt", run<2, 2, true>(out, B, 256, 0));
    rep("px4 py2 (512 B pieces, 64 KB/WG) 512thr nt", run<4, 2, true>(out, B, 512, 0));
    rep("px8 py2 (512 B pieces, 128 KB/WG) 1024thr nt", run<8, 2, true>(out, B, 1024, 0));
    rep("px8 py1 (256 B pieces, 64 KB/WG) 512thr nt", run<8, 1, true>(out, B, 512, 0));
    rep("px16 py1 (256 B pieces, 128 KB/WG) 1024thr nt", run<16, 1, true>(out, B, 1024, 0));
    // 128^3 grids: 32 molecules of 32 x 128^3 floats = the same bytes
    rep("D128 2x4x64  (4 half rows of 256 B) 512thr", run128<0>(out, 32, 512));
    rep("D128 2x2x128 (1 KB contiguous) 512thr", run128<1>(out, 32, 512));
    rep("D128 2x4x128 (2 KB contiguous) 1024thr", run128<2>(out, 32, 1024));
    rep("D128 4x1x128 (512 B pieces) 512thr", run128<3>(out, 32, 512));
    rep("D128 2x4x64  again", run128<0>(out, 32, 512));
    rep("D128 2x2x128 again", run128<1>(out, 32, 512));
    {   // reference: a linear fill of the same bytes
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int it = 0; it < 6; ++it) { hipEventRecord(e0); hipMemsetAsync(out, 0, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1); float m; hipEventElapsedTime(&m, e0, e1); best = std::min(best, m); }
        rep("hipMemsetAsync (linear)", best);
    }
    hipFree(out);
    return 0;
}
