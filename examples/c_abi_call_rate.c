/*
 * Host cost of one per-molecule call through the C ABI alone (no Python): N back-to-back mvx_forward_types calls on a
 * cfg-3-sized molecule, time to ENQUEUE them and time until the queue has drained.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_call_rate.c -Lmolvoxel_amd/csrc -lmvx_hip -Wl,-rpath,$PWD/molvoxel_amd/csrc -o c_abi_call_rate
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "mvx.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != MVX_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mvx_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
    enum { D = 48, N = 1000, C = 4 };
    const int calls = argc > 1 ? atoi(argv[1]) : 2000;
    const double res = 0.5, width = res * (D - 1);
    static double coords[N * 3];
    static int types[N];
    unsigned seed = 3u;
    for (int i = 0; i < N * 3; ++i) {
        seed = seed * 1664525u + 1013904223u;
        coords[i] = ((seed >> 8) / 16777216.0 - 0.5) * width;
    }
    for (int i = 0; i < N; ++i) {
        seed = seed * 1664525u + 1013904223u;
        types[i] = (int)((seed >> 10) % C);
    }
    mvx_config cfg = {res, 0.5, D, 8, MVX_BINARY, 0, 32, 0};
    mvx_handle *h = NULL;
    CHECK(mvx_create(&cfg, &h));
    void *d_coords = NULL, *d_types = NULL, *d_grid = NULL;
    CHECK(mvx_alloc(h, sizeof coords, &d_coords));
    CHECK(mvx_alloc(h, sizeof types, &d_types));
    CHECK(mvx_alloc(h, (size_t)C * D * D * D * sizeof(float), &d_grid));
    CHECK(mvx_memcpy(h, d_coords, coords, sizeof coords, MVX_DEVICE, MVX_HOST, NULL));
    CHECK(mvx_memcpy(h, d_types, types, sizeof types, MVX_DEVICE, MVX_HOST, NULL));
    for (int i = 0; i < 50; ++i)
        CHECK(mvx_forward_types(h, (const double *)d_coords, (const int32_t *)d_types, NULL, 1.0, MVX_RADII_SCALAR, N, C, NULL, d_grid,
                                MVX_DEVICE, MVX_DEVICE, NULL));
    static float one[4];
    CHECK(mvx_memcpy(h, one, d_grid, sizeof one, MVX_HOST, MVX_DEVICE, NULL)); /* drains the queue */
    const double t0 = now_us();
    for (int i = 0; i < calls; ++i)
        CHECK(mvx_forward_types(h, (const double *)d_coords, (const int32_t *)d_types, NULL, 1.0, MVX_RADII_SCALAR, N, C, NULL, d_grid,
                                MVX_DEVICE, MVX_DEVICE, NULL));
    const double t1 = now_us();
    CHECK(mvx_memcpy(h, one, d_grid, sizeof one, MVX_HOST, MVX_DEVICE, NULL));
    const double t2 = now_us();
    printf("cfg-3 molecule, %d calls through the C ABI: %.2f us per call to enqueue, %.2f us per call with the queue drained\n", calls,
           (t1 - t0) / calls, (t2 - t0) / calls);
    CHECK(mvx_free(h, d_coords));
    CHECK(mvx_free(h, d_types));
    CHECK(mvx_free(h, d_grid));
    CHECK(mvx_destroy(h));
    return 0;
}
