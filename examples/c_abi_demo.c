/*
 * The C ABI from plain C: one forward_features call with device-resident buffers, checked against a brute-force
 * evaluation of the contribution rule on the host.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -Lmolvoxel_amd/csrc -lmvx_hip -Wl,-rpath,$PWD/molvoxel_amd/csrc -lm -o c_abi_demo
 *   ./c_abi_demo            (needs an MI355X; prints "ok" and exits 0)
 *
 * No HIP headers are needed on the caller's side: device memory comes from mvx_alloc / mvx_memcpy.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mvx.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != MVX_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mvx_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(void) {
    enum { D = 16, N = 40, C = 4 };
    const double res = 0.5, width = res * (D - 1), r = 1.0;
    static double coords[N * 3];
    static float feat[N * C], grid[C * D * D * D];
    unsigned seed = 12345u;
    for (int i = 0; i < N * 3; ++i) {
        seed = seed * 1664525u + 1013904223u;
        coords[i] = ((seed >> 8) / 16777216.0 - 0.5) * (width + 2.0);
    }
    for (int i = 0; i < N * C; ++i) {
        seed = seed * 1664525u + 1013904223u;
        feat[i] = (float)((seed >> 8) / 16777216.0);
    }

    mvx_config cfg = {res, 0.5, D, D /* one reference block: no block cull */, MVX_BINARY, 0, 32, 0};
    mvx_handle *h = NULL;
    CHECK(mvx_create(&cfg, &h));
    void *d_coords = NULL, *d_feat = NULL, *d_grid = NULL;
    CHECK(mvx_alloc(h, sizeof coords, &d_coords));
    CHECK(mvx_alloc(h, sizeof feat, &d_feat));
    CHECK(mvx_alloc(h, sizeof grid, &d_grid));
    CHECK(mvx_memcpy(h, d_coords, coords, sizeof coords, MVX_DEVICE, MVX_HOST, NULL));
    CHECK(mvx_memcpy(h, d_feat, feat, sizeof feat, MVX_DEVICE, MVX_HOST, NULL));
    CHECK(mvx_forward_features(h, (const double *)d_coords, d_feat, NULL, r, MVX_RADII_SCALAR, N, C, NULL, d_grid,
                               MVX_DEVICE, MVX_DEVICE, NULL));
    CHECK(mvx_memcpy(h, grid, d_grid, sizeof grid, MVX_HOST, MVX_DEVICE, NULL));

    /* binary density, scalar radius: out[c][i][j][k] = sum of feat[n][c] over atoms inside the box (strict) with
       float32(float32(sqrt(d2)) / float32(r)) <= 1 */
    double worst = 0.0;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j)
            for (int k = 0; k < D; ++k) {
                float ref[C] = {0};
                const double g[3] = {i * res - width / 2, j * res - width / 2, k * res - width / 2};
                for (int n = 0; n < N; ++n) {
                    const double *p = coords + 3 * n;
                    int inside = 1;
                    for (int a = 0; a < 3; ++a) inside = inside && p[a] > -width / 2 - r && p[a] < width / 2 + r;
                    const double dx = p[0] - g[0], dy = p[1] - g[1], dz = p[2] - g[2];
                    const float dist = (float)sqrt((dx * dx + dy * dy) + dz * dz);
                    if (inside && dist / (float)r <= 1.0f)
                        for (int c = 0; c < C; ++c) ref[c] += feat[n * C + c];
                }
                for (int c = 0; c < C; ++c) {
                    const double e = fabs((double)ref[c] - (double)grid[((c * D + i) * D + j) * D + k]);
                    if (e > worst) worst = e;
                }
            }
    CHECK(mvx_free(h, d_coords));
    CHECK(mvx_free(h, d_feat));
    CHECK(mvx_free(h, d_grid));
    CHECK(mvx_destroy(h));
    printf("libmvx_hip %d, max |gpu - brute force| = %.3g\n", mvx_version(), worst);
    if (worst > 1e-5) return 2;
    puts("ok");
    return 0;
}
