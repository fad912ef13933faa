/*
 * mvx_oracle.c — CPU restatement of the molvoxel numpy backend's voxelization rule.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under molvoxel_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker. The product path is the HIP library
 * (molvoxel_amd/csrc) and fails loudly when that library is missing.
 *
 * Parity pin: this restatement is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by importing the reference numpy backend itself
 * (oracle/gen_golden.py -> tests/golden/); binary outputs are bit-identical,
 * Gaussian outputs agree to <= 1e-6 (float32 exp / summation order only).
 *
 * What is restated (reference file:line, relative to /root/reference):
 *   grid axis, blocks, bounds ......... molvoxel/voxelizer/numpy/voxelizer.py:37-58
 *   box cull (strict, fp64) ........... molvoxel/voxelizer/numpy/voxelizer.py:481-494
 *   per-block cull (strict, fp64) ..... molvoxel/voxelizer/numpy/voxelizer.py:496-527
 *   distance / density ................ molvoxel/voxelizer/numpy/voxelizer.py:531-560
 *       dist = float32( cdist_f64 ) ; dr = dist / r (float32)
 *       binary  : dr <= 1 -> 1.0f
 *       gaussian: expf(-0.5f * (dr/sigma)^2), zero where dr > 1     (all float32)
 *   features: out[c] = sum_n F[n,c]*val   molvoxel/voxelizer/numpy/voxelizer.py:194-236
 *   types   : out[t_n] += val (atom order) molvoxel/voxelizer/numpy/voxelizer.py:344-366
 *   single  : out[0] = sum_n val           molvoxel/voxelizer/numpy/voxelizer.py:457-477
 *   channel-wise features: cull with float32 max radius (NEP-50 quirk: lb - r and
 *       ub + r are evaluated in float32), membership with radii[c]
 *                                          molvoxel/voxelizer/numpy/voxelizer.py:138,213-224
 *
 * The reference walks 8^3 blocks and materialises (V_b, 512) intermediates; this file
 * states the same result as a per-(atom, voxel) rule: an atom contributes to a voxel
 * iff it survives the box cull, the cull of the reference block containing that voxel,
 * and dr <= 1. Inputs are coordinates AFTER centring / random transform (host side).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: cdist does (dx*dx + dy*dy) + dz*dz without FMA.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double resolution;
    int32_t dimension;
    int32_t blockdim;   /* reference `blockdim` (default 8); >= dimension means one block, no block cull */
    int32_t density;    /* 0 gaussian, 1 binary */
    int32_t radii_mode; /* 0 scalar, 1 per-atom array, 2 channel-wise (features only) */
    double sigma;
} ovx_params;

enum { OVX_FEATURES = 0, OVX_TYPES = 1, OVX_SINGLE = 2 };

/* axis[i] = i*res - width/2 with width = res*(D-1): voxelizer.py:41-43, base/voxelizer.py:28 */
static void make_axis(const ovx_params *p, double *axis) {
    const double width = p->resolution * (double)(p->dimension - 1);
    const double half = width / 2.0;
    for (int i = 0; i < p->dimension; ++i) axis[i] = (double)i * p->resolution - half;
}

/* float32 density value given float32 dr (already known to be <= 1) */
static inline float density_value(const ovx_params *p, float dr) {
    if (p->density == 1) return 1.0f;
    const float sig = (float)p->sigma; /* python float is a weak scalar: dr/sigma is a float32 divide */
    float q = dr / sig;
    q = q * q;
    return expf(-0.5f * q);
}

/*
 * Per-axis block admission for one atom: ok[b] != 0 iff the atom is listed for reference
 * block b along this axis (voxelizer.py:500-513). `r` enters as fp64.
 */
static void axis_block_ok(const double *bounds, int nb, double pc, double r, unsigned char *ok) {
    for (int b = 0; b < nb; ++b) {
        int v = 1;
        if (nb > 1) {
            if (b >= 1) v = v && (pc > bounds[b - 1] - r);
            if (b <= nb - 2) v = v && (pc < bounds[b] + r);
        }
        ok[b] = (unsigned char)v;
    }
}

/*
 * Core. chan_kind: OVX_FEATURES (feat N x C row-major), OVX_TYPES (types N int32), OVX_SINGLE.
 * radii: NULL for scalar mode; N floats (per-atom) or C floats (channel-wise).
 * out: C x D x D x D float32, fully overwritten.
 * Returns 0, or a negative code on bad arguments.
 */
static int ovx_run(const ovx_params *p, int chan_kind, const double *coords, const float *feat,
                   const int32_t *types, double r_scalar, const float *radii, int64_t N, int32_t C,
                   float *out) {
    const int D = p->dimension;
    if (D <= 0 || C <= 0 || N < 0) return -1;
    if (p->radii_mode == 2 && chan_kind != OVX_FEATURES) return -2;
    if (p->radii_mode != 0 && radii == NULL) return -3;
    const int bd = p->blockdim > 0 ? p->blockdim : 8;
    const int nb = (D + bd - 1) / bd;
    const int64_t D3 = (int64_t)D * D * D;
    const double res = p->resolution;
    const double width = res * (double)(D - 1);
    const double ub = width / 2.0;
    const double lb = -1 * ub;

    double *axis = (double *)malloc(sizeof(double) * (size_t)D);
    double *bounds = (double *)malloc(sizeof(double) * (size_t)(nb > 1 ? nb - 1 : 1));
    make_axis(p, axis);
    for (int m = 1; m < nb; ++m) bounds[m - 1] = axis[m * bd] + (res / 2.0); /* voxelizer.py:55 */


    /* channel-wise: cull radius is the float32 max of radii (voxelizer.py:138) */
    float rmax32 = 0.0f;
    if (p->radii_mode == 2) {
        rmax32 = radii[0];
        for (int c = 1; c < C; ++c) rmax32 = radii[c] > rmax32 ? radii[c] : rmax32;
    }

    /* ---- per-atom admission (box cull + per-axis block lists), computed once ---- */
    unsigned char *keep = (unsigned char *)malloc((size_t)(N > 0 ? N : 1));
    unsigned char *okb = (unsigned char *)malloc((size_t)(N > 0 ? N : 1) * 3 * (size_t)nb);
    for (int64_t n = 0; n < N; ++n) {
        const double *pc = coords + 3 * n;
        int k = 1;
        double rc; /* fp64 radius used by the block cull */
        if (p->radii_mode == 0) {
            /* scalar: coords > lb - r and coords < ub + r (voxelizer.py:487-488) */
            rc = r_scalar;
            for (int a = 0; a < 3; ++a) k = k && (pc[a] > lb - rc) && (pc[a] < ub + rc);
        } else if (p->radii_mode == 1) {
            /* array: coords + r > lb and coords - r < ub (voxelizer.py:491-492); r float32 -> fp64 */
            rc = (double)radii[n];
            for (int a = 0; a < 3; ++a) k = k && (pc[a] + rc > lb) && (pc[a] - rc < ub);
        } else {
            /* channel-wise: np.float32 scalar => (python float - float32) evaluates in float32 */
            rc = (double)rmax32;
            const double lo = (double)((float)lb - rmax32);
            const double hi = (double)((float)ub + rmax32);
            for (int a = 0; a < 3; ++a) k = k && (pc[a] > lo) && (pc[a] < hi);
        }
        keep[n] = (unsigned char)k;
        if (k)
            for (int a = 0; a < 3; ++a) axis_block_ok(bounds, nb, pc[a], rc, okb + ((size_t)n * 3 + a) * nb);
    }

    const float r_scalar32 = (float)r_scalar; /* np.divide(float32 array, python float) */

    /* ---- per x-plane candidate lists, in atom order (counting sort): an (x-plane, y-band) tile then visits the atoms whose
     *      radius window and x block admit that plane instead of every atom of the molecule ---- */
    int32_t *xlo = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    int32_t *xhi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    int64_t *xstart = (int64_t *)calloc((size_t)D + 1, sizeof(int64_t));
    for (int64_t n = 0; n < N; ++n) {
        xlo[n] = 1;
        xhi[n] = 0;
        if (!keep[n]) continue;
        double rw = (p->radii_mode == 0) ? (double)r_scalar32 : (p->radii_mode == 1 ? (double)radii[n] : (double)rmax32);
        rw = rw * 1.0000002 + 1e-9;
        const double px = coords[3 * n];
        int lo = (int)floor((px - rw - axis[0]) / res) - 1, hi = (int)ceil((px + rw - axis[0]) / res) + 1;
        if (lo < 0) lo = 0;
        if (hi > D - 1) hi = D - 1;
        xlo[n] = lo;
        xhi[n] = hi;
        for (int ix = lo; ix <= hi; ++ix) xstart[ix + 1]++;
    }
    for (int ix = 0; ix < D; ++ix) xstart[ix + 1] += xstart[ix];
    int32_t *xatoms = (int32_t *)malloc(sizeof(int32_t) * (size_t)(xstart[D] > 0 ? xstart[D] : 1));
    {
        int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)D);
        for (int ix = 0; ix < D; ++ix) fill[ix] = xstart[ix];
        for (int64_t n = 0; n < N; ++n)
            for (int ix = xlo[n]; ix <= xhi[n]; ++ix) xatoms[fill[ix]++] = (int32_t)n;
        free(fill);
    }

    /* ---- accumulate: threads own (x-plane, band of y rows) tiles; a tile is accumulated in a thread-local buffer
     *      [C][rows][D] (atoms visited in order => deterministic, same sums as a voxel-by-voxel loop) and written out once,
     *      row by row: every output row is written by exactly one tile, zeros included - no memset, no strided
     *      read-modify-write of the 33-MB grid. ~4 tiles per thread, so every core the host offers has work. ---- */
    int ytiles = 1;
#ifdef _OPENMP
    ytiles = (4 * omp_get_max_threads() + D - 1) / D;
#endif
    if (ytiles > (D + 3) / 4) ytiles = (D + 3) / 4;
    if (ytiles < 1) ytiles = 1;
    const int rows_max = (D + ytiles - 1) / ytiles + 1;
#pragma omp parallel
    {
    float *buf = (float *)malloc(sizeof(float) * (size_t)C * (size_t)rows_max * (size_t)D);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int ix = 0; ix < D; ++ix) {
      for (int ty = 0; ty < ytiles; ++ty) {
        const int bx = ix / bd;
        const int ty0 = (int)((int64_t)D * ty / ytiles), ty1 = (int)((int64_t)D * (ty + 1) / ytiles) - 1;
        const int rows = ty1 - ty0 + 1;
        if (rows <= 0) continue;
        const int64_t plane = (int64_t)rows * D; /* floats per channel in the tile buffer */
        memset(buf, 0, sizeof(float) * (size_t)C * (size_t)plane);
        for (int64_t q = xstart[ix]; q < xstart[ix + 1]; ++q) {
            const int64_t n = xatoms[q];
            const unsigned char *okx = okb + ((size_t)n * 3 + 0) * nb;
            const unsigned char *oky = okb + ((size_t)n * 3 + 1) * nb;
            const unsigned char *okz = okb + ((size_t)n * 3 + 2) * nb;
            if (!okx[bx]) continue;
            const double px = coords[3 * n], py = coords[3 * n + 1], pz = coords[3 * n + 2];
            /* widest radius that can matter for this atom, for the conservative voxel window */
            double rw;
            if (p->radii_mode == 0) rw = (double)r_scalar32;
            else if (p->radii_mode == 1) rw = (double)radii[n];
            else rw = (double)rmax32;
            rw = rw * 1.0000002 + 1e-9;
            const double dx = px - axis[ix];
            if (fabs(dx) > rw) continue;
            int jlo = (int)floor((py - rw - axis[0]) / res) - 1, jhi = (int)ceil((py + rw - axis[0]) / res) + 1;
            int klo = (int)floor((pz - rw - axis[0]) / res) - 1, khi = (int)ceil((pz + rw - axis[0]) / res) + 1;
            if (jlo < 0) jlo = 0;
            if (klo < 0) klo = 0;
            if (jhi > D - 1) jhi = D - 1;
            if (khi > D - 1) khi = D - 1;
            if (jlo < ty0) jlo = ty0; /* this tile's rows only */
            if (jhi > ty1) jhi = ty1;
            const double dx2 = dx * dx;
            for (int iy = jlo; iy <= jhi; ++iy) {
                if (!oky[iy / bd]) continue;
                const double dy = py - axis[iy];
                const double dxy2 = dx2 + dy * dy;
                for (int iz = klo; iz <= khi; ++iz) {
                    if (!okz[iz / bd]) continue;
                    const double dz = pz - axis[iz];
                    const double d2 = dxy2 + dz * dz;       /* (dx^2 + dy^2) + dz^2, no FMA */
                    const float dist = (float)sqrt(d2);     /* cdist fp64 -> astype(float32) */
                    const int64_t vox = (int64_t)(iy - ty0) * D + iz; /* inside the tile */
                    if (p->radii_mode == 2) {
                        for (int c = 0; c < C; ++c) {
                            const float dr = dist / radii[c];
                            if (dr <= 1.0f) buf[(int64_t)c * plane + vox] += feat[n * C + c] * density_value(p, dr);
                        }
                        continue;
                    }
                    const float rr = (p->radii_mode == 0) ? r_scalar32 : radii[n];
                    const float dr = dist / rr;
                    if (!(dr <= 1.0f)) continue;
                    const float val = density_value(p, dr);
                    if (chan_kind == OVX_FEATURES) {
                        const float *f = feat + n * C;
                        for (int c = 0; c < C; ++c) buf[(int64_t)c * plane + vox] += f[c] * val;
                    } else if (chan_kind == OVX_TYPES) {
                        buf[(int64_t)types[n] * plane + vox] += val;
                    } else {
                        buf[vox] += val;
                    }
                }
            }
        }
        for (int c = 0; c < C; ++c)
            memcpy(out + (int64_t)c * D3 + ((int64_t)ix * D + ty0) * D, buf + (int64_t)c * plane, sizeof(float) * (size_t)plane);
      }
    }
    free(buf);
    }
    free(xlo);
    free(xhi);
    free(xstart);
    free(xatoms);
    free(axis);
    free(bounds);
    free(keep);
    free(okb);
    return 0;
}

int ovx_forward_features(const ovx_params *p, const double *coords, const float *feat, double r_scalar,
                         const float *radii, int64_t N, int32_t C, float *out) {
    return ovx_run(p, OVX_FEATURES, coords, feat, NULL, r_scalar, radii, N, C, out);
}

int ovx_forward_types(const ovx_params *p, const double *coords, const int32_t *types, double r_scalar,
                      const float *radii, int64_t N, int32_t C, float *out) {
    for (int64_t n = 0; n < N; ++n)
        if (types[n] < 0 || types[n] >= C) return -4;
    return ovx_run(p, OVX_TYPES, coords, NULL, types, r_scalar, radii, N, C, out);
}

int ovx_forward_single(const ovx_params *p, const double *coords, double r_scalar, const float *radii,
                       int64_t N, float *out) {
    return ovx_run(p, OVX_SINGLE, coords, NULL, NULL, r_scalar, radii, N, 1, out);
}

int ovx_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ovx_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
