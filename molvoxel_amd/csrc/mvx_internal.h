// Internal declarations shared by the kernel TU (mvx_kernels.hip) and the C-ABI TU (mvx_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvx.h"

namespace mvx {

// Sub-tile owned by one wave (one voxel per lane): SUBX x SUBY x SUBZ = 64 voxels. 2 x 4 x 8 keeps the footprint
// compact (Minkowski volume with a radius-2-voxel ball: 497 voxels; 440 for 4^3; 1442 for 1 x 1 x 64) while eight
// waves side by side along z cover whole 256-B rows at D = 64 (4 y-rows of one (channel, x) plane = 1 KiB
// contiguous). Measured on cfg-2 with the slab-line kernel: 2x4x8 0.400 ms, 4x4x4 (128-B runs) 0.415 ms.
constexpr int SUBX_SH = 1, SUBY_SH = 2, SUBZ_SH = 3;
constexpr int SUBX = 1 << SUBX_SH, SUBY = 1 << SUBY_SH, SUBZ = 1 << SUBZ_SH;
static_assert(SUBX * SUBY * SUBZ == 64, "one voxel per lane");
constexpr int RPC = SUBX * SUBY; // tile rows per channel

// Per-atom record written by the prep kernel and consumed by the voxelize kernels (64 B, AoS so
// that one 16-lane dword load moves a whole record into LDS).
struct __attribute__((aligned(16))) AtomRec {
    double px, py, pz; // coordinates after centring / transform (fp64, reference op order)
    double T;          // membership threshold on d2: contributes iff d2 <= T (exact restatement of
                       // float32(float32(sqrt(d2))/r) <= 1, or of sqrt(d2)/r <= 1 in float64 for float64 grids);
                       // negative = never
    float k;           // gaussian: value = exp2(k * d2), k = -0.5*log2(e)/(r*sigma)^2
    int32_t type;      // forward_types channel
    uint32_t xr, yr, zr; // admitted voxel index range per axis, lo | hi << 16 (inclusive); 0x0000ffff = empty
    uint32_t pad[3];   // float64 grids: pad[1..2] = the float64 gaussian coefficient (exp(c * d2))
};
static_assert(sizeof(AtomRec) == 64, "AtomRec must be 64 bytes");

enum Mode { MODE_FEATURES = 0, MODE_TYPES = 1, MODE_SINGLE = 2 };
// radius source seen by the prep kernel
enum RadiiSrc { RAD_SCALAR = 0, RAD_ATOM = 1, RAD_CHANNEL_FEATURES = 2, RAD_CHANNEL_BY_TYPE = 3 };

struct Geom {
    double res;  // resolution
    double half; // width / 2, width = res * (D - 1)
    int32_t D;
    int32_t bd;  // reference blockdim (cull emulation)
    int32_t nb;  // ceil(D / bd)
    uint32_t bd_inv; // ceil(2^32 / bd): v / bd == __umulhi(v, bd_inv) for voxel indices (v * bd < 2^32; bd >= 2)
    double inv_res;   // 1.0 / res
    double inv_pitch; // 1.0 / (bd * res)
};

struct PrepArgs {
    const double *coords;   // (total, 3)
    const void *radii;      // per RadiiSrc; float, or double when precision == 64
    const int32_t *types;   // (total,) or null
    const void *features;   // (total, C) or null; float / double
    int32_t mode;           // Mode: which channel weights go behind the records
    int32_t Cpad;           // channel weights per atom in wbuf (zero padded)
    const int64_t *offsets; // device, B + 1
    const mvx_xform *xforms; // device, B records, or null
    mvx_xform xf_one;       // xforms == null: the transform of the launch's only molecule (flags == 0: none)
    const void *chan_aux;   // device: [0] = max channel radius (float / double) for RAD_CHANNEL_FEATURES
    int32_t precision;      // 32 | 64: element type of radii, features, packed weights and the grid
    int64_t first;          // atoms [first, total) are processed by this launch (pipelined chunks)
    int64_t total;
    int32_t B;
    int32_t C;
    double radius_scalar;
    double T_scalar;        // RAD_SCALAR, float32 grids: d2_threshold(float(radius)) and gauss_coeff, evaluated once on the
    float k_scalar;         // host (scalar_radius_constants)
    int32_t radii_src;
    int32_t density;
    float sigma32;
    double sigma64;    // the Gaussian sigma as the reference holds it (python float): float64 grids
    Geom g;
    AtomRec *rec;      // per-atom records
    void *wbuf;        // packed channel weights (Cpad per atom: features zero padded / one-hot type / 1), or null when
                       // the voxelize kernels read the caller's feature rows directly (features, C == Cpad)
    uint2 *xp;         // what the binning pass scans, 8 B per atom: {admitted x range, packed y/z ranges in slab units}
};

struct VoxParams { // by-value kernel parameters (scalars only: pointers are separate __restrict__ arguments)
    double res, half;
    int32_t D, C, B;
    int32_t nsx, nsy, nzc, ncc; // slabs along x, along y, z chunks, channel chunks
    uint32_t nsy_inv, nzc_inv; // ceil(2^32 / d): n / d == __umulhi(n, inv) for the slab ids used here (n * d < 2^32)
    uint32_t nsx_inv, ncc_inv; // ... the same for nsx (slabs in x-fastest order, grids of several slabs per row) and the channel chunks
    int32_t b0;            // first molecule of this launch (blockIdx.y = (molecule - b0) * ncc + channel chunk)
    int32_t c0;            // first channel of this launch's chunks (voxelize_kernel: a remainder launch with a narrower kernel)
    int32_t NW;            // waves per workgroup = 4^3 sub-tiles per slab
    uint32_t nslab;        // nsx * nsy * nzc: slabs per molecule (the kernels' line index; not gridDim.x - an implicit argument, one more
                           // scalar round trip in front of the line's header)
    int32_t w_stride;      // floats between the channel weights of consecutive atoms
    int32_t dcap;          // candidate rows staged per round
    int32_t vec_store;     // D % 4 == 0 and out 16-B aligned
    int32_t pace;          // 1: empty slabs hold their stores back ~1.7 us (launches of more than 4096 workgroups); 2: light slabs pace their write-out rounds too (>= 49 152)
    int32_t xcd_ranges;    // 1: every XCD takes a contiguous range of slabs (run-wise write-out of whole-row slabs; gridDim.x = 8 ceil(T / 8))
    double sigma;          // float64 grids: the Gaussian sigma as the reference holds it (python float)
#ifdef MVX_DIAG
    int32_t dbg;           // diagnostic builds: run-time ablation switches of voxelize_pair_kernel
#endif
};

// voxelize_pair_kernel: the atoms as the caller passed them (PrepArgs without workspace pointers) plus, for a
// single molecule, its extent and transform by value (no metadata upload).
struct DirectArgs {
    PrepArgs pa;   // rec / wbuf / xp / chan_aux unused; offsets / xforms: device arrays, or null for one molecule
                   // (its transform then is pa.xf_one)
    int64_t N;     // atoms of the only molecule when pa.offsets is null
};

// Channel-wise radii for features (numpy/voxelizer.py:213-224: one membership test and one density per channel): channels
// that share a radius share both. chan_aux_kernel numbers the distinct radii of every chunk of 32 channels (slots, by
// descending radius; one ChanGroups per chunk) and the grouped voxelize launch evaluates, per candidate pair, one threshold
// test and one exp2 per SLOT and feeds the matrix cores the weight row masked to that slot's channels - d2, staging, culls
// and the stores are shared by all slots. 32 channels have at most 32 distinct radii: there is no fallback kernel.
constexpr int CHAN_GROUP_SLOTS = 32;
struct ChanGroups {
    int32_t nslots, pad[3];
    struct {
        double T;  // d2_threshold(radius)
        float k;   // gauss_coeff(radius, sigma)
        int32_t pad;
    } slot[CHAN_GROUP_SLOTS];
};

struct VoxArgs {
    const unsigned *rec;   // per-atom records (16 words each)
    const unsigned *w;     // channel weights: the caller's feature rows or prep's packed copy (p.w_stride apart)
    const uint2 *xlist;    // x-slab lists (xbin_kernel)
    const uint2 *slist;    // per-slab candidate lines (xbin_kernel), SLOTS entries each
    const uint2 *slist_ext; // their extensions (entries 64..255), EXT_SLOTS entries each
    const int64_t *offsets; // device copy of the batch offsets (x-list path only), or null: one molecule of n_one atoms
    int64_t n_one;
    const double *Tc;      // channel-wise features: per-channel d2 thresholds (float64 grids: the radii themselves)
    const float *kc;       //                        per-channel gaussian coefficients
    void *out;             // (B, C, D, D, D) float, or double for float64 grids
    int32_t narrow_sub;    // test / A-B switch ("narrow_sub"): sub-tiles per wave of narrow chunks - 1 (voxelize_kernel), 2, 4; 0 = the rule
    VoxParams p;
};

// launchers (host side: mvx_prep.hip, mvx_slab.hip, mvx_pair.hip, mvx_f64.hip)
// chan_slot: C ints, the slot of every channel (grouped launch)
hipError_t launch_chan_aux(const float *radii, int32_t C, int32_t density, float sigma32, float *rmax, ChanGroups *groups,
                           int32_t *chan_slot, hipStream_t s);
hipError_t launch_chan_aux64(const double *radii, int32_t C, int32_t density, double sigma, double *rmax, double *Tc, double *kc,
                             hipStream_t s);
hipError_t launch_prep(const PrepArgs &a, hipStream_t s);
hipError_t launch_xbin(const uint2 *xp, const int64_t *offsets, int64_t n_one, int32_t b0, int32_t nb, int64_t max_atoms, int32_t nsx, int32_t nsy,
                       int32_t nzc, int32_t NW, uint2 *xlist, uint2 *slist, uint2 *slist_ext, hipStream_t s);
constexpr int SLAB_LINE_ENTRIES = 64;  // = SLOTS in mvx_device.h
constexpr int SLAB_EXT_ENTRIES = 192;  // = EXT_SLOTS
hipError_t launch_transform(const double *coords, int64_t N, const mvx_xform *xf_dev, double *out, hipStream_t s);
// ct: channels per thread (1, 4, 8, 16, 32); lane_range: per-lane index-range check needed
// voxelize molecules [a.p.b0, a.p.b0 + nb): every slab, whatever its candidate count (line, line + extension, x-list)
hipError_t launch_voxelize(const VoxArgs &a, int32_t nb, int32_t ct, bool gauss, bool lane_range, hipStream_t s);
// channel-wise features, grouped by radius: a.Tc must point at the chunks' ChanGroups tables, a.kc at the channels' slots;
// chunks of 32 channels (a.p.ncc = ceil(C / 32)), feature rows read in place
hipError_t launch_voxelize_grouped(const VoxArgs &a, int32_t nb, bool gauss, bool lane_range, hipStream_t s);
// float64 grids: every slab of the whole batch through the general slab loop (ct <= 16; a.p.dcap must be 64)
hipError_t launch_voxelize64(const VoxArgs &a, int32_t ct, bool gauss, bool chanwise, bool lane_range, hipStream_t s);
// the whole call in one launch (voxelize_pair_kernel, mvx_pair.hip: float32 grids, NW <= 8): no workspace, no pre-pass.
// max_atoms: the largest molecule of the call (sizes the per-wave candidate lists); lane_range: sub-tiles cut by reference blocks
hipError_t launch_voxelize_direct(const DirectArgs &d, const VoxParams &p, int64_t max_atoms, float *out, int32_t ct, bool gauss,
                                  bool lane_range, hipStream_t s);
void scalar_radius_constants(double radius_scalar, float sigma32, bool gauss, double *T, float *k);
// profiled launches: the next voxelize launch on this thread carries these events on its own dispatch packet
void set_launch_events(hipEvent_t start, hipEvent_t stop);
bool launch_events_pending();
#ifdef MVX_DIAG
hipError_t set_diag_buffer(void *p);
hipError_t set_diag_buffer_xb(void *p);
#endif
size_t voxelize_lds_bytes(int32_t ct, int32_t NW, int32_t crmax);
int32_t voxelize_dcap(int32_t ct, int32_t NW);

} // namespace mvx
