/*
 * mvx.h — C ABI of the MI355X-native molecular voxelizer (libmvx_hip.so).
 *
 * This is the drop-in boundary for the hot path of SeonghwanSeo/molvoxel:
 * Voxelizer.forward_features / forward_types / forward_single (and the loop over a batch of
 * molecules the reference's timing harness runs, test/test_time_numpy.py:11-15).
 * The reference has no FFI layer of its own (it is pure Python); what it would bind is
 * exactly this header, from a new backend module molvoxel/voxelizer/hip/voxelizer.py via
 * ctypes (see INTEGRATION.md for the stub). Citations are file:line in the reference tree.
 *
 * Conventions
 *   - every function returns 0 (MVX_OK) or a negative mvx_status; mvx_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - no ownership transfer: every buffer is caller-owned. Pointers are tagged host/device by
 *     the *_kind arguments (MVX_HOST / MVX_DEVICE). Device pointers must belong to the handle's
 *     device. A 16-byte aligned `out` with dimension % 4 == 0 gets 16-B stores; any other
 *     float-aligned `out` (e.g. slice i of a batch grid of odd dimension) is written run by run: aligned
 *     16-B stores inside each contiguous run of the grid, 4-B stores at its ends (float64 grids: 8-B stores).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). With MVX_DEVICE
 *     outputs the call is asynchronous on that stream; with MVX_HOST outputs it returns after
 *     the copy back has completed.
 *   - a handle is not re-entrant: one host thread at a time. Calls may arrive on different streams:
 *     the handle's workspace is shared, so a call on another stream than the previous call's first
 *     makes its stream wait for the previous stream (hipStreamWaitEvent; no host synchronisation).
 *     A stream must stay alive until the next call on the handle (or mvx_destroy) has returned.
 *   - argument-shape errors are the Python layer's AssertionErrors (same messages as the
 *     reference, molvoxel/voxelizer/numpy/voxelizer.py:181-192, 327-342, 443-455) and are
 *     raised before the call; this library only validates what it needs to stay memory-safe.
 *
 * Numerical contract (verified against the imported reference by the goldens)
 *   An atom n at p (fp64, after centring / transform) with radius r contributes to voxel
 *   (i,j,k), g[i] = i*res - res*(D-1)/2, iff it passes
 *     1. the box cull            molvoxel/voxelizer/numpy/voxelizer.py:481-494  (strict, fp64)
 *     2. the cull of the reference block (blockdim) holding the voxel   :496-527 (strict, fp64)
 *     3. float32( float32(sqrt_f64((dx^2+dy^2)+dz^2)) / float32(r) ) <= 1          :544-555
 *   with value 1 (binary) or exp(-0.5*(dr/sigma)^2) in float32 (gaussian)            :557-560.
 *   Membership (1-3) is reproduced exactly; gaussian values agree to ~1e-6.
 *   A precision-64 handle keeps step 3 and the values in float64, as the reference does with
 *   precision=64 (:33-34): sqrt_f64(d2) / r <= 1, exp(-0.5*((dr/sigma)^2)) in float64; values agree to ~1e-15.
 */
#ifndef MVX_H
#define MVX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVX_VERSION 140 /* 0.1.4: mvx_plan_call (the decision table as a pure function), channel-wise radii grouped per chunk of 32 channels, narrow chunks in candidate pairs; 0.1.3: one voxelize launch per batched call, channel-wise radii grouped on the device; 0.1.2: mvx_xform.center_ptr, stream hand-over, unaligned out, mvx_debug_set_option */

typedef enum mvx_status {
    MVX_OK = 0,
    MVX_ERR_INVALID = -1, /* bad argument */
    MVX_ERR_HIP = -2,     /* a HIP runtime call failed (message has the hipError string) */
    MVX_ERR_NO_DEVICE = -3,
    MVX_ERR_ALLOC = -4
} mvx_status;

enum mvx_memkind { MVX_HOST = 0, MVX_DEVICE = 1 };
enum mvx_density { MVX_GAUSSIAN = 0, MVX_BINARY = 1 }; /* base/voxelizer.py:13  DENSITY_TYPE_LIST */
enum mvx_radii {                                         /* base/voxelizer.py:12  RADII_TYPE_LIST */
    MVX_RADII_SCALAR = 0,  /* one python float for every atom */
    MVX_RADII_ATOM = 1,    /* mvx_real (N,)  "atom-wise"    */
    MVX_RADII_CHANNEL = 2  /* mvx_real (C,)  "channel-wise" */
};
enum mvx_xform_flags {
    MVX_XF_CENTER = 1,    /* p = p - center first */
    MVX_XF_ROTATE = 2,    /* p = q * p * q^-1 */
    MVX_XF_TRANSLATE = 4, /* p = p + trans (twice when MVX_XF_ROTATE is also set, as the reference does) */
    MVX_XF_RECENTER = 8,  /* p = p + center after the rotation (do_transform with a center, numpy/transform.py:51-54) */
    MVX_XF_CENTER_PTR = 16 /* the centre is read from center_ptr (3 doubles in the memory `in_kind` names) instead of
                              center[]: a device-resident `center` tensor never has to visit the host */
};

/*
 * Geometry + density of one voxelizer. Replaces the constructor state of
 * BaseVoxelizer.__init__ (base/voxelizer.py:15-38) and numpy Voxelizer.__init__/_setup_block
 * (numpy/voxelizer.py:22-58).
 */
typedef struct mvx_config {
    double resolution; /* base/voxelizer.py:26 */
    double sigma;      /* base/voxelizer.py:37-38, default 0.5; ignored for binary */
    int32_t dimension; /* D = H = W */
    int32_t blockdim;  /* reference `blockdim` whose per-block cull is emulated (numpy/voxelizer.py:38,55);
                          <= 0 means the reference default 8; >= dimension means one block (no block cull) */
    int32_t density;   /* enum mvx_density */
    int32_t device;    /* HIP device ordinal */
    int32_t precision; /* 32 (or 0) | 64: the `precision` argument of Voxelizer.__init__ (numpy/voxelizer.py:28,33-34):
                          element type of features, radii and the grid, and the type distances, densities and sums
                          are evaluated in */
    int32_t reserved;  /* 0 */
} mvx_config;

/* float for a precision-32 handle, double for a precision-64 handle (the reference's `self.fp`). */
typedef void mvx_real;

/*
 * Per-molecule rigid transform applied on the device before voxelization, in exactly the
 * reference's fp64 operation order: p = coords - center (numpy/voxelizer.py:120-121), then
 * do_transform(p, None, translation, quaternion) (numpy/transform.py:44-60, _quaternion.py:24-50),
 * including the reference's double application of the translation when a rotation is present.
 */
typedef struct mvx_xform {
    double center[3];
    double quat[4];   /* (q0, q1, q2, q3) as returned by random_quaternion, _quaternion.py:13-21 */
    float trans[3];   /* float32 like numpy/transform.py:76 */
    uint32_t flags;   /* enum mvx_xform_flags */
    const double *center_ptr; /* MVX_XF_CENTER_PTR: where the centre lives (same memory kind as coords) */
} mvx_xform;

typedef struct mvx_handle mvx_handle;

int mvx_version(void);
const char *mvx_last_error(void);
int mvx_device_count(int *count);

/* Replaces create_voxelizer(..., library=...) -> Voxelizer(...)  (molvoxel/__init__.py:25-40). */
int mvx_create(const mvx_config *cfg, mvx_handle **out);
int mvx_destroy(mvx_handle *h);
/* Replaces the density_type property setter (base/voxelizer.py:65-70). */
int mvx_set_density(mvx_handle *h, int32_t density, double sigma);
/*
 * Cross-call overlap for loops of large batched calls (no counterpart in the reference, which is synchronous).
 * A batched call is a pre-pass over the atoms (records, candidate lines: ~7 % of a 256-molecule cfg-2 step) followed
 * by the voxelize launches. With enable != 0 the handle keeps two workspace sets and runs the pre-pass of call k+1 on
 * an internal side stream while call k's voxelize launches still occupy the caller's stream; the voxelize launches
 * of call k+1 follow on the caller's stream as usual, so OUTPUTS keep plain stream-order semantics.
 * The INPUTS contract changes: the side stream does not wait for the caller's stream, so coords / features / types /
 * radii must be complete when the call is made (uploaded and synchronised earlier, or produced before a host-side
 * synchronisation), and must stay unchanged until the call's launches have run - not merely be ordered before the call
 * on the stream. Applies to MVX_DEVICE inputs and outputs on the batched three-launch path; other calls are unaffected.
 */
int mvx_set_overlap(mvx_handle *h, int32_t enable);

/*
 * Batched entry points: B molecules stored back to back, molecule b owning atoms
 * [offsets[b], offsets[b+1]). `offsets` (B+1 int64) and `xforms` (B records, may be NULL =
 * identity) are host pointers. coords / features / types / radii share `in_kind`.
 * out is (B, C, D, D, D) mvx_real, fully overwritten (zeros included), `out_kind` tagged.
 *
 *   radii_type SCALAR : radius = radius_scalar for every atom (radii ignored, may be NULL)
 *              ATOM   : radii[sumN]
 *              CHANNEL: radii[C], shared by all molecules. forward_types gathers radii[types]
 *                       (numpy/voxelizer.py:284-285); forward_features tests each channel with its
 *                       own radius and culls with max(radii) (numpy/voxelizer.py:138,213-224).
 *
 * mvx_forward_features_batch replaces Voxelizer.forward_features (numpy/voxelizer.py:97-169)
 *   features: (sumN, C) mvx_real row-major.
 * mvx_forward_types_batch replaces Voxelizer.forward_types (numpy/voxelizer.py:240-315)
 *   types: (sumN,) int32 in [0, C); out has C channels (C may exceed max(types)+1, numpy/voxelizer.py:337).
 * mvx_forward_single_batch replaces Voxelizer.forward_single (numpy/voxelizer.py:370-436)
 *   out is (B, 1, D, D, D).
 */
int mvx_forward_features_batch(mvx_handle *h, const double *coords, const mvx_real *features, const mvx_real *radii,
                               double radius_scalar, int32_t radii_type, const int64_t *offsets,
                               const mvx_xform *xforms, int32_t B, int32_t C, mvx_real *out, int32_t in_kind,
                               int32_t out_kind, void *stream);
int mvx_forward_types_batch(mvx_handle *h, const double *coords, const int32_t *types, const mvx_real *radii,
                            double radius_scalar, int32_t radii_type, const int64_t *offsets,
                            const mvx_xform *xforms, int32_t B, int32_t C, mvx_real *out, int32_t in_kind,
                            int32_t out_kind, void *stream);
int mvx_forward_single_batch(mvx_handle *h, const double *coords, const mvx_real *radii, double radius_scalar,
                             int32_t radii_type, const int64_t *offsets, const mvx_xform *xforms, int32_t B,
                             mvx_real *out, int32_t in_kind, int32_t out_kind, void *stream);

/* Single-molecule forms (B = 1, xform may be NULL): the reference's per-call signature. */
int mvx_forward_features(mvx_handle *h, const double *coords, const mvx_real *features, const mvx_real *radii,
                         double radius_scalar, int32_t radii_type, int64_t N, int32_t C, const mvx_xform *xform,
                         mvx_real *out, int32_t in_kind, int32_t out_kind, void *stream);
int mvx_forward_types(mvx_handle *h, const double *coords, const int32_t *types, const mvx_real *radii,
                      double radius_scalar, int32_t radii_type, int64_t N, int32_t C, const mvx_xform *xform,
                      mvx_real *out, int32_t in_kind, int32_t out_kind, void *stream);
int mvx_forward_single(mvx_handle *h, const double *coords, const mvx_real *radii, double radius_scalar,
                       int32_t radii_type, int64_t N, const mvx_xform *xform, mvx_real *out, int32_t in_kind,
                       int32_t out_kind, void *stream);

/*
 * Replaces do_transform on an (N,3) fp64 point cloud (numpy/transform.py:44-60): out = transformed coords.
 * Exposed so that RandomTransform/T objects can run on device-resident coordinates.
 */
int mvx_transform_coords(mvx_handle *h, const double *coords, int64_t N, const mvx_xform *xform, double *out,
                         int32_t in_kind, int32_t out_kind, void *stream);

/* Kernel timing, measured with HIP events recorded on the launch stream immediately before and
 * after the dominant (voxelize) kernel of every call while profiling is enabled. Recording does not
 * synchronise; up to MVX_PROFILE_RING launches are kept. mvx_profile_read synchronises on the last
 * event, writes the per-launch durations (ms, oldest first) and resets the ring. bench.py uses it
 * for `roofline.achieved`; mvx_last_kernel_ms is the single-launch convenience form. */
#define MVX_PROFILE_RING 1024
int mvx_set_profiling(mvx_handle *h, int32_t enable);
int mvx_profile_read(mvx_handle *h, float *ms, int32_t capacity, int32_t *count);
int mvx_last_kernel_ms(mvx_handle *h, float *ms);

/* Device memory helpers for callers without a device allocator of their own (torch-less use;
 * numpy/voxelizer.py:60-70 get_empty_grid's role on the device). */
int mvx_alloc(mvx_handle *h, int64_t bytes, void **ptr);
int mvx_free(mvx_handle *h, void *ptr);
int mvx_memcpy(mvx_handle *h, void *dst, const void *src, int64_t bytes, int32_t dst_kind, int32_t src_kind,
               void *stream);
int mvx_memset_zero(mvx_handle *h, void *ptr, int64_t bytes, void *stream);
int mvx_stream_sync(mvx_handle *h, void *stream);

/* Testing aid: copy the first n 64-byte atom records of the last call (px, py, pz, T as 4 doubles;
 * k float; type int32; x/y/z admitted voxel ranges as lo | hi << 16; 12 B pad) to host memory.
 * Synchronises the stream. Lets the tests check the prep stage (transform, culls, thresholds) alone. */
int mvx_debug_read_records(mvx_handle *h, void *host_dst, int64_t n, void *stream);
/* Testing aid: explicit per-handle switches for code paths production sizes rarely reach. The library itself reads
 * no environment variable.
 *   "chunks" = k (1..16): cut batches of >= 4k molecules into k molecule chunks whose pre-pass runs on a side stream
 *              one chunk ahead (the loop that otherwise only runs beyond 65535 (molecule, channel chunk) pairs);
 *   "max_ct" = 1..32: upper bound on the channels one workgroup accumulates (more channel chunks);
 *   "direct" = 1 / 0 / -1: always / never / automatically take the single-launch per-molecule kernel (float32 grids);
 *   "mall_budget_kb" = k: cut batches into chunks of at most k KiB of pre-pass data (production: 288 MB, sized for
 *              the 256 MiB Infinity Cache), so that small test batches exercise the chunk-by-chunk launch order;
 *   "max_ct64" = 16 | 32: float64 grids: channels one workgroup accumulates (default 32: Gaussian grids of more than
 *              16 channels take 32 per workgroup on 4-wave slabs; 16 = the two-chunk form every other grid uses);
 *   "nw" = 1..16: waves (8-voxel z sub-tiles) per slab instead of the plan's (0 = the plan); measurement aid;
 *   "dense_grid": accepted and ignored (round 2's second voxelize launch no longer exists). */
int mvx_debug_set_option(mvx_handle *h, const char *name, int32_t value);

/*
 * How a call of a given shape is executed: the library's whole decision table (route, slab decomposition, channel and
 * molecule chunks, pacing, write-out path) as a pure host function - no handle, no device, nothing is launched. The forward
 * entry points take exactly these decisions (with the handle's debug options applied on top). Exposed so that the table can
 * be pinned by tests and read by callers who size their batches (the reference has no counterpart: it has one code path).
 */
enum mvx_route {
    MVX_ROUTE_BINNED = 0,    /* prep -> xbin -> voxelize_kernel (slab lines; the batched float32 pipeline) */
    MVX_ROUTE_DIRECT = 1,    /* voxelize_pair_kernel: the whole call in one launch */
    MVX_ROUTE_F64_DENSE = 2, /* float64 grids, general slab loop */
    MVX_ROUTE_F64_MX = 3     /* float64 grids, 32-channel chunks on the matrix cores */
};
typedef struct mvx_plan_query {
    int32_t dimension;
    int32_t blockdim;      /* <= 0: the reference default 8 */
    int32_t precision;     /* 32 (or 0) | 64 */
    int32_t mode;          /* 0 features, 1 types, 2 single */
    int32_t radii_type;    /* enum mvx_radii */
    int32_t B, C;          /* molecules, channels */
    int32_t out_aligned16; /* 1: the grid pointer is 16-byte aligned */
    int64_t total_atoms;   /* over the batch */
    int64_t max_atoms;     /* of one molecule */
} mvx_plan_query;
typedef struct mvx_plan {
    int32_t route;            /* enum mvx_route */
    int32_t nsx, nsy, nzc;    /* slabs along x, along y, z chunks of a row */
    int32_t nw;               /* waves per slab: a slab is 2 x 4 x (8 nw) voxels */
    int32_t ct, ncc;          /* channels per workgroup, channel chunks */
    int32_t nfull, ct_rem;    /* chunks of the main launch; width of the remainder launch's kernel (0: none) */
    int32_t nchunk;           /* molecule chunks (gridDim.y limit, Infinity Cache budget, "chunks" option) */
    int32_t pace;             /* 0 none, 1 empty slabs hold their zero fill back, 2 light slabs pace their rounds too */
    int32_t grouped;          /* channel-wise radii for features: the grouped matrix-core launch */
    int32_t lane_range;       /* sub-tiles straddle reference blocks: per-lane index ranges */
    int32_t vec_store;        /* 16-byte stores (rows are whole 16-byte quads and the grid is aligned) */
    int32_t xcd_ranges;       /* run-wise write-out with one contiguous slab range per XCD */
    int32_t cpad;             /* channel weights per atom the voxelize kernels read */
    int32_t weights_in_place; /* 1: the caller's feature rows are read in place (no packed copy) */
    int32_t reserved;
} mvx_plan;
int mvx_plan_call(const mvx_plan_query *query, mvx_plan *plan);
#ifdef __cplusplus
}
#endif
#endif /* MVX_H */
