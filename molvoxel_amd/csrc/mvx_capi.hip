// mvx_capi.hip — host side of libmvx_hip.so: the C ABI declared in include/mvx.h.
//
// Owns per-handle device workspace (atom records, ranges, metadata, staging for host-resident
// arguments) and a small ring of pinned host slots so that calls with device-resident data are
// fully asynchronous on the caller's stream (no hidden device synchronisation). There is no CPU
// fallback in this library: without a usable HIP device mvx_create fails.
#include "mvx_internal.h"
#include "mvx_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mvx;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return MVX_ERR_HIP;
}

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return fail_hip(_e, #expr); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct PinnedSlot {
    char *p = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};

constexpr int NSLOTS = 4;

} // namespace

// What the binned pipeline's pre-pass writes and its voxelize launches read. Two sets exist so that, with
// mvx_set_overlap, the pre-pass of call k+1 can fill one set on the side stream while call k's voxelize launches
// still read the other.
struct Workspace {
    DevBuf rec, wbuf, xp, xlist, slist, meta, aux;
    std::vector<char> meta_last; // host copy of the offsets the device meta buffer holds
    bool meta_valid = false;
    hipEvent_t ev_pre = nullptr; // pre-pass into this set finished (side stream)
    hipEvent_t ev_vox = nullptr; // the launches reading this set finished (caller's stream)
    bool vox_recorded = false;
};

struct mvx_handle {
    mvx_config cfg;
    Geom g;
    float sigma32;
    int device;
    Workspace ws[2];
    int cur = 0;         // the set the last binned call used
    bool overlap = false; // mvx_set_overlap
    DevBuf xf_buf, in_coords, in_chan, in_radii, out_stage, diag;
    PinnedSlot slots[NSLOTS];
    int next_slot = 0;
    std::vector<hipEvent_t> ev; // 2 * MVX_PROFILE_RING events, created on first use
    int ev_count = 0;           // timed launches recorded since the last read
    bool profiling = false;
    // test / measurement switches (mvx_debug_set_option): waves per slab, channels per workgroup, forced routes, molecule
    // chunks with the pre-pass on a side stream ("chunks": off by default - on cfg-2 x 64 the cross-stream waits and the
    // extra launch boundaries cost more than the 50 us of pre-pass they hide: 0.454 ms per step on one stream, 0.476 with 2
    // chunks, 0.513 with 4), Infinity Cache budget
    PlanKnobs knobs;
    // Stream hand-over: every call reuses the handle's workspace, ordered by the caller's stream. When a call arrives
    // on another stream than the previous one, the new stream first waits for everything the old stream holds
    // (event recorded on the old stream at that moment), so back-to-back calls on different streams never race.
    hipStream_t last_stream = nullptr;
    bool used = false;
    hipEvent_t ev_switch = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr;
    std::vector<hipEvent_t> ev_pre;
    int narrow_sub = 0; // "narrow_sub" option: sub-tiles per wave of narrow chunks (1: voxelize_kernel; 2 | 4: voxelize_narrow_kernel; 0: the rule)
    int dbg = 0; // diagnostic builds (-DMVX_DIAG) only
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

int ensure(DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return MVX_OK;
    if (b.p) HIP_TRY(hipFree(b.p)); // synchronises: nothing in flight may still use the old buffer
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        g_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return MVX_ERR_ALLOC;
    }
    b.cap = want;
    return MVX_OK;
}

int acquire_slot(mvx_handle *h, size_t bytes, PinnedSlot **out) {
    PinnedSlot &s = h->slots[h->next_slot];
    h->next_slot = (h->next_slot + 1) % NSLOTS;
    if (!s.done) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    if (s.in_flight) {
        HIP_TRY(hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    if (bytes > s.cap) {
        if (s.p) HIP_TRY(hipHostFree(s.p));
        s.p = nullptr;
        s.cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.p), want, hipHostMallocDefault));
        s.cap = want;
    }
    *out = &s;
    return MVX_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void make_geom(mvx_handle *h) {
    const mvx_config &c = h->cfg;
    Geom g;
    g.res = c.resolution;
    const double width = c.resolution * (double)(c.dimension - 1); // base/voxelizer.py:28
    g.half = width / 2.0;                                          // base/voxelizer.py:33
    g.D = c.dimension;
    g.bd = c.blockdim > 0 ? c.blockdim : 8;                        // numpy/voxelizer.py:38
    g.nb = (g.D + g.bd - 1) / g.bd;                                // numpy/voxelizer.py:44
    g.bd_inv = g.bd > 1 ? (uint32_t)((0x100000000ull + (uint64_t)g.bd - 1) / (uint64_t)g.bd) : 0u;
    g.inv_res = 1.0 / g.res;
    g.inv_pitch = 1.0 / ((double)g.bd * g.res);
    h->g = g;
    h->sigma32 = (float)c.sigma;
}

struct RunArgs {
    int mode;
    const double *coords;
    const void *channels; // float features (sumN, C) | int32 types (sumN) | null
    const void *radii; // float, or double for a precision-64 handle (like features and out)
    double radius_scalar;
    int radii_type;
    const int64_t *offsets;
    const mvx_xform *xforms;
    int B, C;
    void *out;
    int in_kind, out_kind;
    hipStream_t stream;
};

// Orders this call after the previous call's work when the caller switched streams (see mvx_handle::last_stream).
int adopt_stream(mvx_handle *h, hipStream_t s) {
    if (h->used && h->last_stream != s) {
        if (!h->ev_switch) HIP_TRY(hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming));
        // (the previous stream is alive by contract - mvx.h: "a stream must stay alive until the next call on the handle
        // has returned". If recording on it fails all the same, the error is reported once, the sticky HIP error is
        // cleared and the handle moves on to the new stream behind a device-wide synchronisation: one caller mistake
        // must not wedge the handle for good.)
        hipError_t e = hipEventRecord(h->ev_switch, h->last_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, h->ev_switch, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            h->last_stream = s;
            return fail_hip(e, "stream hand-over (was the previous call's stream destroyed?)");
        }
    }
    h->last_stream = s;
    h->used = true;
    return MVX_OK;
}

// ---- 1. what the library checks itself (shapes are the Python layer's job) -----------------------------------------
int validate(const mvx_handle *h, const RunArgs &r, int64_t &total, int64_t &max_atoms) {
    if (!h) return fail(MVX_ERR_INVALID, "null handle");
    if (r.B < 0 || r.C <= 0) return fail(MVX_ERR_INVALID, "B must be >= 0 and C > 0");
    if (r.B == 0) return MVX_OK;
    if (!r.offsets || !r.out) return fail(MVX_ERR_INVALID, "offsets/out must not be null");
    if (r.radii_type < MVX_RADII_SCALAR || r.radii_type > MVX_RADII_CHANNEL)
        return fail(MVX_ERR_INVALID, "bad radii_type");
    if (r.radii_type == MVX_RADII_CHANNEL && r.mode == MODE_SINGLE)
        return fail(MVX_ERR_INVALID, "Channel-Wise Radii Type is not supported"); // numpy/voxelizer.py:443
    if ((r.in_kind != MVX_HOST && r.in_kind != MVX_DEVICE) || (r.out_kind != MVX_HOST && r.out_kind != MVX_DEVICE))
        return fail(MVX_ERR_INVALID, "bad memory kind");
    if (r.offsets[0] != 0) return fail(MVX_ERR_INVALID, "offsets[0] must be 0");
    total = r.offsets[r.B];
    max_atoms = 0; // of one molecule
    for (int b = 0; b < r.B; ++b) {
        if (r.offsets[b + 1] < r.offsets[b]) return fail(MVX_ERR_INVALID, "offsets must be non-decreasing");
        max_atoms = std::max(max_atoms, r.offsets[b + 1] - r.offsets[b]);
    }
    if (total > 0 && !r.coords) return fail(MVX_ERR_INVALID, "coords must not be null");
    // (an empty molecule comes with empty, possibly null, per-atom arrays)
    if (!r.radii && (r.radii_type == MVX_RADII_CHANNEL || (r.radii_type == MVX_RADII_ATOM && total > 0)))
        return fail(MVX_ERR_INVALID, "radii array required");
    if (total > 0 && r.mode != MODE_SINGLE && !r.channels) return fail(MVX_ERR_INVALID, "channels must not be null");
    if (total >= (int64_t)1 << 31) return fail(MVX_ERR_INVALID, "too many atoms");
    return MVX_OK;
}

// ---- 2. inputs and metadata on the device ---------------------------------------------------------------------------
struct DeviceInputs {
    const int64_t *offsets = nullptr;
    const mvx_xform *xforms = nullptr;
    const double *coords = nullptr;
    const void *channels = nullptr;
    const void *radii = nullptr;
    PinnedSlot *slot = nullptr; // holds the host copies until `done` fires
};

// With host-resident inputs a centre given by pointer (MVX_XF_CENTER_PTR) is a host pointer: fold it into center[].
void resolve_host_centers(mvx_xform *xf, int n) {
    for (int i = 0; i < n; ++i) {
        if (xf[i].flags & MVX_XF_CENTER_PTR) {
            if (xf[i].center_ptr) std::memcpy(xf[i].center, xf[i].center_ptr, 3 * sizeof(double));
            xf[i].flags &= ~(uint32_t)MVX_XF_CENTER_PTR;
            xf[i].center_ptr = nullptr;
        }
    }
}

// Host-resident arrays go through one pinned slot (a single memcpy each, then async H2D on the caller's stream);
// device-resident arrays are used where they are. Offsets and transforms always come from the host.
int stage_inputs(mvx_handle *h, Workspace &w, const RunArgs &r, int64_t total, size_t esz, hipStream_t s, DeviceInputs &in) {
    const size_t off_bytes = align_up((size_t)(r.B + 1) * sizeof(int64_t), 16);
    const size_t xf_bytes = r.xforms ? align_up((size_t)r.B * sizeof(mvx_xform), 16) : 0;
    const bool host_in = (r.in_kind == MVX_HOST);
    const size_t chan_elem = (r.mode == MODE_FEATURES) ? (size_t)r.C * esz : (r.mode == MODE_TYPES ? sizeof(int32_t) : 0);
    size_t rad_count = 0;
    if (r.radii_type == MVX_RADII_ATOM) rad_count = (size_t)total;
    else if (r.radii_type == MVX_RADII_CHANNEL) rad_count = (size_t)r.C;
    const size_t co_bytes = host_in ? align_up((size_t)total * 3 * sizeof(double), 16) : 0;
    const size_t ch_bytes = host_in ? align_up((size_t)total * chan_elem, 16) : 0;
    const size_t ra_bytes = host_in ? align_up(rad_count * esz, 16) : 0;

    int rc;
    const void *meta_before = w.meta.p;
    if ((rc = ensure(w.meta, off_bytes + xf_bytes))) return rc;
    if (w.meta.p != meta_before) w.meta_valid = false;
    // offsets (+ transforms) go to the device only when they differ from what the last call left there
    // (same-shaped batches, the common case in a training loop, skip a 5 us copy kernel)
    const size_t meta_used = (size_t)(r.B + 1) * sizeof(int64_t);
    const bool meta_same = !r.xforms && w.meta_valid && w.meta_last.size() == meta_used &&
                           std::memcmp(w.meta_last.data(), r.offsets, meta_used) == 0;
    // a pinned slot (and the event that releases it: a barrier packet on the stream, ~6 us of idle GPU) only when
    // something is staged through it
    char *pin = nullptr;
    if (!meta_same || host_in) {
        if ((rc = acquire_slot(h, off_bytes + xf_bytes + co_bytes + ch_bytes + ra_bytes, &in.slot))) return rc;
        pin = in.slot->p;
    }
    if (!meta_same) {
        std::memcpy(pin, r.offsets, meta_used);
        if (r.xforms) {
            std::memcpy(pin + off_bytes, r.xforms, (size_t)r.B * sizeof(mvx_xform));
            if (host_in) resolve_host_centers(reinterpret_cast<mvx_xform *>(pin + off_bytes), r.B);
        }
        HIP_TRY(hipMemcpyAsync(w.meta.p, pin, off_bytes + xf_bytes, hipMemcpyHostToDevice, s));
        w.meta_last.assign(reinterpret_cast<const char *>(r.offsets), reinterpret_cast<const char *>(r.offsets) + meta_used);
        w.meta_valid = !r.xforms; // (a later call on another stream waits for this stream first: adopt_stream)
    }
    in.offsets = reinterpret_cast<const int64_t *>(w.meta.p);
    in.xforms = r.xforms ? reinterpret_cast<const mvx_xform *>((char *)w.meta.p + off_bytes) : nullptr;
    in.coords = r.coords;
    in.channels = r.channels;
    in.radii = r.radii;
    if (!host_in) return MVX_OK;

    auto upload = [&](DevBuf &dst, const void *src, size_t used, size_t padded, char *staging, const void *&dev) -> int {
        if (used == 0) return MVX_OK;
        if (int e = ensure(dst, padded)) return e;
        std::memcpy(staging, src, used);
        HIP_TRY(hipMemcpyAsync(dst.p, staging, padded, hipMemcpyHostToDevice, s));
        dev = dst.p;
        return MVX_OK;
    };
    char *q = pin + off_bytes + xf_bytes;
    const void *dev_coords = in.coords;
    if ((rc = upload(h->in_coords, r.coords, (size_t)total * 3 * sizeof(double), co_bytes, q, dev_coords))) return rc;
    in.coords = static_cast<const double *>(dev_coords);
    if ((rc = upload(h->in_chan, r.channels, (size_t)total * chan_elem, ch_bytes, q + co_bytes, in.channels))) return rc;
    if ((rc = upload(h->in_radii, r.radii, rad_count * esz, ra_bytes, q + co_bytes + ch_bytes, in.radii))) return rc;
    return MVX_OK;
}

// ---- 3. how the call is executed: plan_call (mvx_plan.hip) ---------------------------------------------------------
inline uint32_t umulhi_inverse(int d) { // n / d == __umulhi(n, inv) for the slab ids used (n * d < 2^32); d == 1 is special-cased by the kernel
    return d == 1 ? 0xffffffffu : (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d);
}

// ---- 4. launches bracketed by profiling events when mvx_set_profiling is on ----------------------------------------
template <typename Launch>
int timed_launch(mvx_handle *h, hipStream_t s, Launch &&launch) {
    const bool timed = h->profiling && h->ev_count < MVX_PROFILE_RING;
    if (timed) set_launch_events(h->ev[2 * h->ev_count], h->ev[2 * h->ev_count + 1]);
    const hipError_t e = launch();
    if (timed) {
        if (launch_events_pending()) { // nothing was launched (an empty job): keep the pair well defined
            set_launch_events(nullptr, nullptr);
            HIP_TRY(hipEventRecord(h->ev[2 * h->ev_count], s));
            HIP_TRY(hipEventRecord(h->ev[2 * h->ev_count + 1], s));
        }
        ++h->ev_count;
    }
    HIP_TRY(e);
    return MVX_OK;
}

int run(mvx_handle *h, const RunArgs &r) {
    int64_t total = 0, max_atoms = 0;
    int rc = validate(h, r, total, max_atoms);
    if (rc || r.B == 0) return rc;

    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    hipStream_t s = r.stream;
    if ((rc = adopt_stream(h, s))) return rc;
    const Geom &g = h->g;
    const int D = g.D;
    const bool f64 = (h->cfg.precision == 64);
    const size_t esz = f64 ? sizeof(double) : sizeof(float); // element size of features, radii and the grid
    const size_t out_bytes = (size_t)r.B * r.C * D * D * D * esz;
    void *d_out = r.out;
    if (r.out_kind == MVX_HOST) {
        if ((rc = ensure(h->out_stage, out_bytes))) return rc;
        d_out = h->out_stage.p;
    }

    // ---- how this call is executed: one pure function of its shape (mvx_plan.hip, pinned by tests/test_plan.py) --------
    mvx_plan_query q{};
    q.dimension = D;
    q.blockdim = g.bd;
    q.precision = f64 ? 64 : 32;
    q.mode = r.mode;
    q.radii_type = r.radii_type;
    q.B = r.B;
    q.C = r.C;
    q.out_aligned16 = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0 ? 1 : 0;
    q.total_atoms = total;
    q.max_atoms = max_atoms;
    const mvx_plan plan = plan_call(q, h->knobs);
    const bool direct = plan.route == MVX_ROUTE_DIRECT;
    const bool mx64 = plan.route == MVX_ROUTE_F64_MX;
    const int ct = plan.ct, ncc = plan.ncc, nchunk = plan.nchunk;
    const size_t per_molecule = (size_t)plan.nsx * plan.nsy * plan.nzc;

    const bool forced_pipeline = h->knobs.pipeline > 1 && r.B >= 4 * h->knobs.pipeline;
    // Cross-call overlap (mvx_set_overlap): this call's pre-pass fills the other workspace set on the side stream,
    // under the previous call's voxelize launches. Device-resident inputs and outputs only.
    // Batches only (mvx.h: "the batched three-launch path"): a per-molecule call hands over arrays its Python layer may
    // have converted on the caller's stream a moment ago, which the side stream would not wait for.
    const bool overlap = h->overlap && !direct && r.B > 1 && nchunk == 1 && !forced_pipeline && r.in_kind == MVX_DEVICE && r.out_kind == MVX_DEVICE;
    if (overlap) h->cur ^= 1;
    Workspace &w = h->ws[h->cur];
    // (the side stream only serves the "chunks" test option and mvx_set_overlap; chunks cut for the cache, or for the
    // gridDim.y limit, run their launches back to back on the caller's stream)
    const bool side_stream = forced_pipeline || overlap;
    hipStream_t pre = s;
    if (side_stream) {
        if (!h->side) {
            // lowest priority: the pre-pass should take the slots the voxelize launch leaves, not compete
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_TRY(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo));
        }
        if (!h->ev_in) HIP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
        pre = h->side;
    }
    if (overlap) {
        if (!w.ev_pre) HIP_TRY(hipEventCreateWithFlags(&w.ev_pre, hipEventDisableTiming));
        if (!w.ev_vox) HIP_TRY(hipEventCreateWithFlags(&w.ev_vox, hipEventDisableTiming));
        // this set was last read by the call before the previous one: its launches must have drained. (Nothing makes
        // the side stream wait for the caller's stream otherwise: that is the caller's promise about the inputs.)
        if (w.vox_recorded) {
            HIP_TRY(hipStreamWaitEvent(pre, w.ev_vox, 0));
        } else { // first use of the set: order behind everything the caller's stream holds so far
            HIP_TRY(hipEventRecord(h->ev_in, s));
            HIP_TRY(hipStreamWaitEvent(pre, h->ev_in, 0));
        }
    }

    DeviceInputs in;
    const bool by_value = (r.B == 1 && r.in_kind == MVX_DEVICE && nchunk == 1); // one molecule of device arrays
    if (by_value) { // nothing to stage: extent and transform travel with the launches
        in.coords = r.coords;
        in.channels = r.channels;
        in.radii = r.radii;
    } else if ((rc = stage_inputs(h, w, r, total, esz, overlap ? pre : s, in))) {
        return rc;
    }

    // ---- workspace --------------------------------------------------------------------------------
    const size_t n_alloc = (size_t)std::max<int64_t>(total, 1);
    const int Cpad = plan.cpad;
    const bool direct_w = plan.weights_in_place != 0;
    const size_t nslabs = (size_t)r.B * per_molecule;
    if (nslabs * (size_t)ncc + 1 > (size_t)0x7fffffff) return fail(MVX_ERR_INVALID, "batch too large for one call");
    if (!direct) {
        if ((rc = ensure(w.rec, n_alloc * sizeof(AtomRec)))) return rc;
        if (!direct_w && (rc = ensure(w.wbuf, n_alloc * (size_t)Cpad * esz))) return rc;
        if ((rc = ensure(w.xp, n_alloc * sizeof(uint2)))) return rc;
        // x-lists: packed regions, (sum(N) + 2*B) * nsx entries; slab lines: primary + extension entries per slab
        // (+ 64 B: a wave reads its eight entries of an overflowing slab's list as one group, up to seven past the list's end)
        if ((rc = ensure(w.xlist, ((size_t)total + 2 * (size_t)r.B) * plan.nsx * sizeof(uint2) + 64))) return rc;
        if ((rc = ensure(w.slist, nslabs * (SLAB_LINE_ENTRIES + SLAB_EXT_ENTRIES) * sizeof(uint2)))) return rc;
    }
    uint2 *d_xlist = reinterpret_cast<uint2 *>(w.xlist.p);
    uint2 *d_slist = reinterpret_cast<uint2 *>(w.slist.p);
    uint2 *d_slist_ext = d_slist ? d_slist + nslabs * SLAB_LINE_ENTRIES : nullptr; // extension lines live behind the primary lines

    const bool gauss = (h->cfg.density == MVX_GAUSSIAN);
    const bool chanwise = (r.radii_type == MVX_RADII_CHANNEL && r.mode == MODE_FEATURES);
    void *d_rmax = nullptr;
    double *d_Tc = nullptr;
    float *d_kc = nullptr;
    ChanGroups *d_groups = nullptr;
    int32_t *d_chan_slot = nullptr;
    if (chanwise) {
        if (f64) { // [max radius | per-channel thresholds | per-channel coefficients]
            const size_t tc_off = 16, kc_off = tc_off + align_up((size_t)r.C * sizeof(double), 16);
            if ((rc = ensure(w.aux, kc_off + (size_t)r.C * sizeof(double)))) return rc;
            d_rmax = w.aux.p;
            d_Tc = reinterpret_cast<double *>((char *)w.aux.p + tc_off);
            d_kc = reinterpret_cast<float *>((char *)w.aux.p + kc_off); // (double coefficients behind a float pointer)
            HIP_TRY(launch_chan_aux64(static_cast<const double *>(in.radii), r.C, h->cfg.density, h->cfg.sigma,
                                      static_cast<double *>(d_rmax), d_Tc, reinterpret_cast<double *>(d_kc), overlap ? pre : s));
        } else { // [max radius | one ChanGroups per chunk of 32 channels | the channels' radius slots (int32 x C)]
            const size_t slot_off = 16 + (size_t)ncc * sizeof(ChanGroups);
            if ((rc = ensure(w.aux, slot_off + (size_t)r.C * sizeof(int32_t)))) return rc;
            d_rmax = w.aux.p;
            d_groups = reinterpret_cast<ChanGroups *>((char *)w.aux.p + 16);
            d_chan_slot = reinterpret_cast<int32_t *>((char *)w.aux.p + slot_off);
            HIP_TRY(launch_chan_aux(static_cast<const float *>(in.radii), r.C, h->cfg.density, h->sigma32, static_cast<float *>(d_rmax),
                                    d_groups, d_chan_slot, overlap ? pre : s));
        }
    }

    // ---- kernel arguments -------------------------------------------------------------------------
    PrepArgs pa;
    pa.coords = in.coords;
    pa.radii = in.radii;
    pa.types = (r.mode == MODE_TYPES) ? reinterpret_cast<const int32_t *>(in.channels) : nullptr;
    pa.features = (r.mode == MODE_FEATURES) ? in.channels : nullptr;
    pa.mode = r.mode;
    pa.Cpad = Cpad;
    pa.offsets = in.offsets;
    pa.xforms = in.xforms;
    std::memset(&pa.xf_one, 0, sizeof(pa.xf_one));
    if (by_value && r.xforms) pa.xf_one = r.xforms[0];
    pa.chan_aux = d_rmax;
    pa.precision = f64 ? 64 : 32;
    pa.first = 0;
    pa.total = total;
    pa.B = r.B;
    pa.C = r.C;
    pa.radius_scalar = r.radius_scalar;
    pa.T_scalar = -1.0;
    pa.k_scalar = 0.0f;
    if (r.radii_type == MVX_RADII_SCALAR && !f64)
        scalar_radius_constants(r.radius_scalar, h->sigma32, h->cfg.density == MVX_GAUSSIAN, &pa.T_scalar, &pa.k_scalar);
    if (r.radii_type == MVX_RADII_SCALAR) pa.radii_src = RAD_SCALAR;
    else if (r.radii_type == MVX_RADII_ATOM) pa.radii_src = RAD_ATOM;
    else pa.radii_src = chanwise ? RAD_CHANNEL_FEATURES : RAD_CHANNEL_BY_TYPE;
    pa.density = h->cfg.density;
    pa.sigma32 = h->sigma32;
    pa.sigma64 = h->cfg.sigma;
    pa.g = g;
    pa.rec = reinterpret_cast<AtomRec *>(w.rec.p);
    pa.wbuf = direct_w ? nullptr : w.wbuf.p;
    pa.xp = reinterpret_cast<uint2 *>(w.xp.p);

    VoxArgs va;
    va.rec = reinterpret_cast<const unsigned *>(w.rec.p);
    va.w = reinterpret_cast<const unsigned *>(direct_w ? in.channels : w.wbuf.p);
    va.xlist = d_xlist;
    va.slist = d_slist;
    va.slist_ext = d_slist_ext;
    va.offsets = in.offsets;
    va.n_one = total;
    va.Tc = d_Tc;
    va.kc = d_kc;
    va.out = d_out;
    va.narrow_sub = h->narrow_sub;
    va.p.res = g.res;
    va.p.half = g.half;
    va.p.D = D;
    va.p.C = r.C;
    va.p.B = r.B;
    va.p.nsx = plan.nsx;
    va.p.nsy = plan.nsy;
    va.p.nzc = plan.nzc;
    va.p.ncc = ncc;
    va.p.nsy_inv = umulhi_inverse(plan.nsy);
    va.p.nzc_inv = umulhi_inverse(plan.nzc);
    va.p.nsx_inv = umulhi_inverse(plan.nsx);
    va.p.ncc_inv = umulhi_inverse(ncc);
    va.p.b0 = 0;
    va.p.c0 = 0;
    va.p.NW = plan.nw;
    va.p.nslab = (uint32_t)(plan.nsx * plan.nsy * plan.nzc);
    va.p.w_stride = plan.grouped ? r.C : Cpad;
    va.p.dcap = f64 ? (mx64 ? 0 : 64) : 0; // (float64: rows per round of the general slab loop, 0 selects the matrix-core kernel)
    va.p.vec_store = plan.vec_store;
    va.p.xcd_ranges = plan.xcd_ranges;
    va.p.pace = plan.pace;
    va.p.sigma = h->cfg.sigma;
#ifdef MVX_DIAG
    va.p.dbg = h->dbg;
#endif
    // float32 grids whose rows are not whole 16-byte quads need the run-wise write-out (store_runs): compiled into the
    // per-lane-range kernels, voxelize_runs_kernel (the matrix-core walk of 32-channel chunks) and voxelize_pair_runs_kernel
    // (the candidate-pair walk of narrow chunks), nowhere else
    const bool lr_blocks = plan.lane_range != 0;
    const bool runs = !f64 && !va.p.vec_store;
    const bool lane_range = lr_blocks || runs;
    auto lane_range_for = [&](int32_t) { return lr_blocks; }; // (run-wise write-out: voxelize_runs_kernel / voxelize_pair_runs_kernel)

    if (direct) {
        DirectArgs da;
        da.pa = pa;
        da.pa.rec = nullptr;
#ifdef MVX_DIAG // stamps of every workgroup (8 x 8 B each), read back with mvx_debug_read_records
        if ((rc = ensure(w.rec, nslabs * (size_t)ncc * 64))) return rc;
        da.pa.rec = reinterpret_cast<AtomRec *>(w.rec.p);
        HIP_TRY(hipMemsetAsync(w.rec.p, 0, nslabs * (size_t)ncc * 64, s)); // (slots filled by atomicMax need a zero start)
#endif
        da.pa.wbuf = nullptr;
        da.pa.xp = nullptr;
        da.pa.chan_aux = nullptr;
        da.N = total;
        if (r.B == 1) { // one molecule: extent and transform by value, no metadata on the device
            da.pa.offsets = nullptr;
            da.pa.xforms = nullptr;
            std::memset(&da.pa.xf_one, 0, sizeof(da.pa.xf_one));
            if (r.xforms) {
                da.pa.xf_one = r.xforms[0];
                if (r.in_kind == MVX_HOST) resolve_host_centers(&da.pa.xf_one, 1);
            }
        }
        if ((rc = timed_launch(h, s, [&] { return launch_voxelize_direct(da, va.p, max_atoms, static_cast<float *>(d_out), ct, gauss, lr_blocks, s); })))
            return rc;
        if (in.slot) {
            HIP_TRY(hipEventRecord(in.slot->done, s));
            in.slot->in_flight = true;
        }
        if (r.out_kind == MVX_HOST) {
            HIP_TRY(hipMemcpyAsync(r.out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        return MVX_OK;
    }

    // ---- launches -----------------------------------------------------------------------------------------------
    if (side_stream && !overlap) {
        while ((int)h->ev_pre.size() < nchunk) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->ev_pre.push_back(e);
        }
        // inputs (and the workspace, still read by the previous call's launches) are ready once `s` gets here
        HIP_TRY(hipEventRecord(h->ev_in, s));
        HIP_TRY(hipStreamWaitEvent(pre, h->ev_in, 0));
    }
    auto chunk_begin = [&](int k) { return (int)((int64_t)r.B * k / nchunk); };
    auto prepass = [&](int k) -> int {
        const int b0 = chunk_begin(k), b1 = chunk_begin(k + 1);
        pa.first = r.offsets[b0];
        pa.total = r.offsets[b1];
        HIP_TRY(launch_prep(pa, pre));
        HIP_TRY(launch_xbin(pa.xp, in.offsets, total, b0, b1 - b0, max_atoms, plan.nsx, plan.nsy, plan.nzc, plan.nw, d_xlist, d_slist,
                            d_slist_ext, pre));
        if (side_stream) HIP_TRY(hipEventRecord(overlap ? w.ev_pre : h->ev_pre[k], pre));
        return MVX_OK;
    };
    const bool interleave = !side_stream && !f64 && nchunk > 1; // chunk by chunk: pre-pass, then its voxelize launch
    for (int k = 0; k < nchunk && !interleave; ++k)
        if ((rc = prepass(k))) return rc;
    if (f64) { // float64 grids: one launch over the whole batch
        for (int k = 0; k < nchunk && side_stream; ++k) HIP_TRY(hipStreamWaitEvent(s, overlap ? w.ev_pre : h->ev_pre[k], 0));
        if ((rc = timed_launch(h, s, [&] { return launch_voxelize64(va, ct, gauss, chanwise, lr_blocks, s); }))) return rc;
    } else {
        for (int k = 0; k < nchunk; ++k) {
            const int b0 = chunk_begin(k), b1 = chunk_begin(k + 1);
            if (interleave && (rc = prepass(k))) return rc;
            if (side_stream) HIP_TRY(hipStreamWaitEvent(s, overlap ? w.ev_pre : h->ev_pre[k], 0));
            va.p.b0 = b0;
            if (plan.grouped) {
                // channels grouped by radius (chan_aux_kernel): chunks of 32 channels on the matrix-core path, one
                // threshold test and one density per radius slot and candidate, feature rows read in place
                VoxArgs vg = va;
                vg.Tc = reinterpret_cast<const double *>(d_groups);
                vg.kc = reinterpret_cast<const float *>(d_chan_slot);
                if ((rc = timed_launch(h, s, [&] { return launch_voxelize_grouped(vg, b1 - b0, gauss, lane_range, s); }))) return rc;
                continue;
            }
            va.p.ncc = plan.nfull;
            va.p.c0 = 0;
            // the bracket holds voxelize_kernel alone (what rocprofv3 reports under that name)
            if ((rc = timed_launch(h, s, [&] { return launch_voxelize(va, b1 - b0, ct, gauss, lane_range_for(ct), s); }))) return rc;
            if (plan.ct_rem) { // the remainder channels [nfull * ct, C) with a narrower kernel
                va.p.ncc = 1;
                va.p.c0 = plan.nfull * ct;
                if ((rc = timed_launch(h, s, [&] { return launch_voxelize(va, b1 - b0, plan.ct_rem, gauss, lane_range_for(plan.ct_rem), s); }))) return rc;
            }
        }
    }

    if (overlap) { // the next call but one refills this set
        HIP_TRY(hipEventRecord(w.ev_vox, s));
        w.vox_recorded = true;
    } else {
        w.vox_recorded = false; // (its reads are ordered on the caller's stream only)
    }
    if (in.slot) {
        HIP_TRY(hipEventRecord(in.slot->done, s));
        in.slot->in_flight = true;
    }

    if (r.out_kind == MVX_HOST) {
        HIP_TRY(hipMemcpyAsync(r.out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return MVX_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int mvx_version(void) { return MVX_VERSION; }

const char *mvx_last_error(void) { return g_err.c_str(); }

int mvx_device_count(int *count) {
    if (!count) return fail(MVX_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip(e, "hipGetDeviceCount");
    }
    *count = n;
    return MVX_OK;
}

int mvx_create(const mvx_config *cfg, mvx_handle **out) {
    if (!cfg || !out) return fail(MVX_ERR_INVALID, "null argument");
    *out = nullptr;
    if (!(cfg->resolution > 0.0)) return fail(MVX_ERR_INVALID, "resolution must be > 0");
    if (cfg->dimension < 1 || cfg->dimension > 1020) return fail(MVX_ERR_INVALID, "dimension must be in [1, 1020]");
    if (cfg->density != MVX_GAUSSIAN && cfg->density != MVX_BINARY) return fail(MVX_ERR_INVALID, "bad density");
    if (cfg->density == MVX_GAUSSIAN && !(cfg->sigma > 0.0)) return fail(MVX_ERR_INVALID, "sigma must be > 0");
    if (cfg->precision != 0 && cfg->precision != 32 && cfg->precision != 64)
        return fail(MVX_ERR_INVALID, "precision must be 32 or 64"); // numpy/voxelizer.py:33
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_err = std::string("no usable HIP device (libmvx_hip has no CPU fallback): ") +
                (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return MVX_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= n) return fail(MVX_ERR_INVALID, "device ordinal out of range");
    mvx_handle *h = new (std::nothrow) mvx_handle();
    if (!h) return fail(MVX_ERR_ALLOC, "out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device;
    make_geom(h);
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) {
        delete h;
        return fail_hip(guard.err, "hipSetDevice");
    }
    *out = h;
    return MVX_OK;
}

int mvx_destroy(mvx_handle *h) {
    if (!h) return MVX_OK;
    DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    std::vector<DevBuf *> bufs = {&h->xf_buf, &h->in_coords, &h->in_chan, &h->in_radii, &h->out_stage};
    for (Workspace &w : h->ws) {
        for (DevBuf *b : {&w.rec, &w.wbuf, &w.xp, &w.xlist, &w.slist, &w.meta, &w.aux}) bufs.push_back(b);
        if (w.ev_pre) (void)hipEventDestroy(w.ev_pre);
        if (w.ev_vox) (void)hipEventDestroy(w.ev_vox);
    }
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (PinnedSlot &s : h->slots) {
        if (s.p) (void)hipHostFree(s.p);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_pre) (void)hipEventDestroy(e);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_switch) (void)hipEventDestroy(h->ev_switch);
    if (h->side) (void)hipStreamDestroy(h->side);
    delete h;
    return MVX_OK;
}

int mvx_set_density(mvx_handle *h, int32_t density, double sigma) {
    if (!h) return fail(MVX_ERR_INVALID, "null handle");
    if (density != MVX_GAUSSIAN && density != MVX_BINARY) return fail(MVX_ERR_INVALID, "bad density");
    if (density == MVX_GAUSSIAN && !(sigma > 0.0)) return fail(MVX_ERR_INVALID, "sigma must be > 0");
    h->cfg.density = density;
    if (density == MVX_GAUSSIAN) {
        h->cfg.sigma = sigma;
        h->sigma32 = (float)sigma;
    }
    return MVX_OK;
}

int mvx_set_overlap(mvx_handle *h, int32_t enable) {
    if (!h) return fail(MVX_ERR_INVALID, "null handle");
    h->overlap = enable != 0;
    return MVX_OK;
}

int mvx_forward_features_batch(mvx_handle *h, const double *coords, const void *features, const void *radii,
                               double radius_scalar, int32_t radii_type, const int64_t *offsets,
                               const mvx_xform *xforms, int32_t B, int32_t C, void *out, int32_t in_kind,
                               int32_t out_kind, void *stream) {
    RunArgs r{MODE_FEATURES, coords, features, radii, radius_scalar, radii_type, offsets, xforms, B, C, out,
              in_kind, out_kind, reinterpret_cast<hipStream_t>(stream)};
    return run(h, r);
}

int mvx_forward_types_batch(mvx_handle *h, const double *coords, const int32_t *types, const void *radii,
                            double radius_scalar, int32_t radii_type, const int64_t *offsets,
                            const mvx_xform *xforms, int32_t B, int32_t C, void *out, int32_t in_kind,
                            int32_t out_kind, void *stream) {
    RunArgs r{MODE_TYPES, coords, types, radii, radius_scalar, radii_type, offsets, xforms, B, C, out,
              in_kind, out_kind, reinterpret_cast<hipStream_t>(stream)};
    return run(h, r);
}

int mvx_forward_single_batch(mvx_handle *h, const double *coords, const void *radii, double radius_scalar,
                             int32_t radii_type, const int64_t *offsets, const mvx_xform *xforms, int32_t B,
                             void *out, int32_t in_kind, int32_t out_kind, void *stream) {
    RunArgs r{MODE_SINGLE, coords, nullptr, radii, radius_scalar, radii_type, offsets, xforms, B, 1, out,
              in_kind, out_kind, reinterpret_cast<hipStream_t>(stream)};
    return run(h, r);
}

int mvx_forward_features(mvx_handle *h, const double *coords, const void *features, const void *radii,
                         double radius_scalar, int32_t radii_type, int64_t N, int32_t C, const mvx_xform *xform,
                         void *out, int32_t in_kind, int32_t out_kind, void *stream) {
    const int64_t off[2] = {0, N};
    return mvx_forward_features_batch(h, coords, features, radii, radius_scalar, radii_type, off, xform, 1, C, out,
                                      in_kind, out_kind, stream);
}

int mvx_forward_types(mvx_handle *h, const double *coords, const int32_t *types, const void *radii,
                      double radius_scalar, int32_t radii_type, int64_t N, int32_t C, const mvx_xform *xform,
                      void *out, int32_t in_kind, int32_t out_kind, void *stream) {
    const int64_t off[2] = {0, N};
    return mvx_forward_types_batch(h, coords, types, radii, radius_scalar, radii_type, off, xform, 1, C, out,
                                   in_kind, out_kind, stream);
}

int mvx_forward_single(mvx_handle *h, const double *coords, const void *radii, double radius_scalar,
                       int32_t radii_type, int64_t N, const mvx_xform *xform, void *out, int32_t in_kind,
                       int32_t out_kind, void *stream) {
    const int64_t off[2] = {0, N};
    return mvx_forward_single_batch(h, coords, radii, radius_scalar, radii_type, off, xform, 1, out, in_kind,
                                    out_kind, stream);
}

int mvx_transform_coords(mvx_handle *h, const double *coords, int64_t N, const mvx_xform *xform, double *out,
                         int32_t in_kind, int32_t out_kind, void *stream) {
    if (!h || !xform || (N > 0 && (!coords || !out))) return fail(MVX_ERR_INVALID, "null argument");
    if (N <= 0) return MVX_OK;
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (int rc0 = adopt_stream(h, s)) return rc0;
    const size_t xf_bytes = align_up(sizeof(mvx_xform), 16);
    const size_t co_bytes = (size_t)N * 3 * sizeof(double);
    const bool host_in = in_kind == MVX_HOST, host_out = out_kind == MVX_HOST;
    PinnedSlot *slot = nullptr;
    int rc = acquire_slot(h, xf_bytes + (host_in ? co_bytes : 0), &slot);
    if (rc) return rc;
    if ((rc = ensure(h->xf_buf, xf_bytes))) return rc;
    std::memcpy(slot->p, xform, sizeof(mvx_xform));
    if (host_in) resolve_host_centers(reinterpret_cast<mvx_xform *>(slot->p), 1);
    HIP_TRY(hipMemcpyAsync(h->xf_buf.p, slot->p, xf_bytes, hipMemcpyHostToDevice, s));
    const double *d_in = coords;
    if (host_in) {
        if ((rc = ensure(h->in_coords, co_bytes))) return rc;
        std::memcpy(slot->p + xf_bytes, coords, co_bytes);
        HIP_TRY(hipMemcpyAsync(h->in_coords.p, slot->p + xf_bytes, co_bytes, hipMemcpyHostToDevice, s));
        d_in = reinterpret_cast<const double *>(h->in_coords.p);
    }
    double *d_out = out;
    if (host_out) {
        if ((rc = ensure(h->out_stage, co_bytes))) return rc;
        d_out = reinterpret_cast<double *>(h->out_stage.p);
    }
    HIP_TRY(launch_transform(d_in, N, reinterpret_cast<const mvx_xform *>(h->xf_buf.p), d_out, s));
    HIP_TRY(hipEventRecord(slot->done, s));
    slot->in_flight = true;
    if (host_out) {
        HIP_TRY(hipMemcpyAsync(out, d_out, co_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return MVX_OK;
}

int mvx_set_profiling(mvx_handle *h, int32_t enable) {
    if (!h) return fail(MVX_ERR_INVALID, "null handle");
    DeviceGuard guard(h->device);
    if (enable && h->ev.empty()) {
        h->ev.reserve(2 * MVX_PROFILE_RING);
        for (int i = 0; i < 2 * MVX_PROFILE_RING; ++i) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            h->ev.push_back(e);
        }
    }
    h->profiling = enable != 0;
    h->ev_count = 0;
    return MVX_OK;
}

int mvx_profile_read(mvx_handle *h, float *ms, int32_t capacity, int32_t *count) {
    if (!h || !count || (capacity > 0 && !ms)) return fail(MVX_ERR_INVALID, "null argument");
    DeviceGuard guard(h->device);
    const int n = std::min<int>(h->ev_count, capacity);
    if (h->ev_count > 0) HIP_TRY(hipEventSynchronize(h->ev[2 * h->ev_count - 1]));
    for (int i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], h->ev[2 * i], h->ev[2 * i + 1]));
    *count = n;
    h->ev_count = 0;
    return MVX_OK;
}

int mvx_last_kernel_ms(mvx_handle *h, float *ms) {
    if (!h || !ms) return fail(MVX_ERR_INVALID, "null argument");
    if (h->ev_count <= 0) return fail(MVX_ERR_INVALID, "no timed launch (call mvx_set_profiling(h, 1) first)");
    DeviceGuard guard(h->device);
    const int i = h->ev_count - 1;
    HIP_TRY(hipEventSynchronize(h->ev[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(ms, h->ev[2 * i], h->ev[2 * i + 1]));
    return MVX_OK;
}

int mvx_debug_read_records(mvx_handle *h, void *host_dst, int64_t n, void *stream) {
    if (!h || !host_dst || n < 0) return fail(MVX_ERR_INVALID, "bad argument");
    const DevBuf &rec = h->ws[h->cur].rec;
    if ((size_t)n * sizeof(AtomRec) > rec.cap) return fail(MVX_ERR_INVALID, "more records requested than the workspace holds");
    if (n == 0) return MVX_OK;
    DeviceGuard guard(h->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(host_dst, rec.p, (size_t)n * sizeof(AtomRec), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return MVX_OK;
}

int mvx_debug_set_option(mvx_handle *h, const char *name, int32_t value) {
    if (!h || !name) return fail(MVX_ERR_INVALID, "null argument");
    const std::string n(name);
    PlanKnobs &k = h->knobs;
    if (n == "chunks") k.pipeline = std::max(1, std::min(16, (int)value));
    else if (n == "max_ct") k.max_ct = std::max(1, std::min(32, (int)value));
    else if (n == "direct") k.direct_mode = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "max_ct64") k.max_ct64 = value >= 32 ? 32 : 16;
    else if (n == "narrow_sub") h->narrow_sub = (value == 1 || value == 2 || value == 4) ? value : 0;
    else if (n == "dense_grid") (void)value; // (accepted and ignored: there is no second voxelize launch any more)
    else if (n == "nw") k.force_nw = (value >= 1 && value <= 16) ? value : 0; // waves (8-voxel z sub-tiles) per slab; 0 = the default plan
    else if (n == "mall_budget_kb") k.mall_budget = value > 0 ? 1024.0 * (double)value : MALL_BUDGET;
#ifdef MVX_DIAG
    else if (n == "dbg") h->dbg = value;
    else if (n == "vk_stamps") { // value = workgroups to make room for (0: off); read back with mvx_debug_read_diag
        DeviceGuard guard(h->device);
        if (value > 0) {
            if (int rc = ensure(h->diag, (size_t)value * 128)) return rc;
            HIP_TRY(hipMemset(h->diag.p, 0, (size_t)value * 128));
        }
        HIP_TRY(set_diag_buffer(value > 0 ? h->diag.p : nullptr));
    }
    else if (n == "xb_stamps") { // the same for xbin_kernel: value = blocks, 64 B each
        DeviceGuard guard(h->device);
        if (value > 0) {
            if (int rc = ensure(h->diag, (size_t)value * 64)) return rc;
            HIP_TRY(hipMemset(h->diag.p, 0, (size_t)value * 64));
        }
        HIP_TRY(set_diag_buffer_xb(value > 0 ? h->diag.p : nullptr));
    }
#endif
    else return fail(MVX_ERR_INVALID, "unknown option: " + n);
    return MVX_OK;
}

#ifdef MVX_DIAG
extern "C" int mvx_debug_read_diag(mvx_handle *h, void *host_dst, int64_t bytes) {
    if (!h || !host_dst || bytes < 0 || (size_t)bytes > h->diag.cap) return fail(MVX_ERR_INVALID, "bad argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_dst, h->diag.p, (size_t)bytes, hipMemcpyDeviceToHost));
    return MVX_OK;
}
#endif

int mvx_alloc(mvx_handle *h, int64_t bytes, void **ptr) {
    if (!h || !ptr || bytes < 0) return fail(MVX_ERR_INVALID, "bad argument");
    DeviceGuard guard(h->device);
    *ptr = nullptr;
    hipError_t e = hipMalloc(ptr, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) {
        g_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return MVX_ERR_ALLOC;
    }
    return MVX_OK;
}

int mvx_free(mvx_handle *h, void *ptr) {
    if (!h) return fail(MVX_ERR_INVALID, "null handle");
    if (!ptr) return MVX_OK;
    DeviceGuard guard(h->device);
    HIP_TRY(hipFree(ptr));
    return MVX_OK;
}

int mvx_memcpy(mvx_handle *h, void *dst, const void *src, int64_t bytes, int32_t dst_kind, int32_t src_kind,
               void *stream) {
    if (!h || bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(MVX_ERR_INVALID, "bad argument");
    if (bytes == 0) return MVX_OK;
    DeviceGuard guard(h->device);
    hipMemcpyKind kind;
    if (dst_kind == MVX_DEVICE && src_kind == MVX_HOST) kind = hipMemcpyHostToDevice;
    else if (dst_kind == MVX_HOST && src_kind == MVX_DEVICE) kind = hipMemcpyDeviceToHost;
    else if (dst_kind == MVX_DEVICE && src_kind == MVX_DEVICE) kind = hipMemcpyDeviceToDevice;
    else kind = hipMemcpyHostToHost;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, kind, s));
    HIP_TRY(hipStreamSynchronize(s)); // host memory may be pageable: return only when the copy is done
    return MVX_OK;
}

int mvx_memset_zero(mvx_handle *h, void *ptr, int64_t bytes, void *stream) {
    if (!h || bytes < 0 || (bytes > 0 && !ptr)) return fail(MVX_ERR_INVALID, "bad argument");
    if (bytes == 0) return MVX_OK;
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemsetAsync(ptr, 0, (size_t)bytes, reinterpret_cast<hipStream_t>(stream)));
    return MVX_OK;
}

int mvx_stream_sync(mvx_handle *h, void *stream) {
    if (!h) return fail(MVX_ERR_INVALID, "null handle");
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return MVX_OK;
}

} // extern "C"
