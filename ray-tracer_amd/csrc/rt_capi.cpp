// rt_capi.cpp -- the C ABI declared in include/rt_mi355x.h.
//
// Host side of the library: owns the device copies of the scene, the work-queue
// head, the ray counter and the scratch frame buffers, validates arguments and
// enqueues the kernels of rt_kernels.hip.  There is no CPU implementation of
// the path behind this boundary: without a HIP device every call fails.
#include "rt_bvh.h"
#include "rt_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace rtamd;

static_assert(kMaxJobs == RT_MAX_JOBS, "the kernels' job table and the header's limit");
constexpr size_t kCounterBytes = (size_t)kCounterSlots * kCounterWords * sizeof(unsigned long long);

// Geometry lists and LDS tiles are padded to whole candidate groups of the matrix scan.
static constexpr uint32_t kPad = 128;

struct rt_context {
    int device = 0;
    int n_cus = 0;
    hipStream_t stream = nullptr; // used by the synchronous entry points

    Geom *d_geom_exact = nullptr;  // {centre, fl(r*r)}
    Geom *d_geom_filter = nullptr; // {centre, r*r*(1 + 2^-16)}: vector-ALU conservative filter
    Geom *d_geom_matrix = nullptr; // {centre - shift, K}: matrix-core conservative filter
    float shift[3] = { 0.f, 0.f, 0.f };
    rt_sphere *d_spheres = nullptr;
    uint32_t n_spheres = 0;
    bool have_scene = false;
    bool scene_tame = false;       // every sphere within the conservative filters' preconditions
    int filter = kFilterMatrix;
    bool filter_forced = false;    // set by RT_OPT_FILTER: no automatic choice by scene size

    // host copies kept for the hierarchy, which is built on first use
    std::vector<rt_sphere> h_spheres;
    std::vector<Geom> h_exact;
    BvhHost bvh;                   // host copy of what d_bvh_* hold
    bool bvh_ready = false;
    int bvh_leaf_max = 0;          // RT_OPT_BVH_LEAF_SPHERES; 0 = chosen by trial (build_hierarchy())
    uint32_t bvh_suspend_lanes = 16;  // see RT_OPT_BVH_SUSPEND_LANES
    uint32_t bvh_lds_nodes_max = 512; // trees up to this many nodes (64 B each) are walked in LDS
    uint32_t subtiles_max = 8;     // work queue: tiles may be cut into this many strips of rows (1, 2, 4 or 8)
    uint32_t chunks_per_wave = 32; // work queue: split tiles' samples until there are this many chunks per resident wave
    uint32_t direct_add_samples = 1; // RT_OPT_DIRECT_ADD_SAMPLES
    uint32_t max_chunks = 0x3fffffffu; // RT_OPT_MAX_CHUNKS_PER_LAUNCH: beyond it a call's sample range goes in several launches
    bool count_work = false;       // RT_OPT_COUNT_WORK: launch the counting build of the trace kernel
    uint32_t bvh_auto_min = 13;    // automatic choice: hierarchy from this many spheres up (measured: it beats the
                                   // vector scan from 16 spheres; lists of up to 12 have a kernel of their own)
    float4 *d_bvh_nodes = nullptr;
    Geom *d_bvh_geom = nullptr;
    uint32_t *d_bvh_index = nullptr;

    uint32_t *d_chunk_counter = nullptr;
    // This device's copy of the library's static scratch array (sums, then queue heads), asked of
    // the runtime once, in rt_create, with the device current; used by rt_render_device_static.
    void *static_scratch = nullptr;
    // Trace launches of one context share the work queue's heads, so each waits for the one before
    // it, whatever streams the caller passes: ev_trace is recorded behind every launch.
    hipEvent_t ev_trace = nullptr;
    bool trace_enqueued = false;
    unsigned long long *d_rays = nullptr;

    uint32_t *d_sums = nullptr; // scratch for rt_render*/rt_render_device
    size_t sums_bytes = 0;
    uint8_t *d_rgb = nullptr;   // scratch for rt_render
    size_t rgb_bytes = 0;
    uint8_t *h_rows = nullptr;  // pinned host staging for regions narrower than the frame
    size_t rows_bytes = 0;
    // rt_render_progressive: a second frame, a copy stream and "frame resolved" events, so that the
    // next pass is traced while the last one's frame travels to the host and the callback runs
    uint8_t *d_rgb2 = nullptr;  // the frames of a group of passes after the first
    size_t rgb2_bytes = 0;
    uint32_t *d_sums2 = nullptr; // and their sums
    size_t sums2_bytes = 0;
    uint32_t progressive_group = 0; // RT_OPT_PROGRESSIVE_PASSES_PER_LAUNCH; 0 = by the passes' size
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_resolved[2] = { nullptr, nullptr };
    void *d_io_in = nullptr;    // scratch for rt_intersect's rays
    size_t io_in_bytes = 0;
    void *d_io_out = nullptr;   // scratch for rt_intersect's records and rt_primary_hits' map
    size_t io_out_bytes = 0;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr; // device time of the synchronous calls' kernels
    // resident workgroups per CU of the trace kernel, by (filter, LDS bytes): asked of the runtime once
    struct OccEntry { int filter; bool counting; size_t lds; int per_cu; };
    std::vector<OccEntry> occupancy_cache;

    uint32_t tile_cap_max = 512;  // spheres per LDS tile (2 x 16 B each: 16 KiB; 4 workgroups per CU)
    int wg_per_cu_override = 0;

    double last_kernel_ms = 0.0, last_wall_ms = 0.0;
    std::string error;
    // The reference calls its seam from four threads at once (TraceScene, Source/main.cpp:647-659):
    // entry points on one context take this lock, so such calls are safe and run one after another.
    std::recursive_mutex mtx;
};

namespace {

int fail(rt_context *ctx, int status, const std::string &msg)
{
    if (ctx)
        ctx->error = msg;
    return status;
}

int fail_hip(rt_context *ctx, hipError_t e, const char *what)
{
    return fail(ctx, RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define RT_HIP(ctx, call)                                  \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess)                              \
            return fail_hip((ctx), e_, #call);             \
    } while (0)

// Makes the context's device current for the duration of an entry point and gives the caller's
// device back on every way out, so that a host driving several contexts (or its own HIP code)
// never finds another device selected behind its back.
class DeviceScope {
public:
    explicit DeviceScope(int device)
    {
        have_prev_ = hipGetDevice(&prev_) == hipSuccess;
        status_ = (have_prev_ && prev_ == device) ? hipSuccess : hipSetDevice(device);
        switched_ = status_ == hipSuccess && (!have_prev_ || prev_ != device);
    }
    ~DeviceScope()
    {
        if (switched_ && have_prev_)
            (void)hipSetDevice(prev_);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
    hipError_t status() const { return status_; }

private:
    int prev_ = 0;
    bool have_prev_ = false, switched_ = false;
    hipError_t status_ = hipSuccess;
};

#define RT_DEVICE(ctx)                          \
    DeviceScope device_scope_((ctx)->device);   \
    RT_HIP((ctx), device_scope_.status())

struct Region {
    int x0, y0, w, h;
};

// The device's counters, summed over their kCounterSlots copies (the caller has synchronised).
int read_counters(rt_context *ctx, unsigned long long out[kCounterWords])
{
    std::vector<unsigned long long> all((size_t)kCounterSlots * kCounterWords);
    RT_HIP(ctx, hipMemcpy(all.data(), ctx->d_rays, kCounterBytes, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < kCounterWords; ++i) {
        out[i] = 0;
        for (uint32_t s = 0; s < kCounterSlots; ++s)
            out[i] += all[(size_t)s * kCounterWords + i];
    }
    return RT_OK;
}

// Preconditions of the conservative filters' error bounds and of the hierarchy builder: every
// magnitude finite and below 2^28, r*r at least 2^-40.  Written as comparisons that are false
// for NaN (fmax would drop a NaN operand).
bool below_limit(float v) { return std::fabs(v) < 0x1p28f; }

bool sphere_tame(const rt_sphere &sp, float rr)
{
    return below_limit(sp.center[0]) && below_limit(sp.center[1]) && below_limit(sp.center[2]) &&
           below_limit(sp.radius) && rr >= 0x1p-40f;
}

// Clips TraceThreadFunction's tile to the frame: pixels outside it are dropped
// by PutPixel in the reference (Source/main.cpp:275) and are not traced here.
Region clip_region(const rt_params &p, int xs, int ys, int xn, int yn)
{
    long x0 = xs, y0 = ys, x1 = (long)xs + xn, y1 = (long)ys + yn;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > (long)p.width) x1 = p.width;
    if (y1 > (long)p.height) y1 = p.height;
    Region r;
    r.x0 = (int)x0; r.y0 = (int)y0;
    r.w = x1 > x0 ? (int)(x1 - x0) : 0;
    r.h = y1 > y0 ? (int)(y1 - y0) : 0;
    return r;
}

int check_params(rt_context *ctx, const rt_camera *cam, const rt_params *p)
{
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!cam || !p) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null camera or params");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_set_scene has not been called");
    if (p->width == 0 || p->height == 0 || p->spp == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "width, height and spp must be positive");
    if ((uint64_t)p->width * p->height > (1ull << 30))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "frame larger than 2^30 pixels");
    // per-pixel sums are uint32 and a sample adds up to 255 per channel (Source/main.cpp:633-635)
    if (p->spp > 0xFFFFFFFFu / 255u)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "spp above 16843009: the 32-bit colour sums could overflow");
    if (p->max_depth > 4096)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "max_depth above 4096");
    if (p->sample_count && (uint64_t)p->sample_begin + p->sample_count > p->spp)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "sample range exceeds spp");
    if (!p->sample_count && p->sample_begin != 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "sample_begin without sample_count");
    if ((p->flags & ~(uint32_t)(RT_FLAG_DIELECTRIC | RT_FLAG_GAMMA2 | RT_FLAG_FLOAT_SHADING)) != 0u || p->reserved != 0u)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown flags");
    return RT_OK;
}

int ensure(rt_context *ctx, void **ptr, size_t *have, size_t want)
{
    if (*have >= want)
        return RT_OK;
    if (*ptr) {
        RT_HIP(ctx, hipFree(*ptr));
        *ptr = nullptr;
        *have = 0;
    }
    hipError_t e = hipMalloc(ptr, want);
    if (e != hipSuccess)
        return fail(ctx, RT_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    *have = want;
    return RT_OK;
}

void drop_bvh(rt_context *ctx)
{
    if (ctx->d_bvh_nodes) (void)hipFree(ctx->d_bvh_nodes);
    if (ctx->d_bvh_geom) (void)hipFree(ctx->d_bvh_geom);
    if (ctx->d_bvh_index) (void)hipFree(ctx->d_bvh_index);
    ctx->d_bvh_nodes = nullptr;
    ctx->d_bvh_geom = nullptr;
    ctx->d_bvh_index = nullptr;
    ctx->bvh = BvhHost();
    ctx->bvh_ready = false;
}

// Builds the hierarchy with the leaf size the options ask for, or (RT_OPT_BVH_LEAF_SPHERES 0) chosen by
// trial.  A tree walked in LDS costs LDS, and with it resident waves: leaves of up to 4 are best while
// the node copy leaves room for seven workgroups per CU (the 500-sphere frame: 162 nodes), but 1 000
// spheres in leaves of 4 are 309 nodes and five workgroups, 1 500 are 496 nodes and three.  So: the
// smallest of 4, 8, 12, 15 whose tree still leaves seven workgroups per CU (at eight bounces), else the
// fewest nodes LDS may hold (1080p / 32 spp, ms: 700 spheres 4.00 -> 3.59, 1 000 4.62 -> 3.80, 1 500
// 5.89 -> 4.03, 2 000 4.92 -> 4.24).  Trees that cannot be walked in LDS at all (more than
// kBvhLdsRecords spheres, or more nodes than RT_OPT_BVH_LDS_NODES at every size) get 8: there a node
// visit costs more than a few more exact tests (10 000 spheres: 4 % faster than 4).
void build_hierarchy(const rt_context *ctx, BvhHost &b)
{
    const rt_sphere *sp = ctx->h_spheres.data();
    const Geom *ex = ctx->h_exact.data();
    if (ctx->bvh_leaf_max > 0) {
        bvh_build(sp, ex, ctx->n_spheres, ctx->bvh_leaf_max, b);
        return;
    }
    if (ctx->bvh_lds_nodes_max > 0u && ctx->n_spheres <= kBvhLdsRecords) {
        bool have = false;
        for (int leaf : { 4, 8, 12, 15 }) {
            BvhHost t;
            bvh_build(sp, ex, ctx->n_spheres, leaf, t);
            const uint32_t nodes = (uint32_t)t.nodes.size(), levels = t.levels ? t.levels : 1u;
            if (nodes > ctx->bvh_lds_nodes_max)
                continue; // would be walked through the caches
            const size_t lds = trace_lds_bytes(kFilterBvh, 0u, trace_aux_bytes(kFilterBvh, levels, nodes), 8, 2u);
            b = std::move(t);
            have = true;
            if (lds * 7u <= 160u * 1024u)
                return;
        }
        if (have)
            return; // the largest leaves that fit: the fewest nodes
    }
    bvh_build(sp, ex, ctx->n_spheres, 8, b);
}

// Builds the hierarchy over the current scene and copies it to the device (once per scene
// and leaf size).
// `prebuilt` (may be null): the hierarchy another context built over the same list with the same leaf size.
int ensure_bvh(rt_context *ctx, const BvhHost *prebuilt = nullptr)
{
    if (ctx->bvh_ready)
        return RT_OK;
    drop_bvh(ctx);
    BvhHost &b = ctx->bvh;
    if (prebuilt)
        b = *prebuilt;
    else
        build_hierarchy(ctx, b);
    if (b.levels > 96u)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "hierarchy deeper than 96 levels");
    // the kernels follow child references and leaf ranges without bounds checks: make sure on the
    // host that every one of them is in range before anything is launched
    const std::string why = bvh_validate(ctx->h_spheres.data(), ctx->n_spheres, b);
    if (!why.empty())
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "hierarchy failed its structural check: " + why);
    const size_t nb = std::max<size_t>(b.nodes.size(), 1) * sizeof(BvhNode);
    // three records of padding: the walk reads a leaf's records four at a time
    const size_t gb = (b.leaf_geom.size() + 3u) * sizeof(Geom);
    const size_t ib = std::max<size_t>(b.leaf_index.size(), 1) * sizeof(uint32_t);
    if (hipMalloc((void **)&ctx->d_bvh_nodes, nb) != hipSuccess || hipMalloc((void **)&ctx->d_bvh_geom, gb) != hipSuccess ||
        hipMalloc((void **)&ctx->d_bvh_index, ib) != hipSuccess) {
        drop_bvh(ctx);
        return fail(ctx, RT_ERR_OUT_OF_MEMORY, "hipMalloc failed for the hierarchy");
    }
    if (!b.nodes.empty())
        RT_HIP(ctx, hipMemcpy(ctx->d_bvh_nodes, b.nodes.data(), b.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemset(ctx->d_bvh_geom, 0, gb));
    if (!b.leaf_geom.empty()) {
        RT_HIP(ctx, hipMemcpy(ctx->d_bvh_geom, b.leaf_geom.data(), b.leaf_geom.size() * sizeof(Geom), hipMemcpyHostToDevice));
        RT_HIP(ctx, hipMemcpy(ctx->d_bvh_index, b.leaf_index.data(), b.leaf_index.size() * sizeof(uint32_t),
                              hipMemcpyHostToDevice));
    }
    ctx->bvh_ready = true;
    return RT_OK;
}

// The scan the current options select for this scene.
int chosen_filter(const rt_context *ctx)
{
    int f = ctx->scene_tame ? ctx->filter : (int)kFilterExact;
    if (f == kFilterMatrix && !ctx->filter_forced) {
        // the matrix scan works on groups of 128 spheres; a handful of spheres is cheaper on the
        // vector ALU, and from a few dozen up a walk over boxes beats both (measured crossover)
        if (ctx->n_spheres < 96u)
            f = kFilterConservative;
        if (ctx->n_spheres >= ctx->bvh_auto_min)
            f = kFilterBvh;
    }
    return f;
}

// Builds the hierarchy if need be and fills the kernels' view of it.
int bvh_args(rt_context *ctx, BvhArgs &out, bool may_build = true)
{
    if (!ctx->bvh_ready && !may_build)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT,
                    "the hierarchy is not on the device yet and this entry point may not allocate it: "
                    "set the options before rt_set_scene, or render once through another entry point");
    const int st = ensure_bvh(ctx);
    if (st != RT_OK)
        return st;
    const BvhHost &b = ctx->bvh;
    out.nodes = ctx->d_bvh_nodes;
    out.leaf_geom = ctx->d_bvh_geom;
    out.leaf_index = ctx->d_bvh_index;
    out.n_nodes = (uint32_t)b.nodes.size();
    out.n_large = b.n_large;
    out.levels = b.levels ? b.levels : 1u;
    out.lds_nodes = (out.n_nodes <= ctx->bvh_lds_nodes_max && out.n_nodes < 32768u && b.leaf_geom.size() <= kBvhLdsRecords)
                        ? out.n_nodes : 0u;
    out.suspend_lanes = ctx->bvh_suspend_lanes;
    memcpy(out.center, b.center, sizeof out.center);
    out.radius2 = std::nextafter(b.radius * b.radius, INFINITY);
    out.rmin = b.rmin;
    out.rmin2w = b.rmin2w;
    return RT_OK;
}

// One job of a trace launch as the entry points hand it over: already checked (check_params), region clipped.
struct JobSpec {
    const rt_camera *cam;
    rt_params p;
    Region rg;
    uint32_t *d_sums;
};

// Enqueues ONE trace launch for `n` jobs (the sample loops of n frames, sample ranges or tile sets,
// each into its own sums; same seed and flags).  `queue_heads` = the work queue to use (null: the
// context's own); with `may_allocate` false nothing is allocated on the way (the hierarchy must be on
// the device already).
// `ordered` false (a stream under graph capture): the launch is not tied to the context's previous
// trace launch by an event -- the caller's graph orders it.
int enqueue_jobs(rt_context *ctx, const JobSpec *specs, uint32_t n, hipStream_t stream, uint32_t *queue_heads = nullptr,
                 bool may_allocate = true, bool ordered = true)
{
    TraceArgs a;
    memset(&a, 0, sizeof a);
    // ---- the jobs that have something to trace
    for (uint32_t i = 0; i < n; ++i) {
        const JobSpec &js = specs[i];
        if (js.rg.w == 0 || js.rg.h == 0)
            continue;
        if (js.p.max_depth <= 0) // GetRayColor(ray, 0) is RGB(0,0,0) (Source/main.cpp:555-558): nothing to add
            continue;
        TraceJob &j = a.jobs[a.n_jobs];
        j.tiles_x = (uint32_t)(js.rg.w + kTileW - 1) / kTileW;
        j.tiles_y = (uint32_t)(js.rg.h + kTileH - 1) / kTileH;
        const uint32_t n_tiles = j.tiles_x * j.tiles_y;
        j.tile_stride = js.p.tile_stride ? js.p.tile_stride : 1u;
        j.tile_first = js.p.tile_first;
        if (j.tile_stride == 1u && j.tile_first != 0u)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "tile_first without tile_stride");
        if (j.tile_first >= j.tile_stride && j.tile_stride > 1u)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "tile_first must be below tile_stride");
        j.n_chunks = j.tile_first < n_tiles ? (n_tiles - j.tile_first + j.tile_stride - 1u) / j.tile_stride : 0u;
        if (j.n_chunks == 0u)
            continue;
        if (a.n_jobs == 0u) {
            a.flags = js.p.flags;
            a.key0 = (uint32_t)js.p.seed;
            a.key1 = (uint32_t)(js.p.seed >> 32);
        } else if (a.flags != js.p.flags || a.key0 != (uint32_t)js.p.seed || a.key1 != (uint32_t)(js.p.seed >> 32)) {
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "the jobs of one launch must share seed and flags");
        }
        j.cam = *js.cam;
        j.width = js.p.width;
        j.height = js.p.height;
        j.width_d = (double)(float)js.p.width;
        j.height_d = (double)(float)js.p.height;
        j.inv_width = 1.0 / j.width_d;
        j.inv_height = 1.0 / j.height_d;
        j.sample_begin = js.p.sample_begin;
        j.sample_count = js.p.sample_count ? js.p.sample_count : js.p.spp;
        j.max_depth = js.p.max_depth;
        j.x0 = js.rg.x0; j.y0 = js.rg.y0; j.w = js.rg.w; j.h = js.rg.h;
        j.sums = js.d_sums;
        a.max_depth = std::max(a.max_depth, j.max_depth);
        ++a.n_jobs;
    }
    if (a.n_jobs == 0u)
        return RT_OK;

    // ---- what the jobs share: the scene, the scan, the kernel's shape
    a.filter = chosen_filter(ctx);
    if (a.filter == kFilterBvh) {
        const int st = bvh_args(ctx, a.bvh, may_allocate);
        if (st != RT_OK)
            return st;
    }
    a.geom = a.filter == kFilterMatrix ? ctx->d_geom_matrix : ctx->d_geom_filter;
    a.exact = ctx->d_geom_exact;
    memcpy(a.shift, ctx->shift, sizeof a.shift);
    a.spheres = ctx->d_spheres;
    a.n = ctx->n_spheres;
    a.count_work = ctx->count_work ? 1u : 0u;
    a.rays = ctx->d_rays;
    a.chunk_counter = queue_heads ? queue_heads : ctx->d_chunk_counter;

    // LDS tile: the whole (padded) list when it fits, else tiles of tile_cap_max
    const uint32_t n_pad = (a.n + (kPad - 1)) & ~(uint32_t)(kPad - 1);
    a.tile_cap = n_pad <= ctx->tile_cap_max ? (n_pad ? n_pad : kPad) : ctx->tile_cap_max;
    if (a.filter == kFilterConservative && a.n <= kShortList)
        a.filter = kFilterShort;
    if (a.filter == kFilterBvh || a.filter == kFilterShort)
        a.tile_cap = 0u; // the walk reads nodes and leaf records from memory, the short list is read with scalar loads
    a.aux_bytes = trace_aux_bytes(a.filter, a.bvh.levels, a.bvh.lds_nodes);

    a.stack_entry_bytes = a.n <= 65536u ? 2u : 4u;
    size_t lds = trace_lds_bytes(a.filter, a.tile_cap, a.aux_bytes, a.max_depth, a.stack_entry_bytes);
    if (lds > 64u * 1024u && a.bvh.lds_nodes != 0u) {
        // a copy of the nodes this large would leave room for one or two workgroups per CU:
        // walk them through the caches instead
        a.bvh.lds_nodes = 0u;
        a.aux_bytes = trace_aux_bytes(a.filter, a.bvh.levels, 0u);
        lds = trace_lds_bytes(a.filter, a.tile_cap, a.aux_bytes, a.max_depth, a.stack_entry_bytes);
    }
    if (lds > 160u * 1024u)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "max_depth too large: the bounce stack does not fit in LDS");
    int per_cu = 0;
    for (const rt_context::OccEntry &e : ctx->occupancy_cache)
        if (e.filter == a.filter && e.counting == (a.count_work != 0u) && e.lds == lds)
            per_cu = e.per_cu;
    if (per_cu == 0) {
        RT_HIP(ctx, trace_occupancy(a.filter, a.count_work != 0u, lds, &per_cu));
        if (ctx->occupancy_cache.size() < 64u)
            ctx->occupancy_cache.push_back({ a.filter, a.count_work != 0u, lds, per_cu });
    }
    if (per_cu < 1)
        return fail(ctx, RT_ERR_HIP, "trace kernel does not fit on a CU");
    if (ctx->wg_per_cu_override > 0 && ctx->wg_per_cu_override < per_cu)
        per_cu = ctx->wg_per_cu_override;
    // every workgroup's 4 waves pull chunks until the queue is empty, so the grid
    // only needs to fill the chip
    uint32_t n_wg = (uint32_t)ctx->n_cus * (uint32_t)per_cu;

    // ---- the work queue: every job's tiles cut into chunks
    // A chunk is owned by one wave.  A job with few tiles per resident wave (a small frame, or
    // one rank's share of a frame) would leave waves idle or the last ones running alone, so its
    // tiles are cut into smaller chunks of at least 256 paths: first into 2 or 4 strips of rows
    // (disjoint pixels, plain stores), then the sample range into segments (atomic adds).
    uint64_t total_chunks = 0u;
    for (uint32_t i = 0; i < a.n_jobs; ++i) {
        TraceJob &j = a.jobs[i];
        j.subtiles = 1u;
        j.segments = 1u;
        j.seg_samples = j.sample_count;
        // (the jobs of a launch follow each other through the queue: each is cut as if it were alone, so
        // that the waves find the same granularity at every job's end)
        const uint64_t wanted = (uint64_t)ctx->chunks_per_wave * 4u * n_wg;
        uint64_t per_tile = (wanted + j.n_chunks - 1u) / j.n_chunks; // chunks wanted of each tile
        // Strips of rows first, and as many as leave a chunk kMinChunkPaths paths: strips own their pixels
        // and flush with plain stores whatever their number (a tile's pixels are written once either way),
        // and the finer the chunks, the more evenly a launch ends -- also where tiles abound (4K at 64 spp:
        // 24.8 ms with two strips per tile, 24.1 with four).  Sample segments, which share pixels and flush
        // with atomics (one counted write per addition), only for what is still missing.
        constexpr uint32_t kMinChunkPaths = 512u;
        if (ctx->chunks_per_wave != 0u)
            while (j.subtiles * 2u <= ctx->subtiles_max && (uint64_t)(64u / (j.subtiles * 2u)) * j.sample_count >= kMinChunkPaths)
                j.subtiles *= 2u;
        // (small calls may go below that, down to 256 paths a chunk, to have enough chunks at all)
        if (per_tile >= 2u && ctx->subtiles_max >= 2u && j.sample_count >= 8u && j.subtiles < 2u)
            j.subtiles = 2u;
        if (per_tile >= 3u && ctx->subtiles_max >= 4u && j.sample_count >= 16u && j.subtiles < 4u)
            j.subtiles = 4u;
        if (per_tile >= 5u && ctx->subtiles_max >= 8u && j.sample_count >= 32u && j.subtiles < 8u)
            j.subtiles = 8u;
        per_tile = (per_tile + j.subtiles - 1u) / j.subtiles;
        const uint32_t min_segment = 4u * j.subtiles; // 256 paths on 64, 32, 16 or 8 pixels
        uint32_t per = j.sample_count;
        if (per_tile >= 2u && j.sample_count >= 2u * min_segment) {
            per = (uint32_t)((j.sample_count + per_tile - 1u) / per_tile);
            if (per < min_segment)
                per = min_segment;
        }
        // a wave sums a chunk's samples of a pixel in 16 bits per channel: at most 256 samples of 255
        if (per > kSegmentSamplesMax)
            per = kSegmentSamplesMax;
        if (per < j.sample_count) {
            j.segments = (j.sample_count + per - 1u) / per; // every segment non-empty
            j.seg_samples = per;
        }
        uint64_t total = (uint64_t)j.n_chunks * j.subtiles * j.segments;
        if (total > 0x3fffffffull && j.subtiles > 1u) {
            j.subtiles = 1u;
            total = (uint64_t)j.n_chunks * j.segments;
        }
        if (total_chunks + total > 0x7fffffffull)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "too many tiles x samples for one launch's work queue");
        j.chunk_first = (uint32_t)total_chunks;
        j.n_chunks = (uint32_t)total;
        total_chunks += total;
        // plain stores only where this launch has a single writer per pixel: samples in one segment, and no
        // other job adding into the same memory
        j.atomic_flush = j.segments > 1u ? 1u : 0u;
        const uintptr_t lo = (uintptr_t)j.sums, hi = lo + (size_t)j.width * j.height * 3u * sizeof(uint32_t);
        for (uint32_t o = 0; o < a.n_jobs; ++o) {
            const TraceJob &other = a.jobs[o];
            const uintptr_t olo = (uintptr_t)other.sums, ohi = olo + (size_t)other.width * other.height * 3u * sizeof(uint32_t);
            if (o != i && lo < ohi && olo < hi)
                j.atomic_flush = 1u;
        }
    }
    a.n_chunks = (uint32_t)total_chunks;
    a.direct_add = (a.n_jobs == 1u && a.jobs[0].seg_samples <= ctx->direct_add_samples) ? 1u : 0u;
    const uint32_t by_chunks = (a.n_chunks + 3u) / 4u;
    if (n_wg > by_chunks)
        n_wg = by_chunks;
    a.n_queues = 1u;
    while (a.n_queues * 2u <= kQueues && a.n_queues * 2u <= n_wg)
        a.n_queues *= 2u;

    if (ordered && ctx->trace_enqueued)
        RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_trace, 0));
    if (ordered)
        RT_HIP(ctx, hipMemsetAsync(a.chunk_counter, 0, kQueues * kQueueStride * sizeof(uint32_t), stream));
    else // under graph capture: a kernel, which every replay runs
        RT_HIP(ctx, launch_zero(a.chunk_counter, kQueues * kQueueStride * sizeof(uint32_t), stream));
    RT_HIP(ctx, launch_trace(a, n_wg, stream));
    if (ordered) {
        RT_HIP(ctx, hipEventRecord(ctx->ev_trace, stream));
        ctx->trace_enqueued = true;
    }
    return RT_OK;
}

// Enqueues the sample loop for `region` into d_sums: one job, or, where the job's tiles x sample
// segments outnumber what one launch's work queue can count, its sample range in several launches.
int enqueue_accumulate(rt_context *ctx, const rt_camera *cam, const rt_params *p, const Region &rg,
                       uint32_t *d_sums, hipStream_t stream, uint32_t *queue_heads = nullptr,
                       bool may_allocate = true, bool ordered = true)
{
    JobSpec js = { cam, *p, rg, d_sums };
    const uint32_t first = p->sample_begin, count = p->sample_count ? p->sample_count : p->spp;
    // at most max_chunks (2^30 - 1) chunks per launch, counted at the coarsest cut: whole tiles, 256 samples per chunk
    const uint64_t tiles = (uint64_t)((rg.w + kTileW - 1) / kTileW) * (uint64_t)((rg.h + kTileH - 1) / kTileH);
    const uint64_t segs = (count + kSegmentSamplesMax - 1u) / kSegmentSamplesMax;
    if (tiles * segs <= ctx->max_chunks)
        return enqueue_jobs(ctx, &js, 1u, stream, queue_heads, may_allocate, ordered);
    const uint64_t segs_per_launch = std::max<uint64_t>(ctx->max_chunks / std::max<uint64_t>(tiles, 1u), 1u);
    const uint64_t step = segs_per_launch * kSegmentSamplesMax;
    for (uint64_t done = 0u; done < count; done += step) {
        js.p.sample_begin = first + (uint32_t)done;
        js.p.sample_count = (uint32_t)std::min<uint64_t>(step, count - done);
        const int st = enqueue_jobs(ctx, &js, 1u, stream, queue_heads, may_allocate, ordered);
        if (st != RT_OK)
            return st;
    }
    return RT_OK;
}

// rt_render_device_static's scratch belongs to the GPU, not to a context: calls on one GPU are put in
// order whatever context and stream they come through, by an event per device that every such call
// waits for and records again.
struct StaticScratchOrder {
    std::mutex mtx;
    hipEvent_t done = nullptr;
    bool recorded = false;
};
constexpr int kMaxDevices = 64;
StaticScratchOrder g_static_order[kMaxDevices];

} // namespace

// ---------------------------------------------------------------------------

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

const char *rt_status_string(int status)
{
    switch (status) {
    case RT_OK: return "ok";
    case RT_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RT_ERR_HIP: return "HIP runtime error";
    case RT_ERR_NO_DEVICE: return "no usable HIP device";
    case RT_ERR_NO_SCENE: return "no scene set";
    case RT_ERR_OUT_OF_MEMORY: return "out of device memory";
    default: return "unknown status";
    }
}

const char *rt_last_error(const rt_context *ctx) { return ctx ? ctx->error.c_str() : ""; }

int rt_create(int device_id, rt_context **out)
{
    if (!out)
        return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return RT_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count)
        return RT_ERR_INVALID_ARGUMENT;
    rt_context *ctx = new (std::nothrow) rt_context;
    if (!ctx)
        return RT_ERR_OUT_OF_MEMORY;
    ctx->device = device_id;
    hipDeviceProp_t prop;
    DeviceScope scope(device_id);
    if (scope.status() != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        delete ctx;
        return RT_ERR_NO_DEVICE;
    }
    ctx->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_trace, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&ctx->ev_t0) != hipSuccess || hipEventCreate(&ctx->ev_t1) != hipSuccess ||
        hipMalloc((void **)&ctx->d_chunk_counter, kQueues * kQueueStride * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_rays, kCounterBytes) != hipSuccess ||
        static_scratch_address(&ctx->static_scratch) != hipSuccess || ctx->static_scratch == nullptr ||
        hipMemset(ctx->d_rays, 0, kCounterBytes) != hipSuccess) {
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    *out = ctx;
    return RT_OK;
}

int rt_destroy(rt_context *ctx)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    DeviceScope scope(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->d_geom_exact) (void)hipFree(ctx->d_geom_exact);
    if (ctx->d_geom_filter) (void)hipFree(ctx->d_geom_filter);
    if (ctx->d_geom_matrix) (void)hipFree(ctx->d_geom_matrix);
    if (ctx->d_spheres) (void)hipFree(ctx->d_spheres);
    drop_bvh(ctx);
    if (ctx->d_chunk_counter) (void)hipFree(ctx->d_chunk_counter);
    if (ctx->d_rays) (void)hipFree(ctx->d_rays);
    if (ctx->d_sums) (void)hipFree(ctx->d_sums);
    if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
    if (ctx->h_rows) (void)hipHostFree(ctx->h_rows);
    if (ctx->d_rgb2) (void)hipFree(ctx->d_rgb2);
    if (ctx->d_sums2) (void)hipFree(ctx->d_sums2);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : ctx->ev_resolved)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_io_in) (void)hipFree(ctx->d_io_in);
    if (ctx->d_io_out) (void)hipFree(ctx->d_io_out);
    if (ctx->ev_trace) (void)hipEventDestroy(ctx->ev_trace);
    if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
    if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

int rt_set_scene(rt_context *ctx, const rt_sphere *spheres, uint32_t n)
{
    return rtamd::set_scene_sharing_hierarchy(ctx, spheres, n, nullptr);
}

} // extern "C"

int rtamd::set_scene_sharing_hierarchy(rt_context *ctx, const rt_sphere *spheres, uint32_t n, rt_context *donor)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (n && !spheres)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null sphere list");
    if (n > (1u << 24))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "more than 2^24 spheres");
    RT_DEVICE(ctx);
    RT_HIP(ctx, hipDeviceSynchronize());
    if (ctx->d_geom_exact) { RT_HIP(ctx, hipFree(ctx->d_geom_exact)); ctx->d_geom_exact = nullptr; }
    if (ctx->d_geom_filter) { RT_HIP(ctx, hipFree(ctx->d_geom_filter)); ctx->d_geom_filter = nullptr; }
    if (ctx->d_geom_matrix) { RT_HIP(ctx, hipFree(ctx->d_geom_matrix)); ctx->d_geom_matrix = nullptr; }
    if (ctx->d_spheres) { RT_HIP(ctx, hipFree(ctx->d_spheres)); ctx->d_spheres = nullptr; }
    ctx->have_scene = false;
    drop_bvh(ctx);

    const uint32_t n_pad = (n + (kPad - 1)) & ~(uint32_t)(kPad - 1);
    const size_t cnt = n_pad ? n_pad : kPad;
    std::vector<Geom> exact(cnt), filt(cnt), matr(cnt);
    memset(exact.data(), 0, cnt * sizeof(Geom));
    memset(filt.data(), 0, cnt * sizeof(Geom));

    // origin of the matrix filter's coordinates: per-axis median of the centres
    // (robust against a huge, far-away ground sphere)
    float shift[3] = { 0.f, 0.f, 0.f };
    if (n) {
        std::vector<float> axis(n);
        for (int k = 0; k < 3; ++k) {
            for (uint32_t i = 0; i < n; ++i) axis[i] = spheres[i].center[k];
            std::nth_element(axis.begin(), axis.begin() + n / 2, axis.end());
            shift[k] = std::isfinite(axis[n / 2]) ? axis[n / 2] : 0.f;
        }
    }

    bool tame = true;
    for (size_t i = 0; i < cnt; ++i) {
        if (i >= n) { // padding: never a candidate (K = 1e30 outweighs d^2 for every tame ray)
            matr[i].cx = matr[i].cy = matr[i].cz = 0.f;
            matr[i].r2 = 1e30f;
            continue;
        }
        const rt_sphere &sp = spheres[i];
        // sphere.Radius * sphere.Radius, Source/main.cpp:478: one rounded product
        volatile float rr = sp.radius * sp.radius;
        exact[i].cx = filt[i].cx = sp.center[0];
        exact[i].cy = filt[i].cy = sp.center[1];
        exact[i].cz = filt[i].cz = sp.center[2];
        exact[i].r2 = rr;
        // widened radius of the conservative filters, rounded up
        volatile float wide = rr + rr * 0x1p-16f;
        const float r2w = std::nextafter((float)wide, INFINITY);
        filt[i].r2 = r2w;
        // matrix filter record: c = fl(C - shift), K = |c|^2 (1 - 2^-18) - r2w (double, rounded once)
        volatile float cx = sp.center[0] - shift[0], cy = sp.center[1] - shift[1], cz = sp.center[2] - shift[2];
        const double cc = (double)cx * cx + (double)cy * cy + (double)cz * cz;
        matr[i].cx = cx; matr[i].cy = cy; matr[i].cz = cz;
        matr[i].r2 = (float)(cc * (1.0 - 0x1p-18) - (double)r2w);
        // preconditions of the filters' error bounds: magnitudes below 2^28 (also after the
        // shift), r*r at least 2^-40
        if (!sphere_tame(sp, rr) || !below_limit(cx) || !below_limit(cy) || !below_limit(cz))
            tame = false;
    }
    if (hipMalloc((void **)&ctx->d_geom_exact, cnt * sizeof(Geom)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_geom_filter, cnt * sizeof(Geom)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_geom_matrix, cnt * sizeof(Geom)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_spheres, (size_t)(n ? n : 1) * sizeof(rt_sphere)) != hipSuccess)
        return fail(ctx, RT_ERR_OUT_OF_MEMORY, "hipMalloc failed for the scene");
    RT_HIP(ctx, hipMemcpy(ctx->d_geom_exact, exact.data(), cnt * sizeof(Geom), hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_geom_filter, filt.data(), cnt * sizeof(Geom), hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(ctx->d_geom_matrix, matr.data(), cnt * sizeof(Geom), hipMemcpyHostToDevice));
    if (n)
        RT_HIP(ctx, hipMemcpy(ctx->d_spheres, spheres, (size_t)n * sizeof(rt_sphere), hipMemcpyHostToDevice));
    memcpy(ctx->shift, shift, sizeof shift);
    ctx->h_spheres.assign(spheres, spheres + n);
    ctx->h_exact.assign(exact.begin(), exact.begin() + n);
    ctx->scene_tame = tame;
    ctx->n_spheres = n;
    ctx->have_scene = true;
    // the hierarchy, if the current options select it, goes to the device with the scene: built here, or
    // taken over from a context that built it for the same list under the same options
    if (chosen_filter(ctx) == kFilterBvh) {
        const BvhHost *prebuilt = nullptr;
        if (donor && donor != ctx && donor->bvh_ready && donor->n_spheres == n &&
            donor->bvh_leaf_max == ctx->bvh_leaf_max && donor->bvh_lds_nodes_max == ctx->bvh_lds_nodes_max &&
            (n == 0u || memcmp(donor->h_spheres.data(), spheres, (size_t)n * sizeof(rt_sphere)) == 0))
            prebuilt = &donor->bvh;
        return ensure_bvh(ctx, prebuilt);
    }
    return RT_OK;
}

extern "C" {

// Camera::Camera, Source/main.cpp:420-438, in glm 0.9.8.5's operation order:
// normalize(v) = v * (1 / sqrt(dot(v, v))), scalar * vector per component.
int rt_camera_init(const float lookfrom[3], const float lookat[3], const float up[3],
                   float fov_degrees, rt_camera *cam)
{
    if (!lookfrom || !lookat || !up || !cam)
        return RT_ERR_INVALID_ARGUMENT;
    auto dot = [](const float *a, const float *b) {
        const float x = a[0] * b[0], y = a[1] * b[1], z = a[2] * b[2];
        return (x + y) + z;
    };
    auto normalize = [&](const float *v, float *o) {
        const float inv = 1.0f / std::sqrt(dot(v, v));
        o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
    };
    auto cross = [](const float *x, const float *y, float *o) {
        o[0] = x[1] * y[2] - y[1] * x[2];
        o[1] = x[2] * y[0] - y[2] * x[0];
        o[2] = x[0] * y[1] - y[0] * x[1];
    };
    const float aspect = 16.0f / 9.0f;
    const float theta = fov_degrees * 0.01745329251994329576923690768489f;
    const float half_h = std::tan(theta / 2.0f);
    const float vh = 2.0f * half_h;
    const float vw = vh * aspect;
    const float back[3] = { lookfrom[0] - lookat[0], lookfrom[1] - lookat[1], lookfrom[2] - lookat[2] };
    float w[3], c[3], u[3], v[3];
    normalize(back, w);
    cross(up, w, c);
    normalize(c, u);
    cross(w, u, v);
    for (int i = 0; i < 3; ++i) {
        cam->origin[i] = lookfrom[i];
        cam->horizontal[i] = vw * u[i];
        cam->vertical[i] = vh * v[i];
        const float a = cam->origin[i] - (cam->horizontal[i] / 2.0f);
        const float b = a - (cam->vertical[i] / 2.0f);
        cam->bottom_left[i] = b - w[i];
    }
    return RT_OK;
}

int rt_accumulate_device(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                         int ystart, int xsize, int ysize, uint32_t *d_sums, void *stream)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    int st = check_params(ctx, cam, params);
    if (st != RT_OK)
        return st;
    if (!d_sums)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null d_sums");
    RT_DEVICE(ctx);
    const Region rg = clip_region(*params, xstart, ystart, xsize, ysize);
    return enqueue_accumulate(ctx, cam, params, rg, d_sums, (hipStream_t)stream);
}

int rt_accumulate_batch_device(rt_context *ctx, const rt_job *jobs, uint32_t n_jobs, void *stream)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (n_jobs == 0u)
        return RT_OK;
    if (!jobs)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null jobs");
    if (n_jobs > RT_MAX_JOBS)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "more than RT_MAX_JOBS jobs in one launch");
    JobSpec specs[RT_MAX_JOBS];
    for (uint32_t i = 0; i < n_jobs; ++i) {
        const int st = check_params(ctx, &jobs[i].cam, &jobs[i].params);
        if (st != RT_OK)
            return st;
        if (!jobs[i].d_sums)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null d_sums");
        specs[i].cam = &jobs[i].cam;
        specs[i].p = jobs[i].params;
        specs[i].rg = clip_region(jobs[i].params, jobs[i].xstart, jobs[i].ystart, jobs[i].xsize, jobs[i].ysize);
        specs[i].d_sums = jobs[i].d_sums;
    }
    RT_DEVICE(ctx);
    return enqueue_jobs(ctx, specs, n_jobs, (hipStream_t)stream);
}

int rt_resolve_device(rt_context *ctx, const uint32_t *d_sums, uint32_t width, uint32_t height,
                      uint32_t spp_div, uint32_t flags, int xstart, int ystart, int xsize, int ysize,
                      uint8_t *d_rgb, void *stream)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!d_sums || !d_rgb || width == 0 || height == 0 || spp_div == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad resolve arguments");
    if ((uint64_t)width * height > (1ull << 30))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "frame larger than 2^30 pixels");
    if ((flags & ~(uint32_t)RT_FLAG_GAMMA2) != 0u)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_resolve_device takes RT_FLAG_GAMMA2 only");
    RT_DEVICE(ctx);
    rt_params p;
    memset(&p, 0, sizeof p);
    p.width = width; p.height = height;
    const Region rg = clip_region(p, xstart, ystart, xsize, ysize);
    ResolveArgs a;
    a.sums = d_sums; a.rgb = d_rgb;
    a.width = width; a.height = height; a.spp_div = spp_div; a.flags = flags;
    a.x0 = rg.x0; a.y0 = rg.y0; a.w = rg.w; a.h = rg.h;
    RT_HIP(ctx, launch_resolve(a, (hipStream_t)stream));
    return RT_OK;
}

// rt_render_device and rt_render_device_static: the same sequence, the sums and the queue heads
// either in the context's own buffers (grown as needed) or in the library's static array.
static int render_device(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                         int ystart, int xsize, int ysize, uint8_t *d_rgb, void *stream, bool use_static)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    int st = check_params(ctx, cam, params);
    if (st != RT_OK)
        return st;
    if (!d_rgb)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null d_rgb");
    RT_DEVICE(ctx);
    hipStream_t s = (hipStream_t)stream;
    uint32_t *sums = nullptr, *queue_heads = nullptr;
    if (use_static) {
        if ((uint64_t)params->width * params->height > kStaticMaxPixels)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT,
                        "frame larger than rt_render_device_static's scratch (RT_STATIC_MAX_PIXELS = 2097152 pixels)");
        sums = (uint32_t *)ctx->static_scratch;
        queue_heads = (uint32_t *)((unsigned char *)ctx->static_scratch + kStaticSumsBytes);
    } else {
        const size_t sums_bytes = (size_t)params->width * params->height * 3u * sizeof(uint32_t);
        st = ensure(ctx, (void **)&ctx->d_sums, &ctx->sums_bytes, sums_bytes);
        if (st != RT_OK)
            return st;
        sums = ctx->d_sums;
    }
    const Region rg = clip_region(*params, xstart, ystart, xsize, ysize);
    if (rg.w == 0 || rg.h == 0)
        return RT_OK;
    // A stream under graph capture takes no part in the event ordering below: an event recorded outside
    // the capture cannot be waited for inside it, and one recorded inside it means nothing to later
    // calls.  The captured sequence (memset, memset, trace, resolve) is ordered by the graph itself.
    bool capturing = false;
    if (use_static && s != nullptr) {
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        RT_HIP(ctx, hipStreamIsCapturing(s, &status));
        capturing = status != hipStreamCaptureStatusNone;
    }
    // Zero the rows of the region, trace, divide.  The sums are the context's own scratch and
    // calls may come on different streams (the reference's four strips share every row), so the
    // whole sequence is ordered behind the previous user of the scratch: ev_trace is waited for
    // before the memset and recorded again behind the resolve.  The static scratch is shared by all
    // contexts of the GPU: its calls are ordered per device as well.
    StaticScratchOrder *order = (use_static && !capturing && ctx->device < kMaxDevices) ? &g_static_order[ctx->device] : nullptr;
    std::unique_lock<std::mutex> order_lock;
    if (order) {
        order_lock = std::unique_lock<std::mutex>(order->mtx);
        if (!order->done)
            RT_HIP(ctx, hipEventCreateWithFlags(&order->done, hipEventDisableTiming));
        if (order->recorded)
            RT_HIP(ctx, hipStreamWaitEvent(s, order->done, 0));
    }
    if (!capturing && ctx->trace_enqueued)
        RT_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_trace, 0));
    const size_t row_bytes = (size_t)params->width * 3u * sizeof(uint32_t);
    if (capturing) {
        // rows are 12 bytes per pixel: the zeroed span is widened to 16-byte bounds inside the scratch
        // array (whose sums are all dead between calls)
        const size_t first = ((size_t)rg.y0 * row_bytes) & ~(size_t)15u;
        const size_t last = (((size_t)(rg.y0 + rg.h) * row_bytes) + 15u) & ~(size_t)15u;
        RT_HIP(ctx, launch_zero((uint8_t *)sums + first, last - first, s));
    } else {
        RT_HIP(ctx, hipMemsetAsync((uint8_t *)sums + (size_t)rg.y0 * row_bytes, 0, (size_t)rg.h * row_bytes, s));
    }
    st = enqueue_accumulate(ctx, cam, params, rg, sums, s, queue_heads, !use_static, !capturing);
    if (st != RT_OK)
        return st;
    ResolveArgs a;
    a.sums = sums; a.rgb = d_rgb;
    a.width = params->width; a.height = params->height; a.spp_div = params->spp; a.flags = params->flags;
    a.x0 = rg.x0; a.y0 = rg.y0; a.w = rg.w; a.h = rg.h;
    RT_HIP(ctx, launch_resolve(a, s));
    if (!capturing) {
        RT_HIP(ctx, hipEventRecord(ctx->ev_trace, s));
        ctx->trace_enqueued = true;
    }
    if (order) {
        RT_HIP(ctx, hipEventRecord(order->done, s));
        order->recorded = true;
    }
    return RT_OK;
}

// Copies region `rg` of the device frame into the host frame of the same layout, behind the work
// enqueued on `stream`, and waits for it.  A pitched copy to pageable memory is carried out row by
// row (8.6 ms for a 1080p frame against 0.6 for one run of bytes), so rows that span the frame's width
// go as one contiguous copy, and a narrower region's rows are fetched whole into pinned staging
// memory and their part of interest copied out by the host.
static int copy_region_to_host(rt_context *ctx, uint8_t *rgb, const uint8_t *d_rgb, uint32_t width, const Region &rg,
                               hipStream_t stream)
{
    const size_t pitch = (size_t)width * 3u;
    const size_t row0 = (size_t)rg.y0 * pitch;
    if ((uint32_t)rg.w == width) {
        RT_HIP(ctx, hipMemcpyAsync(rgb + row0, d_rgb + row0, (size_t)rg.h * pitch, hipMemcpyDeviceToHost, stream));
        RT_HIP(ctx, hipStreamSynchronize(stream));
        return RT_OK;
    }
    const size_t want = (size_t)rg.h * pitch;
    if (ctx->rows_bytes < want) {
        if (ctx->h_rows)
            RT_HIP(ctx, hipHostFree(ctx->h_rows));
        ctx->h_rows = nullptr;
        ctx->rows_bytes = 0;
        if (hipHostMalloc((void **)&ctx->h_rows, want, hipHostMallocDefault) != hipSuccess)
            return fail(ctx, RT_ERR_OUT_OF_MEMORY, "hipHostMalloc failed for the copy-back staging");
        ctx->rows_bytes = want;
    }
    RT_HIP(ctx, hipMemcpyAsync(ctx->h_rows, d_rgb + row0, want, hipMemcpyDeviceToHost, stream));
    RT_HIP(ctx, hipStreamSynchronize(stream));
    const size_t x_off = (size_t)rg.x0 * 3u, run = (size_t)rg.w * 3u;
    for (int r = 0; r < rg.h; ++r)
        memcpy(rgb + row0 + (size_t)r * pitch + x_off, ctx->h_rows + (size_t)r * pitch + x_off, run);
    return RT_OK;
}

int rt_render_device(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                     int ystart, int xsize, int ysize, uint8_t *d_rgb, void *stream)
{
    return render_device(ctx, cam, params, xstart, ystart, xsize, ysize, d_rgb, stream, false);
}

int rt_render_device_static(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                            int ystart, int xsize, int ysize, uint8_t *d_rgb, void *stream)
{
    return render_device(ctx, cam, params, xstart, ystart, xsize, ysize, d_rgb, stream, true);
}

int rt_render(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart, int ystart,
              int xsize, int ysize, uint8_t *rgb)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    int st = check_params(ctx, cam, params);
    if (st != RT_OK)
        return st;
    if (!rgb)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null rgb");
    RT_DEVICE(ctx);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t frame_bytes = (size_t)params->width * params->height * 3u;
    st = ensure(ctx, (void **)&ctx->d_rgb, &ctx->rgb_bytes, frame_bytes);
    if (st != RT_OK)
        return st;
    const Region rg = clip_region(*params, xstart, ystart, xsize, ysize);
    if (rg.w == 0 || rg.h == 0)
        return RT_OK;

    const hipEvent_t e0 = ctx->ev_t0, e1 = ctx->ev_t1;
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    st = rt_render_device(ctx, cam, params, xstart, ystart, xsize, ysize, ctx->d_rgb, ctx->stream);
    if (st == RT_OK) {
        hipError_t e = hipEventRecord(e1, ctx->stream);
        if (e != hipSuccess) {
            st = fail_hip(ctx, e, "rt_render");
        } else {
            st = copy_region_to_host(ctx, rgb, ctx->d_rgb, params->width, rg, ctx->stream);
            float ms = 0.f;
            if (st == RT_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess)
                ctx->last_kernel_ms = ms;
        }
    }
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

int rt_render_progressive(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                          int ystart, int xsize, int ysize, uint8_t *rgb, uint32_t samples_per_pass,
                          rt_progress_fn progress, void *user)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    int st = check_params(ctx, cam, params);
    if (st != RT_OK)
        return st;
    if (!rgb || samples_per_pass == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null rgb or zero samples_per_pass");
    if (params->sample_count != 0 || params->sample_begin != 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "progressive render takes the whole sample range");
    RT_DEVICE(ctx);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t frame_bytes = (size_t)params->width * params->height * 3u;
    st = ensure(ctx, (void **)&ctx->d_rgb, &ctx->rgb_bytes, frame_bytes);
    if (st == RT_OK)
        st = ensure(ctx, (void **)&ctx->d_sums, &ctx->sums_bytes, frame_bytes * sizeof(uint32_t));
    if (st != RT_OK)
        return st;
    const Region rg = clip_region(*params, xstart, ystart, xsize, ysize);
    if (rg.w == 0 || rg.h == 0)
        return RT_OK;
    const bool pipelined = samples_per_pass < params->spp; // more than one pass
    // Passes of few samples go several to a launch (the launch's tail, paths running out with the work queue
    // dry, costs as much as 2 samples per pixel): each of the group's passes adds into a sums frame of
    // its own, running totals are formed behind the launch, every pass's frame is resolved and they
    // are delivered one after the other, in order, while the next group is traced.  The first pass
    // goes alone, so that the first frame is not held back.
    const size_t sums_frame = frame_bytes * sizeof(uint32_t);
    uint32_t group = 1u;
    if (pipelined) {
        group = ctx->progressive_group ? ctx->progressive_group : std::max(16u / samples_per_pass, 1u);
        group = std::min(group, (uint32_t)kMaxJobs);
        while (group > 1u && (uint64_t)(group - 1u) * sums_frame > (1ull << 30))
            --group;
        const uint64_t tiles = (uint64_t)((rg.w + kTileW - 1) / kTileW) * (uint64_t)((rg.h + kTileH - 1) / kTileH);
        const uint64_t segs = (samples_per_pass + kSegmentSamplesMax - 1u) / kSegmentSamplesMax;
        if (tiles * segs * group > ctx->max_chunks)
            group = 1u; // (passes this large gain nothing from company; enqueue_accumulate cuts them up)
    }
    if (pipelined) {
        st = ensure(ctx, (void **)&ctx->d_rgb2, &ctx->rgb2_bytes, frame_bytes * std::max(group - 1u, 1u));
        if (st == RT_OK && group > 1u)
            st = ensure(ctx, (void **)&ctx->d_sums2, &ctx->sums2_bytes, sums_frame * (group - 1u));
        if (st != RT_OK)
            return st;
        if (!ctx->copy_stream)
            RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t &e : ctx->ev_resolved)
            if (!e)
                RT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipStream_t s = ctx->stream;
    if (ctx->trace_enqueued) // the sums are shared scratch: behind their previous user, as in rt_render_device
        RT_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_trace, 0));
    const size_t row_bytes = (size_t)params->width * 3u * sizeof(uint32_t);
    const size_t rows_off = (size_t)rg.y0 * row_bytes, rows_len = (size_t)rg.h * row_bytes;
    RT_HIP(ctx, hipMemsetAsync((uint8_t *)ctx->d_sums + rows_off, 0, rows_len, s));
    // sums frames: the running total lives in ring[total_at]; a group's pass j adds into ring[total_at + j]
    uint32_t *ring[kMaxJobs];
    uint8_t *frames[kMaxJobs];
    for (uint32_t j = 0; j < group; ++j) {
        ring[j] = j == 0u ? ctx->d_sums : (uint32_t *)((uint8_t *)ctx->d_sums2 + (size_t)(j - 1u) * sums_frame);
        frames[j] = j == 0u ? ctx->d_rgb : ctx->d_rgb2 + (size_t)(j - 1u) * frame_bytes;
    }
    uint32_t total_at = 0u;
    struct Group {
        uint32_t first = 0u, passes = 0u; // samples before the group, its number of passes
    };
    // Traces the passes of `g` (one launch) on top of the running total.
    auto enqueue_trace = [&](const Group &g) -> int {
        if (g.passes == 1u) {
            rt_params pass = *params;
            pass.sample_begin = g.first;
            pass.sample_count = std::min(samples_per_pass, params->spp - g.first);
            return enqueue_accumulate(ctx, cam, &pass, rg, ring[total_at], s);
        }
        JobSpec specs[kMaxJobs];
        for (uint32_t j = 0; j < g.passes; ++j) {
            uint32_t *into = ring[(total_at + j) % group];
            if (j > 0u)
                RT_HIP(ctx, hipMemsetAsync((uint8_t *)into + rows_off, 0, rows_len, s));
            specs[j] = { cam, *params, rg, into };
            specs[j].p.sample_begin = g.first + j * samples_per_pass;
            specs[j].p.sample_count = std::min(samples_per_pass, params->spp - specs[j].p.sample_begin);
        }
        return enqueue_jobs(ctx, specs, g.passes, s);
    };
    // Behind the trace: running totals pass by pass, each resolved into its frame.
    auto enqueue_frames = [&](const Group &g) -> int {
        uint32_t done = g.first;
        for (uint32_t j = 0; j < g.passes; ++j) {
            uint32_t *sums = ring[(total_at + j) % group];
            if (j > 0u) {
                AddArgs add;
                memset(&add, 0, sizeof add);
                add.dst = sums + rows_off / sizeof(uint32_t);
                add.src[0] = ring[(total_at + j - 1u) % group] + rows_off / sizeof(uint32_t);
                add.n_src = 1u;
                add.count = rows_len / sizeof(uint32_t);
                RT_HIP(ctx, launch_add_sums(add, s));
            }
            done += std::min(samples_per_pass, params->spp - done);
            ResolveArgs a;
            a.sums = sums; a.rgb = frames[j];
            a.width = params->width; a.height = params->height; a.spp_div = done; a.flags = params->flags;
            a.x0 = rg.x0; a.y0 = rg.y0; a.w = rg.w; a.h = rg.h;
            RT_HIP(ctx, launch_resolve(a, s));
        }
        total_at = (total_at + g.passes - 1u) % group;
        if (pipelined)
            RT_HIP(ctx, hipEventRecord(ctx->ev_resolved[0], s));
        RT_HIP(ctx, hipEventRecord(ctx->ev_trace, s));
        return RT_OK;
    };
    auto next_group = [&](const Group &g, uint32_t size) {
        Group n;
        n.first = std::min<uint64_t>((uint64_t)g.first + (uint64_t)g.passes * samples_per_pass, params->spp);
        const uint32_t left = (params->spp - n.first + samples_per_pass - 1u) / samples_per_pass;
        n.passes = std::min(size, left);
        return n;
    };
    // The group in hand is traced and resolved; the next one is traced while the host fetches the
    // frames of the one in hand and calls back, and resolved (into the same frames) afterwards.
    // (A callback that stops the render does so one enqueued group late: that group runs, its
    // samples are never shown.)
    Group cur;
    cur.passes = 1u;
    st = enqueue_trace(cur);
    if (st == RT_OK)
        st = enqueue_frames(cur);
    bool stop = false;
    while (st == RT_OK && !stop && cur.passes != 0u) {
        const Group next = next_group(cur, group);
        hipStream_t cs = s;
        if (pipelined) {
            cs = ctx->copy_stream;
            RT_HIP(ctx, hipStreamWaitEvent(cs, ctx->ev_resolved[0], 0));
            if (next.passes != 0u)
                st = enqueue_trace(next);
        }
        uint32_t done = cur.first;
        for (uint32_t j = 0; st == RT_OK && !stop && j < cur.passes; ++j) {
            done += std::min(samples_per_pass, params->spp - done);
            st = copy_region_to_host(ctx, rgb, frames[j], params->width, rg, cs);
            if (st == RT_OK && progress && progress(user, done, params->spp) != 0)
                stop = true;
        }
        if (st == RT_OK && !stop && next.passes != 0u)
            st = enqueue_frames(next);
        cur = next;
    }
    if (pipelined) // a group enqueued ahead of a stop, or of a failure
        RT_HIP(ctx, hipStreamSynchronize(s));
    if (st != RT_OK)
        return st;
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

static uint32_t own_tiles(uint32_t width, uint32_t height, uint32_t first, uint32_t stride)
{
    const uint64_t n = (uint64_t)((width + kTileW - 1) / kTileW) * ((height + kTileH - 1) / kTileH);
    if (stride == 0u)
        stride = 1u;
    return first < n ? (uint32_t)((n - first + stride - 1u) / stride) : 0u;
}

uint64_t rt_tile_block_bytes(uint32_t width, uint32_t height, uint32_t tile_first, uint32_t tile_stride)
{
    return (uint64_t)own_tiles(width, height, tile_first, tile_stride) * (uint64_t)(kTileW * kTileH * 3);
}

int rt_pack_tiles_device(rt_context *ctx, const uint32_t *d_sums, uint32_t width, uint32_t height,
                         uint32_t spp_div, uint32_t flags, uint32_t tile_first, uint32_t tile_stride,
                         uint8_t *d_tiles, void *stream)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!d_sums || !d_tiles || width == 0 || height == 0 || spp_div == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad pack arguments");
    if ((uint64_t)width * height > (1ull << 30))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "frame larger than 2^30 pixels");
    if ((flags & ~(uint32_t)RT_FLAG_GAMMA2) != 0u)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_pack_tiles_device takes RT_FLAG_GAMMA2 only");
    if (tile_stride == 0u)
        tile_stride = 1u;
    if (tile_first >= tile_stride)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "tile_first must be below tile_stride");
    RT_DEVICE(ctx);
    TilesArgs a;
    memset(&a, 0, sizeof a);
    a.sums = d_sums; a.tiles = d_tiles;
    a.width = width; a.height = height;
    a.tiles_x = (width + kTileW - 1) / kTileW; a.tiles_y = (height + kTileH - 1) / kTileH;
    a.spp_div = spp_div; a.flags = flags;
    a.tile_first = tile_first; a.tile_stride = tile_stride;
    a.n_own = own_tiles(width, height, tile_first, tile_stride);
    RT_HIP(ctx, launch_pack_tiles(a, (hipStream_t)stream));
    return RT_OK;
}

int rt_unpack_tiles_device(rt_context *ctx, const uint8_t *d_blocks, uint32_t n_blocks, uint64_t block_bytes,
                           uint32_t width, uint32_t height, uint8_t *d_rgb, void *stream)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!d_blocks || !d_rgb || width == 0 || height == 0 || n_blocks == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad unpack arguments");
    if ((uint64_t)width * height > (1ull << 30))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "frame larger than 2^30 pixels");
    if (block_bytes < rt_tile_block_bytes(width, height, 0u, n_blocks))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "block_bytes smaller than the first device's block");
    RT_DEVICE(ctx);
    TilesArgs a;
    memset(&a, 0, sizeof a);
    a.tiles_in = d_blocks; a.rgb = d_rgb;
    a.width = width; a.height = height;
    a.tiles_x = (width + kTileW - 1) / kTileW; a.tiles_y = (height + kTileH - 1) / kTileH;
    a.n_blocks = n_blocks; a.block_bytes = (size_t)block_bytes;
    RT_HIP(ctx, launch_unpack_tiles(a, (hipStream_t)stream));
    return RT_OK;
}

int rt_primary_hits_device(rt_context *ctx, const rt_camera *cam, uint32_t width, uint32_t height,
                           int32_t *d_hits, void *stream)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!cam || !d_hits || width == 0 || height == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad primary-hit arguments");
    if (!ctx->have_scene)
        return fail(ctx, RT_ERR_NO_SCENE, "rt_set_scene has not been called");
    RT_DEVICE(ctx);
    HitsArgs a;
    memset(&a, 0, sizeof a);
    if (chosen_filter(ctx) == kFilterBvh) {
        const int st = bvh_args(ctx, a.bvh);
        if (st != RT_OK)
            return st;
        a.bvh.suspend_lanes = 0u; // one ray per lane, nothing to go on to
        a.use_bvh = 1u;
    }
    a.geom = (ctx->scene_tame && ctx->filter != kFilterExact) ? ctx->d_geom_filter : nullptr;
    a.exact = ctx->d_geom_exact;
    a.n = ctx->n_spheres; a.cam = *cam;
    a.width = width; a.height = height; a.hits = d_hits;
    a.inv_width = 1.0 / (double)(float)width;
    a.inv_height = 1.0 / (double)(float)height;
    RT_HIP(ctx, launch_primary_hits(a, (hipStream_t)stream));
    return RT_OK;
}

int rt_primary_hits(rt_context *ctx, const rt_camera *cam, uint32_t width, uint32_t height, int32_t *hits)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!hits)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null hits");
    RT_DEVICE(ctx);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t bytes = (size_t)width * height * sizeof(int32_t);
    if (bytes == 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "empty frame");
    if ((uint64_t)width * height > (1ull << 30))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "frame larger than 2^30 pixels");
    int st = ensure(ctx, &ctx->d_io_out, &ctx->io_out_bytes, bytes); // kept for the next call
    if (st != RT_OK)
        return st;
    int32_t *d = (int32_t *)ctx->d_io_out;
    st = rt_primary_hits_device(ctx, cam, width, height, d, ctx->stream);
    if (st == RT_OK) {
        hipError_t e = hipMemcpyAsync(hits, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess)
            st = fail_hip(ctx, e, "rt_primary_hits copy-back");
    }
    ctx->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

int rt_intersect(rt_context *ctx, const float *rays, uint32_t n, void *out)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!ctx->have_scene)
        return fail(ctx, RT_ERR_NO_SCENE, "rt_set_scene has not been called");
    if (n == 0)
        return RT_OK;
    if (!rays || !out)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null rays or out");
    RT_DEVICE(ctx);
    const size_t in_bytes = (size_t)n * 6u * sizeof(float), out_bytes = (size_t)n * 32u;
    int st = ensure(ctx, &ctx->d_io_in, &ctx->io_in_bytes, in_bytes); // both kept for the next call
    if (st == RT_OK)
        st = ensure(ctx, &ctx->d_io_out, &ctx->io_out_bytes, out_bytes);
    if (st != RT_OK)
        return st;
    float *d_rays = (float *)ctx->d_io_in;
    uint8_t *d_out = (uint8_t *)ctx->d_io_out;
    hipError_t e = hipMemcpyAsync(d_rays, rays, in_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        IntersectArgs a;
        memset(&a, 0, sizeof a);
        if (chosen_filter(ctx) == kFilterBvh) {
            st = bvh_args(ctx, a.bvh);
            a.bvh.suspend_lanes = 0u;
            a.use_bvh = 1u;
        }
        a.geom = (ctx->scene_tame && ctx->filter != kFilterExact) ? ctx->d_geom_filter : nullptr;
        a.exact = ctx->d_geom_exact;
        a.spheres = ctx->d_spheres; a.n = ctx->n_spheres;
        a.rays = d_rays; a.n_rays = n; a.out = d_out;
        if (st == RT_OK)
            e = launch_intersect(a, ctx->stream);
    }
    if (e == hipSuccess && st == RT_OK)
        e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess && st == RT_OK)
        st = fail_hip(ctx, e, "rt_intersect");
    return st;
}

int rt_set_option(rt_context *ctx, int option, int64_t value)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    switch (option) {
    case RT_OPT_FILTER:
        if (value == 0) ctx->filter = kFilterExact;
        else if (value == 1) ctx->filter = kFilterConservative;
        else if (value == 2) ctx->filter = kFilterMatrix;
        else if (value == 3) ctx->filter = kFilterBvh;
        else if (value == -1) { ctx->filter = kFilterMatrix; ctx->filter_forced = false; break; }
        else return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_FILTER takes -1, 0, 1, 2 or 3");
        ctx->filter_forced = true;
        break;
    case RT_OPT_TILE_SPHERES:
        if (value < kPad || value > 4096)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_TILE_SPHERES takes 128..4096");
        ctx->tile_cap_max = (uint32_t)(value / kPad) * kPad;
        return RT_OK;
    case RT_OPT_WORKGROUPS_PER_CU:
        if (value < 0 || value > 8)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_WORKGROUPS_PER_CU takes 0..8");
        ctx->wg_per_cu_override = (int)value;
        return RT_OK;
    case RT_OPT_BVH_LEAF_SPHERES: {
        if (value < 0 || value > kBvhLeafMax)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_BVH_LEAF_SPHERES takes 0 (by scene size) or 1..15");
        const int before = ctx->bvh_leaf_max;
        ctx->bvh_leaf_max = (int)value;
        if (ctx->bvh_ready && ctx->bvh_leaf_max != before) { // the tree on the device may have another shape
            RT_DEVICE(ctx);
            RT_HIP(ctx, hipDeviceSynchronize());
            drop_bvh(ctx);
        }
        break;
    }
    case RT_OPT_BVH_LDS_NODES: {
        if (value < 0 || value > 2048)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_BVH_LDS_NODES takes 0..2048");
        // the automatic leaf size depends on what LDS may hold: a tree built for the old limit is
        // dropped and (below) rebuilt
        const uint32_t before = ctx->bvh_lds_nodes_max;
        ctx->bvh_lds_nodes_max = (uint32_t)value;
        if (ctx->bvh_ready && ctx->bvh_leaf_max == 0 && ctx->bvh_lds_nodes_max != before) {
            RT_DEVICE(ctx);
            RT_HIP(ctx, hipDeviceSynchronize());
            drop_bvh(ctx);
        }
        break;
    }
    case RT_OPT_BVH_SUSPEND_LANES:
        if (value < 0 || value > 63)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_BVH_SUSPEND_LANES takes 0..63");
        ctx->bvh_suspend_lanes = (uint32_t)value;
        return RT_OK;
    case RT_OPT_CHUNKS_PER_WAVE:
        if (value < 0 || value > 256)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_CHUNKS_PER_WAVE takes 0..256");
        ctx->chunks_per_wave = (uint32_t)value;
        return RT_OK;
    case RT_OPT_SUBTILES:
        if (value != 1 && value != 2 && value != 4 && value != 8)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_SUBTILES takes 1, 2, 4 or 8");
        ctx->subtiles_max = (uint32_t)value;
        return RT_OK;
    case RT_OPT_DIRECT_ADD_SAMPLES:
        if (value < 0 || value > 64)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_DIRECT_ADD_SAMPLES takes 0..64");
        ctx->direct_add_samples = (uint32_t)value;
        return RT_OK;
    case RT_OPT_COUNT_WORK:
        if (value != 0 && value != 1)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_COUNT_WORK takes 0 or 1");
        ctx->count_work = value == 1;
        return RT_OK;
    case RT_OPT_MAX_CHUNKS_PER_LAUNCH:
        if (value < 1 || value > 0x3fffffff)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_MAX_CHUNKS_PER_LAUNCH takes 1..2^30-1");
        ctx->max_chunks = (uint32_t)value;
        return RT_OK;
    case RT_OPT_PROGRESSIVE_PASSES_PER_LAUNCH:
        if (value < 0 || value > (int64_t)kMaxJobs)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_PROGRESSIVE_PASSES_PER_LAUNCH takes 0..8");
        ctx->progressive_group = (uint32_t)value;
        return RT_OK;
    case RT_OPT_BVH_MIN_SPHERES:
        if (value < 1 || value > (1ll << 30))
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_BVH_MIN_SPHERES takes 1..2^30");
        ctx->bvh_auto_min = (uint32_t)value;
        break;
    default:
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown option");
    }
    // an option that selects the hierarchy, or changes its shape, puts it on the device right away
    if (ctx->have_scene && chosen_filter(ctx) == kFilterBvh && !ctx->bvh_ready) {
        RT_DEVICE(ctx);
        return ensure_bvh(ctx);
    }
    return RT_OK;
}

int rt_get_option(rt_context *ctx, int option, int64_t *value)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!value)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null value");
    switch (option) {
    case RT_OPT_FILTER:
        *value = !ctx->filter_forced ? -1 : ctx->filter == kFilterExact ? 0 : ctx->filter == kFilterConservative ? 1
                 : ctx->filter == kFilterMatrix ? 2 : 3;
        break;
    case RT_OPT_TILE_SPHERES: *value = ctx->tile_cap_max; break;
    case RT_OPT_WORKGROUPS_PER_CU: *value = ctx->wg_per_cu_override; break;
    case RT_OPT_BVH_LEAF_SPHERES: *value = ctx->bvh_leaf_max; break;
    case RT_OPT_BVH_MIN_SPHERES: *value = ctx->bvh_auto_min; break;
    case RT_OPT_BVH_LDS_NODES: *value = ctx->bvh_lds_nodes_max; break;
    case RT_OPT_BVH_SUSPEND_LANES: *value = ctx->bvh_suspend_lanes; break;
    case RT_OPT_CHUNKS_PER_WAVE: *value = ctx->chunks_per_wave; break;
    case RT_OPT_SUBTILES: *value = ctx->subtiles_max; break;
    case RT_OPT_DIRECT_ADD_SAMPLES: *value = ctx->direct_add_samples; break;
    case RT_OPT_COUNT_WORK: *value = ctx->count_work ? 1 : 0; break;
    case RT_OPT_MAX_CHUNKS_PER_LAUNCH: *value = ctx->max_chunks; break;
    case RT_OPT_PROGRESSIVE_PASSES_PER_LAUNCH: *value = ctx->progressive_group; break;
    default:
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown option");
    }
    return RT_OK;
}

int rt_accel_selfcheck(const rt_sphere *spheres, uint32_t n, int leaf_spheres, uint32_t info[4])
{
    if ((n && !spheres) || leaf_spheres < 1 || leaf_spheres > kBvhLeafMax || n > (1u << 24))
        return RT_ERR_INVALID_ARGUMENT;
    std::vector<Geom> exact(n);
    for (uint32_t i = 0; i < n; ++i) {
        volatile float rr = spheres[i].radius * spheres[i].radius;
        // The builder sorts and bins by coordinates: it is only defined for the scenes that
        // rt_set_scene gives to it (finite, magnitudes below 2^28, r*r at least 2^-40); others
        // are scanned without a hierarchy and are refused here.
        if (!sphere_tame(spheres[i], rr))
            return RT_ERR_INVALID_ARGUMENT;
        exact[i].cx = spheres[i].center[0];
        exact[i].cy = spheres[i].center[1];
        exact[i].cz = spheres[i].center[2];
        exact[i].r2 = rr;
    }
    BvhHost b;
    bvh_build(spheres, exact.data(), n, leaf_spheres, b);
    if (info) {
        info[0] = (uint32_t)b.nodes.size();
        info[1] = b.levels;
        info[2] = b.n_large;
        info[3] = (uint32_t)b.leaf_geom.size();
    }
    return bvh_validate(spheres, n, b).empty() ? RT_OK : RT_ERR_INVALID_ARGUMENT;
}

int rt_get_stats(rt_context *ctx, rt_stats *stats)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!stats)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null stats");
    RT_DEVICE(ctx);
    RT_HIP(ctx, hipDeviceSynchronize());
    unsigned long long counters[kCounterWords];
    const int cst = read_counters(ctx, counters);
    if (cst != RT_OK)
        return cst;
    stats->node_visits = counters[2];
    stats->exact_tests = counters[3];
    stats->root_evaluations = counters[4];
    stats->rays = counters[0];
    stats->sphere_tests = counters[0] * (uint64_t)ctx->n_spheres;
    stats->wave_passes = counters[1];
    stats->last_kernel_ms = ctx->last_kernel_ms;
    stats->last_wall_ms = ctx->last_wall_ms;
    return RT_OK;
}

int rt_get_work_profile(rt_context *ctx, uint64_t *out, uint32_t n)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    if (!out || n > 28u)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_get_work_profile takes up to 28 pairs");
    RT_DEVICE(ctx);
    RT_HIP(ctx, hipDeviceSynchronize());
    unsigned long long counters[kCounterWords];
    const int cst = read_counters(ctx, counters);
    if (cst != RT_OK)
        return cst;
    for (uint32_t i = 0; i < 2u * n; ++i)
        out[i] = counters[8u + i];
    return RT_OK;
}

int rt_reset_stats(rt_context *ctx)
{
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(ctx->mtx);
    RT_DEVICE(ctx);
    RT_HIP(ctx, hipDeviceSynchronize());
    RT_HIP(ctx, hipMemset(ctx->d_rays, 0, kCounterBytes));
    ctx->last_kernel_ms = ctx->last_wall_ms = 0.0;
    return RT_OK;
}

} // extern "C"
