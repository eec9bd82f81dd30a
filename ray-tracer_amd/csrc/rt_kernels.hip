// rt_kernels.hip -- gfx950 kernels for the per-pixel ray-tracing path.
//
// What the kernels compute is fixed by the reference, operation by operation
// (Source/main.cpp, cited per function); how they compute it is CDNA4-specific:
//
//   * one lane = one light path.  A wave keeps its 64 lanes busy by refilling
//     finished lanes from an 8x8-pixel chunk of (pixel, sample) items it pulls
//     from a global queue (ballot + mbcnt ranks, no per-lane atomics), so waves
//     never idle on short paths while long ones finish;
//   * a section of the kernel costs a wave the same for one lane as for 64, so where a pass is cheap
//     (hierarchy walk, short lists) lanes wait for company: idle lanes for a refill of 24 or more,
//     rays with a hit for 32 ready to be shaded, finished paths for a common unwind before a refill.
//     None of it enters a path's arithmetic;
//   * a ray finds the spheres that need the reference's exact root evaluation through a
//     CONSERVATIVE first phase, which may pass a sphere the reference rejects but never the
//     reverse; whatever it passes is then resolved with the reference's exact, unfused
//     arithmetic and root selection, so results are bit-identical whichever phase 1 runs:
//       - a walk over a bounding-box hierarchy built on the host (boxes padded per ray by the
//         reference's own rounding slack; rt_bvh_walk.h), the default from 13 spheres up;
//       - the whole list, streamed through LDS tiles of {cx, cy, cz, r*r}, with a widened
//         discriminant test on the FP32 matrix cores or with fused
//         multiply-adds on the vector ALU (rt_list_scan.h), collecting one bit per test;
//       - the reference's own discriminant for every sphere;
//   * colour is byte-quantised at every recursion level in the reference, so
//     the iterative path records the spheres it bounced off in an LDS stack and
//     replays the per-level byte arithmetic in reverse once it terminates;
//   * GeneratePointInUnitSphere's rejection loop is shared out over the wave: idle lanes
//     evaluate attempts for the lanes still without a point (unit_sphere_points_wave);
//   * samples are summed as integers -- per wave in LDS, flushed to the frame once per
//     8x8 chunk -- so any split of samples or tiles over devices gives bit-identical frames.
//
// Build with -ffp-contract=off: every float multiply and add must round on its
// own, as in the reference's x86-64 build; sqrt and divide are IEEE (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt).
#include <type_traits>

#include "rt_arith.h"
#include "rt_bvh_walk.h"
#include "rt_device.h"
#include "rt_list_scan.h"

// Scheduling thresholds of the path kernel (lanes of a wave; overridable for sweeps): idle lanes before the
// walking kernels refill (_MEM: hierarchy walked in memory), hit rays before they are shaded (hierarchy in
// LDS, in memory; short lists, where 41 means whenever the next pass refills), idle lanes before the list
// scans refill.
#ifndef RT_REFILL_MIN_WALK
#define RT_REFILL_MIN_WALK 24u
#endif
#ifndef RT_SHADE_MIN_WALK
#define RT_SHADE_MIN_WALK 32u
#endif
#ifndef RT_SHADE_MIN_SHORT
#define RT_SHADE_MIN_SHORT 41u
#endif
#ifndef RT_REFILL_MIN_MEM
#define RT_REFILL_MIN_MEM 32u
#endif
#ifndef RT_SHADE_MIN_MEM
#define RT_SHADE_MIN_MEM 20u
#endif
#ifndef RT_REFILL_MIN_LIST
#define RT_REFILL_MIN_LIST 1u
#endif

namespace rtamd {
namespace {

// Lists up to this long are scanned by the reference's plain loop in the vector-scan kernel.

// The bounce stack holds list indices: two bytes per entry while the list has at most 65536
// spheres, else four.
__device__ __forceinline__ void push_bounce(uint32_t *stack, uint32_t entry_bytes, int at, uint32_t index)
{
    if (entry_bytes == 2u)
        reinterpret_cast<uint16_t *>(stack)[at] = (uint16_t)index;
    else
        stack[at] = index;
}

__device__ __forceinline__ uint32_t read_bounce(const uint32_t *stack, uint32_t entry_bytes, int at)
{
    return entry_bytes == 2u ? (uint32_t)reinterpret_cast<const uint16_t *>(stack)[at] : stack[at];
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m);

// GeneratePointInUnitSphere (Source/main.cpp:398-414) for the lanes of a wave that `want` a point,
// one Philox block per attempt, with the lanes helping
// each other.  A lane's attempts are the Philox blocks call, call + 1, ... of its own (pixel,
// sample) stream, and it must take the FIRST one inside the sphere; which lane evaluates an
// attempt does not matter.  Every lane tries its attempt 0 itself (52 % succeed).  The rest say
// in LDS which lanes they are, and in each further round all 64 lanes evaluate `per` consecutive attempts
// for each of them (per = the power of two that fits, at most 16); a lane takes its first success
// and advances its counter past it, or past all `per` attempts if none succeeded.  Two rounds
// usually do where the plain loop makes the wave wait for its unluckiest lane (five or six
// attempts).  Must be called by all 64 lanes; `coop` = 64 bytes of LDS owned by the wave.
__device__ __forceinline__ void unit_sphere_points_wave(bool want, uint32_t pixel, uint32_t sample, uint32_t &call,
                                                        uint32_t k0, uint32_t k1, uint8_t *coop, int lane,
                                                        float &x, float &y, float &z, WorkCount *wc = nullptr)
{
    // Wave votes are taken on comparisons of register values, and the set of lanes still looking
    // for a point is kept as a lane mask: a vote on a flag that was set under a branch costs two
    // more vector instructions to bring the flag into that form.
    float len2 = 0.0f;
    if (want) {
        uint32_t w[4];
        RT_TALLY(wc, kSecPointFirst);
        philox4x32_10(pixel, sample, call++, k0, k1, w);
        x = range_m1_p1(w[0]);
        y = range_m1_p1(w[1]);
        z = range_m1_p1(w[2]);
        len2 = (x * x) + (y * y) + (z * z);
    }
    unsigned long long nm = wave_ballot(!(len2 < 1.0f));
    while (nm != 0ull) {
        const bool need = __builtin_amdgcn_inverse_ballot_w64(nm);
        const uint32_t n = (uint32_t)__popcll(nm);
        // per = the largest power of two with per * n <= 64, at most 16: log2(per) = 6 - ceil(log2(n)),
        // in scalar bit operations (a division here would run on the vector ALU)
        uint32_t lg = 6u - (n > 1u ? 32u - (uint32_t)__builtin_clz(n - 1u) : 0u);
        lg = lg > 4u ? 4u : lg;
        const uint32_t per = 1u << lg;
        const uint32_t k = lane_rank(nm);
        // the k-th lane still looking says which lane it is; its helpers read its stream's
        // coordinates out of its registers
        if (need)
            coop[k] = (uint8_t)lane;
        __builtin_amdgcn_wave_barrier();
        const uint32_t t = (uint32_t)lane >> lg, at = (uint32_t)lane & (per - 1u);
        const int from = (int)coop[t < n ? t : 0u] * 4;
        const uint32_t h_pixel = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)pixel);
        const uint32_t h_sample = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)sample);
        const uint32_t h_call = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)call);
        float hx = 0.0f, hy = 0.0f, hz = 0.0f, hlen2 = 2.0f;
        if (t < n) {
            uint32_t w[4];
            RT_TALLY(wc, kSecPointHelp);
            philox4x32_10(h_pixel, h_sample, h_call + at, k0, k1, w);
            hx = range_m1_p1(w[0]);
            hy = range_m1_p1(w[1]);
            hz = range_m1_p1(w[2]);
            hlen2 = (hx * hx) + (hy * hy) + (hz * hz);
        }
        const unsigned long long okm = wave_ballot(hlen2 < 1.0f);
        __builtin_amdgcn_wave_barrier();
        uint32_t src = (uint32_t)lane, bits = 0u;
        if (need)
            bits = (uint32_t)(okm >> (k << lg)) & ((1u << per) - 1u);
        const bool got = bits != 0u;
        if (got) {
            const uint32_t first = (uint32_t)__builtin_ctz(bits);
            src = (k << lg) + first;
            call += first + 1u;
        } else if (need) {
            call += per;
        }
        const float gx = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src * 4u), __float_as_int(hx)));
        const float gy = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src * 4u), __float_as_int(hy)));
        const float gz = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src * 4u), __float_as_int(hz)));
        if (got) {
            x = gx; y = gy; z = gz;
        }
        nm &= ~wave_ballot(got);
    }
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m)
{
    // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---------------------------------------------------------------------------
// The path kernel: TraceThreadFunction's sample loop + GetRayColor
// (Source/main.cpp:551-598, 613-645), iterative.
// ---------------------------------------------------------------------------

// The kernel's arguments as they lie in its argument segment (constant memory, read with scalar
// loads), behind a pointer the compiler cannot see through.  Arguments that a pass needs once or less
// (camera, frame shape, hierarchy frame, output pointers) are read through this where they are used
// instead of being held in scalar registers from the kernel's start: held, they do not fit -- the
// compiler parks them in lanes of a vector register and fetches each back with a vector-ALU
// instruction per use, on the pipe that limits this kernel; a scalar load costs that pipe nothing.
typedef const __attribute__((address_space(4))) TraceArgs *ConstArgs;

__device__ __forceinline__ ConstArgs fresh_args()
{
    ConstArgs p = (ConstArgs)__builtin_amdgcn_kernarg_segment_ptr(); // TraceArgs is the kernel's only argument
    asm volatile("" : "+s"(p));
    return p;
}

// The hierarchy's frame, read afresh from the argument segment.
__device__ __forceinline__ BvhArgs fresh_bvh_args()
{
    const ConstArgs ka = fresh_args();
    BvhArgs b;
    b.nodes = ka->bvh.nodes;
    b.leaf_geom = ka->bvh.leaf_geom;
    b.leaf_index = ka->bvh.leaf_index;
    b.n_nodes = ka->bvh.n_nodes;
    b.n_large = ka->bvh.n_large;
    b.levels = ka->bvh.levels;
    b.lds_nodes = ka->bvh.lds_nodes;
    b.suspend_lanes = ka->bvh.suspend_lanes;
    b.center[0] = ka->bvh.center[0];
    b.center[1] = ka->bvh.center[1];
    b.center[2] = ka->bvh.center[2];
    b.radius2 = ka->bvh.radius2;
    b.rmin = ka->bvh.rmin;
    b.rmin2w = ka->bvh.rmin2w;
    return b;
}

// VARIANT: kPlain the reference's behaviour and nothing else; kExt with the rt_params.flags
// extensions compiled in; kCount kPlain plus per-lane tallies of the work the scan executes
// (RT_OPT_COUNT_WORK; slower, same results).
enum { kPlain = 0, kExt = 1, kCount = 2 };

// SPLIT: for the list scans, the list does not fit one LDS tile and is scanned tile by tile; for the
// hierarchy scan (which has no tiles), the nodes are read from memory instead of from an LDS copy.
template <bool SPLIT, int FILTER, int VARIANT>
__global__ void __launch_bounds__(kTraceThreads, FILTER == kFilterBvh || FILTER == kFilterShort ? (VARIANT == kCount ? 5 : 7) : (FILTER == kFilterMatrix && VARIANT != kCount ? 5 : 4))
trace_kernel(const TraceArgs a)
{
    constexpr bool EXT = VARIANT == kExt;
    constexpr bool MULTI_TILE = SPLIT && FILTER != kFilterBvh;
    constexpr bool NODES_IN_LDS = !SPLIT && FILTER == kFilterBvh;
    // (a pass of the list scans costs as much with one lane as with 64: they refill every idle lane at once)
    constexpr uint32_t kRefillMin = FILTER == kFilterBvh || FILTER == kFilterShort ? (SPLIT ? RT_REFILL_MIN_MEM : RT_REFILL_MIN_WALK) : RT_REFILL_MIN_LIST;
    // (a tree walked in memory makes the pass dearer: fewer rays wait, and a percent is all it gives)
    constexpr uint32_t kShadeMin = NODES_IN_LDS ? RT_SHADE_MIN_WALK : (FILTER == kFilterBvh ? RT_SHADE_MIN_MEM : 0u);
    constexpr uint32_t kShortShadeMin = RT_SHADE_MIN_SHORT;
    // the walking kernels unwind finished paths in batches, before a refill
    constexpr bool kDeferUnwind = kRefillMin > 1u;
    constexpr int kUnwindPending = 1 << 24; // (max_depth is at most 4096)
    // a tree in LDS has at most 2048 records: its bounce stack entries take two bytes, known here
    const uint32_t bounce_bytes = NODES_IN_LDS || FILTER == kFilterShort ? 2u : a.stack_entry_bytes;
    WorkCount work = {};
    WorkCount *const wc = VARIANT == kCount ? &work : nullptr;
    const unsigned long long t_start = VARIANT == kCount ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long t_mark = t_start;
    // the chip-wide constant-rate clock (the shader clock's counters differ between the chip's dies):
    // when the wave started and when it found the work queue empty
    const unsigned long long t_wall = VARIANT == kCount ? wall_clock64() : 0ull;
    unsigned long long t_dry = 0ull;
    extern __shared__ __align__(16) unsigned char smem[];
    // list scans: the tile's filter records, then its exact records -- except in the matrix scan, whose
    // few candidates fetch their exact records from memory (8 KB of LDS less per 512 spheres)
    Geom *tile_f = reinterpret_cast<Geom *>(smem);
    Geom *tile_e = FILTER == kFilterMatrix ? tile_f : tile_f + a.tile_cap;
    // matrix scan: 4 waves x 9 x 64 floats of operand staging; hierarchy scan: the walk's
    // stack, levels x 256 node references
    // (the walking kernels have no tiles: their area starts the workgroup's LDS, at a constant address)
    float *stage = FILTER == kFilterBvh || FILTER == kFilterShort ? reinterpret_cast<float *>(smem)
                                                                  : reinterpret_cast<float *>(tile_e + a.tile_cap);
    uint32_t *stack = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(stage) + a.aux_bytes);
    const int stack_levels = a.max_depth > 0 ? a.max_depth : 1;
    uint32_t *wave_flags = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(stack) +
                                                        (size_t)stack_levels * kTraceThreads * bounce_bytes);
    // 4 waves x 2 chunk slots x 64 pixels x 2 words: a pixel's red and green sums in the halves of one
    // word, its blue sum in the next (a chunk adds at most 256 samples of at most 255 to a pixel)
    uint32_t *acc_base = wave_flags + 4;
    const Geom *filt_src = FILTER == kFilterExact ? nullptr : a.geom;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t n_pad = round_up_group(a.n);

    if (!MULTI_TILE && FILTER != kFilterBvh && FILTER != kFilterShort) {
        load_tile(tile_f, FILTER == kFilterMatrix ? nullptr : tile_e, filt_src, a.exact, 0u, n_pad);
        __syncthreads();
    }
    // hierarchy scan: the nodes of a small tree live at the start of LDS, the walk's stack behind them
    // (its entries are two bytes then, four for a tree walked in memory).  In the LDS copy a child
    // reference to an inner node is that node's LDS address (below 2^15: it fits the stack's two
    // bytes), so the walk reads a node at the address its reference is, with no arithmetic.
    float4 *bvh_lds = reinterpret_cast<float4 *>(stage);
    unsigned char *walk_stack = reinterpret_cast<unsigned char *>(stage) + (size_t)a.bvh.lds_nodes * 64u;
    // 64 bytes per wave for unit_sphere_points_wave: the head of the wave's own operand staging
    // slice in the matrix scan (idle while paths are shaded), else a region of its own
    uint8_t *coop = FILTER == kFilterBvh
                        ? walk_stack + (size_t)a.bvh.levels * kTraceThreads * (a.bvh.lds_nodes ? 2u : 4u) + (threadIdx.x >> 6) * 64
                        : reinterpret_cast<uint8_t *>(stage) + (threadIdx.x >> 6) * (FILTER == kFilterMatrix ? 9 * 64 * 4 : 64);
    if (FILTER == kFilterBvh && a.bvh.lds_nodes) {
        for (uint32_t i = threadIdx.x; i < a.bvh.lds_nodes * 4u; i += blockDim.x) {
            float4 v = a.bvh.nodes[i];
            if ((i & 3u) < 2u) { // the node's first two quadwords end in its child references
                const int ref = __float_as_int(v.w);
                if (ref >= 0)
                    v.w = __int_as_float(lds_node_reference(bvh_lds, ref));
            }
            bvh_lds[i] = v;
        }
        __syncthreads();
    }

    // Per-wave colour sums of the (at most two) chunks the wave has paths in flight for.  They
    // are added to the frame once, when the chunk's last path has terminated: with plain loads
    // and stores while a tile's samples are one chunk (its 64 pixels then belong to this wave
    // alone during the launch), with atomics when the samples are split over several chunks.
    uint32_t *acc = acc_base + wave * (2 * 64 * 2);
    for (int i = lane; i < 2 * 64 * 2; i += 64)
        acc[i] = 0u;

    // wave-uniform queue state; slot s is free while s_total[s] == 0
    uint32_t w_next = 0u, w_end = 0u;
    // queue in use (every queue has workgroups of its own, which stay until it is empty), and how many
    // queues the wave may still try before it retires
    uint32_t q_at = blockIdx.x & (a.n_queues - 1u), q_left = min(kQueueProbes, a.n_queues);
    // item w_next of the current chunk is pixel w_pix of sample w_smp: w_next = w_smp * pixels + w_pix
    uint32_t w_smp = 0u, w_pix = 0u;
    float w_inv_npx = 1.0f, w_inv_tw = 1.0f; // reciprocals of the chunk's pixel count and width
    int cur = 0;
    uint32_t s_total0 = 0u, s_total1 = 0u, s_done0 = 0u, s_done1 = 0u;
    int s_x0[2] = { 0, 0 }, s_y0[2] = { 0, 0 }, s_tw[2] = { 1, 1 }, s_npx[2] = { 1, 1 };
    uint32_t s_smp[2] = { 0u, 0u }; // first sample of the chunk's segment
    uint32_t s_job[2] = { 0u, 0u }; // the launch's job the chunk belongs to
    bool queue_empty = (a.n_chunks == 0u);

    // per-lane path state
    bool active = false;
    RayState ray = {};
    uint32_t pixel = 0u, sample = 0u, call = 0u;
    uint32_t acc_at = 0u; // index of this path's pixel in acc: slot * 128 + pixel-in-chunk * 2
    int depth = 0, level = 0;
    // Rays (IntersectSceneSpheres calls): counted per lane and summed over the wave at its end, or, in the
    // kernel that walks a tree in memory, per wave in a scalar register (there the lane's register is the
    // dearer of the two: 13.8 -> 13.4 ms at 10 000 spheres; in the other kernels the scalar costs a percent).
    constexpr bool kScalarRayCount = FILTER == kFilterBvh && SPLIT;
    uint32_t n_rays = 0u;
    uint32_t n_rounds = 0u; // wave-uniform: passes over the sphere list
    // hierarchy scan: state of a walk that continues in the next pass
    int w_node = kBvhDone, w_best = -1;
    uint32_t w_sp = 0u;
    float w_closest = INFINITY;
    bool w_pending = false;
    bool waited_last = false; // wave-uniform: the hit rays of the last pass were made to wait

    // Unwinds the recursion of this lane's finished path -- per level, bytes -> float -> bytes (:569-578,
    // :587-591) -- and adds the sample to its pixel (FinalColor += ray_color, :633-635).  Black stays black
    // at every level (ToRGB of Color * 0, or of NaN for a non-finite Color, is 0), so the paths that ran
    // out of depth -- the longest stacks -- have nothing to unwind.
    auto unwind_and_add = [&](int cr, int cg, int cb) {
        for (int l = (cr | cg | cb) != 0 ? level - 1 : -1; l >= 0; --l) {
            const rt_sphere *sp = a.spheres + read_bounce(stack, bounce_bytes, l * kTraceThreads + tid);
            RT_TALLY(wc, kSecReplayLevel);
            if (sp->material == RT_MAT_DIFFUSE) {
                cr >>= 1; cg >>= 1; cb >>= 1;
            }
            cr = to_byte(sp->color[0] * (float)cr);
            cg = to_byte(sp->color[1] * (float)cg);
            cb = to_byte(sp->color[2] * (float)cb);
        }
        const ConstArgs kt = fresh_args();
        if (kt->direct_add) { // (single-job launches only)
            uint32_t *dst = kt->jobs[0].sums + (size_t)pixel * 3u;
            atomicAdd(dst + 0, (uint32_t)cr);
            atomicAdd(dst + 1, (uint32_t)cg);
            atomicAdd(dst + 2, (uint32_t)cb);
        } else {
            atomicAdd(acc + acc_at + 0u, (uint32_t)cr | ((uint32_t)cg << 16));
            atomicAdd(acc + acc_at + 1u, (uint32_t)cb);
        }
    };
    // The paths of the lanes in `term` have added their samples: count them off their chunks, and add
    // the sums of a chunk whose last path this was to the frame.
    auto retire = [&](unsigned long long term) {
        if (term != 0ull) {
            const unsigned long long term1 = term & wave_ballot(acc_at >= 128u);
            s_done1 += (uint32_t)__popcll(term1);
            s_done0 += (uint32_t)__popcll(term & ~term1);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const uint32_t total = sl ? s_total1 : s_total0, done = sl ? s_done1 : s_done0;
                if (total != 0u && done == total) {
                    if (lane < s_npx[sl]) {
                        const ConstArgs ka = fresh_args();
                        const uint32_t job = __builtin_amdgcn_readfirstlane(s_job[sl]);
                        const int ly = lane / s_tw[sl], lx = lane - ly * s_tw[sl];
                        uint32_t *dst = ka->jobs[job].sums +
                                        ((size_t)(s_x0[sl] + lx) + (size_t)(s_y0[sl] + ly) * ka->jobs[job].width) * 3u;
                        uint32_t *src = acc + sl * 128 + lane * 2;
                        const uint32_t sum[3] = { src[0] & 0xffffu, src[0] >> 16, src[1] };
                        src[0] = src[1] = 0u;
                        const bool with_atomics = ka->jobs[job].atomic_flush != 0u;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            if (with_atomics)
                                atomicAdd(dst + ch, sum[ch]);
                            else
                                dst[ch] += sum[ch];
                        }
                    }
                    if (sl) s_total1 = 0u; else s_total0 = 0u;
                }
            }
        }
    };

    unsigned long long idle_lanes = ~0ull; // the lanes without a path (= !active), kept as a lane mask too
    for (;;) {
        // ---- refill idle lanes from the wave's chunk, pulling new chunks as needed
        unsigned long long need = idle_lanes;
        // Idle lanes are refilled only once there are kRefillMin of them (or the wave holds no path):
        // the refill -- a Philox block and the camera ray in double -- costs the wave the same for one
        // lane as for 64, and a pass of the walking kernels is cheap enough that running a few of them
        // with some lanes empty costs less than refilling every pass.  Which pass a path starts in
        // does not enter its arithmetic.
        const bool refill_now = (uint32_t)__popcll(need) >= kRefillMin;
        if (kDeferUnwind) {
            // the paths that ended since the last refill unwind together: the loop runs to the deepest of
            // them once, instead of to the deepest of every pass's few
            const unsigned long long pend = wave_ballot(!active && (depth & kUnwindPending) != 0);
            if (pend != 0ull && (refill_now || queue_empty)) {
                if (__builtin_amdgcn_inverse_ballot_w64(pend)) {
                    const int packed = depth;
                    depth = 0;
                    unwind_and_add(packed & 255, (packed >> 8) & 255, (packed >> 16) & 255);
                }
                retire(pend);
            }
        }
        while (need != 0ull && !queue_empty && refill_now) {
            if (w_next >= w_end) {
                const ConstArgs kq = fresh_args(); // the call's shape is needed once per chunk: not worth registers
                // a new chunk needs a free accumulator slot; with both in use the idle lanes
                // wait until one of the two chunks in flight has drained
                const int fs = kq->direct_add ? 0 : (s_total0 == 0u ? 0 : (s_total1 == 0u ? 1 : -1));
                if (fs < 0)
                    break;
                // The queue has kQueues heads: atomics on one address complete at about 80 million
                // a second over the whole chip, which bounded calls with many small chunks.  A
                // wave starts on its workgroup's queue and moves to the next when one runs dry.
                if (kScalarRayCount && n_rays >= 0x80000000u) { // (a wave that has run for minutes: hand the count over before it wraps)
                    if (lane == 0)
                        atomicAdd(kq->rays + (size_t)(blockIdx.x % kCounterSlots) * kCounterWords, (unsigned long long)n_rays);
                    n_rays = 0u;
                }
                uint32_t c = kq->n_chunks;
                unsigned long long t_fetch = wc ? __builtin_amdgcn_s_memtime() : 0ull;
                while (q_left != 0u) {
                    uint32_t k = 0u;
                    if (lane == 0)
                        k = atomicAdd(kq->chunk_counter + q_at * kQueueStride, 1u);
                    k = __builtin_amdgcn_readfirstlane(k);
                    if (k < (kq->n_chunks + (kq->n_queues - 1u) - q_at) / kq->n_queues) {
                        c = k * kq->n_queues + q_at;
                        break;
                    }
                    q_at = (q_at + 1u) & (kq->n_queues - 1u);
                    --q_left;
                }
                RT_CLOCK(wc, kTimeFetch, t_fetch);
                if (c >= kq->n_chunks) {
                    queue_empty = true;
                    if (wc)
                        t_dry = wall_clock64();
                    break;
                }
                // the job whose chunk this is (the jobs' chunks follow each other in the launch's numbering)
                uint32_t job = 0u;
                while (job + 1u < kq->n_jobs && c >= kq->jobs[job + 1u].chunk_first)
                    ++job;
                job = __builtin_amdgcn_readfirstlane(job);
                const auto *kj = &kq->jobs[job];
                uint32_t k = c - kj->chunk_first, sub = 0u, seg = 0u;
                const uint32_t per_tile = kj->subtiles * kj->segments;
                if (per_tile > 1u) {
                    const uint32_t cj = k;
                    k = cj / per_tile;
                    const uint32_t r = cj - k * per_tile;
                    sub = r / kj->segments;
                    seg = r - sub * kj->segments;
                }
                const uint32_t seg_first = seg * kj->seg_samples;
                const uint32_t seg_count = min(kj->seg_samples, kj->sample_count - seg_first);
                const uint32_t t = kj->tile_first + k * kj->tile_stride;
                const uint32_t ty = t / kj->tiles_x, tx = t - ty * kj->tiles_x;
                // the tile, clipped to the region, then the chunk's strip of it (possibly empty)
                const int x0 = kj->x0 + (int)tx * kTileW;
                int y0 = kj->y0 + (int)ty * kTileH;
                const int tw = min(kTileW, kj->x0 + kj->w - x0);
                int th = min(kTileH, kj->y0 + kj->h - y0);
                if (kj->subtiles > 1u) { // strips of whole tile rows: the same memory lines as the tile's
                    const int rows = kTileH / (int)kj->subtiles;
                    y0 += (int)sub * rows;
                    th = min(rows, th - (int)sub * rows);
                }
                const int npx = (tw > 0 && th > 0) ? tw * th : 0;
                cur = fs;
                s_x0[fs] = x0; s_y0[fs] = y0; s_tw[fs] = max(tw, 1); s_npx[fs] = npx;
                s_smp[fs] = kj->sample_begin + seg_first;
                s_job[fs] = job;
                w_next = 0u;
                w_smp = 0u;
                w_pix = 0u;
                w_inv_npx = __builtin_amdgcn_rcpf((float)max(npx, 1));
                w_inv_tw = __builtin_amdgcn_rcpf((float)max(tw, 1));
                w_end = (uint32_t)npx * seg_count;
                if (kq->direct_add) { // no accumulator to flush: the slot stays free
                } else if (fs == 0) { s_total0 = w_end; s_done0 = 0u; } else { s_total1 = w_end; s_done1 = 0u; }
            }
            const int ch_x0 = s_x0[cur], ch_y0 = s_y0[cur], ch_tw = s_tw[cur], ch_npx = s_npx[cur];
            const uint32_t avail = w_end - w_next;
            const uint32_t rank = lane_rank(need);
            // the idle lanes that get an item, as a lane mask first (a vote on a register comparison),
            // then as this lane's flag
            const unsigned long long got_lanes = need & wave_ballot(rank < avail);
            const bool got = __builtin_amdgcn_inverse_ballot_w64(got_lanes);
            if (got) {
                RT_TALLY(wc, kSecRefill);
                // Item w_next + rank is j = w_pix + rank pixels on from sample w_smp's first.  j / pixels
                // and p / width as truncated float products: j < 128, pixels <= 64, so (j + 0.5) /
                // pixels is at least 1/128 away from an integer, far more than the error of the
                // reciprocal (within an ulp) and of the product.
                const uint32_t j = w_pix + rank;
                const uint32_t s = (uint32_t)(((float)j + 0.5f) * w_inv_npx);
                const uint32_t p = j - __umul24(s, (uint32_t)ch_npx);
                const int ly = (int)(uint32_t)(((float)p + 0.5f) * w_inv_tw);
                const int lx = (int)p - ly * ch_tw;
                const int px = ch_x0 + lx, py = ch_y0 + ly;
                const ConstArgs ka = fresh_args();
                const auto *kj = &ka->jobs[__builtin_amdgcn_readfirstlane(s_job[cur])];
                pixel = (uint32_t)px + (uint32_t)py * kj->width;
                sample = s_smp[cur] + w_smp + s;
                acc_at = (uint32_t)cur * 128u + p * 2u;
                call = 0u;
                uint32_t w[4];
                philox4x32_10(pixel, sample, call++, a.key0, a.key1, w);
                rt_camera cam;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    cam.origin[q] = kj->cam.origin[q];
                    cam.bottom_left[q] = kj->cam.bottom_left[q];
                    cam.horizontal[q] = kj->cam.horizontal[q];
                    cam.vertical[q] = kj->cam.vertical[q];
                }
                camera_ray(cam, kj->width_d, kj->height_d, kj->inv_width, kj->inv_height, px, py, (double)unit_float(w[0]),
                           (double)unit_float(w[1]), ray);
                depth = kj->max_depth; // >= 1: the host gives no job with RAY_DEPTH <= 0 to a launch
                level = 0;
                active = true;
            }
            const uint32_t want = (uint32_t)__popcll(need);
            const uint32_t taken = want < avail ? want : avail; // <= 64
            w_next += taken;
            w_pix += taken; // < 128
            const uint32_t whole = (uint32_t)(((float)w_pix + 0.5f) * w_inv_npx);
            w_smp += whole;
            w_pix -= __umul24(whole, (uint32_t)ch_npx);
            need &= ~got_lanes;
        }

        RT_CLOCK(wc, kTimeRefill, t_mark);
        // ---- termination
        // `need` has a bit for every lane that is still idle (all 64 lanes of a wave exist here)
        const unsigned long long active_lanes = ~need;
        idle_lanes = need;
        const bool wave_active = active_lanes != 0ull;
        if (MULTI_TILE) {
            if (lane == 0)
                wave_flags[wave] = wave_active ? 1u : 0u;
            __syncthreads();
            const uint32_t any_wave = wave_flags[0] | wave_flags[1] | wave_flags[2] | wave_flags[3];
            if (any_wave == 0u)
                break;
        } else if (!wave_active) {
            break;
        }

        if (active)
            RT_TALLY(wc, kSecPass);
        // ---- IntersectSceneSpheres(ray, 0.001f, inf), Source/main.cpp:562
        float closest = INFINITY;
        int best = -1;
        MatrixOperands mo;
        const bool mine = active && ray.tame; // this lane's ray goes through the matrix filter
        if (FILTER == kFilterMatrix && wave_active)
            matrix_operands(ray, mine, a.shift[0], a.shift[1], a.shift[2], stage + wave * (9 * 64), lane, mo);
        bool shade = active; // the ray's closest hit is known after this pass
        if (FILTER == kFilterBvh) {
            const BvhArgs bvh = fresh_bvh_args();
            if (active && !w_pending) { // a ray that starts its scan in this pass
                w_pending = true;
                if (ray.tame) {
                    bvh_begin(bvh, ray, 0.001f, w_node, w_sp, w_closest, w_best,
                              lds_stack_address(walk_stack + tid * (NODES_IN_LDS ? 2 : 4)), wc, NODES_IN_LDS ? lds_node_reference(bvh_lds, 0) : 0);
                } else { // the reference's in-order loop over the whole list, from memory
                    w_closest = INFINITY;
                    w_best = -1;
                    w_node = kBvhDone;
                    // (its per-ray factors are formed here, for the rare ray that needs them, rather than
                    // carried in registers by every ray)
                    RayState plain = ray;
                    ray_prepare(plain);
                    scan_tile<kFilterExact>(a.exact, a.exact, a.n, 0u, plain, 0.001f, w_closest, w_best, wc);
                }
            }
            if (wave_active) {
                // (a lane without a path, or whose ray has finished its walk or takes none, holds kBvhDone)
                if (NODES_IN_LDS)
                    bvh_walk<true>(bvh, bvh_lds, ray, reinterpret_cast<int16_t *>(walk_stack) + tid, 0.001f, w_node, w_sp,
                                   w_closest, w_best, wc);
                else
                    bvh_walk<false>(bvh, nullptr, ray, reinterpret_cast<uint32_t *>(walk_stack) + tid, 0.001f, w_node, w_sp,
                                    w_closest, w_best, wc);
            }
            if (active && w_node == kBvhDone) {
                w_pending = false;
                closest = w_closest;
                best = w_best;
            } else {
                shade = false; // still walking: continues in the next pass
            }
            // Hit rays wait a pass for company when fewer than kShadeMin of them are ready: drawing the
            // random points and scattering cost the wave the same for a few lanes as for 64 (two or three
            // Philox blocks, the hit frame's divisions), and the rays still walking or about to be
            // refilled join them in the next pass.  A waiting ray keeps its finished walk (w_node stays
            // kBvhDone, w_closest and w_best hold the hit).  Never two passes in a row, and only while
            // other rays are on their way, so every path advances.
            if (kShadeMin != 0u) {
                const uint32_t n_hit = (uint32_t)__popcll(wave_ballot(shade && best >= 0));
                const uint32_t n_walk = (uint32_t)__popcll(wave_ballot(active && !shade));
                const uint32_t n_free = queue_empty ? 0u : 64u - n_hit - n_walk; // idle, or about to end in the sky
                if (n_hit != 0u && n_hit < kShadeMin && !waited_last && n_walk + n_free >= 8u) {
                    if (shade && best >= 0) {
                        shade = false;
                        w_pending = true;
                    }
                    waited_last = true;
                } else {
                    waited_last = false;
                }
            }
        }
        if (FILTER == kFilterShort && active && !w_pending) {
            // the same records for every lane, never written while a kernel runs: scalar loads
            typedef const __attribute__((address_space(4))) float *ConstFloats;
            const ConstFloats eg = (ConstFloats)a.exact;
            // four records per round of loads (the array is padded far beyond the list's end)
            auto in_order = [&](auto tame_rays, const RayState &rs) {
                for (uint32_t i0 = 0u; i0 < a.n; i0 += 4u) {
                    float rec[16];
#pragma unroll
                    for (uint32_t q = 0u; q < 16u; ++q)
                        rec[q] = eg[4u * i0 + q];
#pragma unroll
                    for (uint32_t j = 0u; j < 4u; ++j) {
                        if (i0 + j < a.n) {
                            Geom g;
                            g.cx = rec[4u * j + 0u]; g.cy = rec[4u * j + 1u]; g.cz = rec[4u * j + 2u]; g.r2 = rec[4u * j + 3u];
                            if (decltype(tame_rays)::value ? resolve_candidate_tame(rs, g, 0.001f, closest, wc)
                                                           : resolve_candidate(rs, g, 0.001f, closest, wc))
                                best = (int)(i0 + j);
                        }
                    }
                }
            };
            // the short division and square root while every ray of the wave is tame, else (NaN, Inf,
            // zero or huge rays in the wave) the compiler's, with the per-ray factors formed here
            if (!wave_any(!ray.tame)) {
                in_order(std::true_type(), ray);
            } else {
                RayState plain = ray;
                ray_prepare(plain);
                in_order(std::false_type(), plain);
            }
        }
        if (FILTER == kFilterShort && kShortShadeMin != 0u) {
            // the same wait as in the hierarchy kernel; here only refilled lanes can join, so rays wait
            // only when the next pass will refill
            if (active && w_pending) {
                closest = w_closest;
                best = w_best;
                w_pending = false;
            }
            const uint32_t n_hit = (uint32_t)__popcll(wave_ballot(active && best >= 0));
            if (n_hit != 0u && n_hit < kShortShadeMin && !waited_last && !queue_empty && 64u - n_hit >= kRefillMin) {
                if (active && best >= 0) {
                    shade = false;
                    w_pending = true;
                    w_closest = closest;
                    w_best = best;
                }
                waited_last = true;
            } else {
                waited_last = false;
            }
        }
        for (uint32_t t0 = 0u; FILTER != kFilterBvh && FILTER != kFilterShort && t0 < (MULTI_TILE ? a.n : 1u); t0 += a.tile_cap) {
            const uint32_t cnt = MULTI_TILE ? min(a.tile_cap, a.n - t0) : a.n;
            if (MULTI_TILE) {
                __syncthreads(); // everyone is done with the previous tile (and with wave_flags)
                load_tile(tile_f, FILTER == kFilterMatrix ? nullptr : tile_e, filt_src, a.exact, t0, round_up_group(cnt));
                __syncthreads();
            }
            if (FILTER == kFilterMatrix) {
                if (wave_active)
                    scan_tile_matrix(reinterpret_cast<const float *>(tile_f), a.exact + t0, round_up_group(cnt), t0, mo,
                                     ray, mine, lane, 0.001f, closest, best, wc);
                if (active && !ray.tame) { // NaN, Inf, zero or huge rays: the reference's in-order loop
                    RayState plain = ray;
                    ray_prepare(plain);
                    scan_tile<kFilterExact>(a.exact + t0, a.exact + t0, cnt, t0, plain, 0.001f, closest, best, wc);
                }
            } else if (active) {
                scan_tile<FILTER>(tile_f, tile_e, cnt, t0, ray, 0.001f, closest, best, wc);
            }
        }
        if (wave_active)
            ++n_rounds;
        RT_CLOCK(wc, kTimeScan, t_mark);

        // ---- the random point of the Diffuse and Metal branches (:566, :584), drawn for the
        // whole wave at once; nothing else draws between the hit and this point
        int material = -1;
        if (shade && best >= 0)
            material = a.spheres[best].material;
        const bool wants_point = material == RT_MAT_DIFFUSE || material == RT_MAT_METAL;
        float rx = 0.0f, ry = 0.0f, rz = 0.0f;
        if (wave_any(wants_point))
            unit_sphere_points_wave(wants_point, pixel, sample, call, a.key0, a.key1, coop, lane, rx, ry, rz, wc);

        RT_CLOCK(wc, kTimePoints, t_mark);
        // ---- GetRayColor's branches, Source/main.cpp:562-597
        bool terminated = false;
        bool deferred = false; // terminated, unwind still to come
        if (kScalarRayCount)
            n_rays += (uint32_t)__popcll(wave_ballot(shade));
        if (shade) {
            if (!kScalarRayCount)
                ++n_rays;
            RT_TALLY(wc, kSecShade);
            int cr = 0, cg = 0, cb = 0;
            const bool float_shading = EXT && (a.flags & RT_FLAG_FLOAT_SHADING) != 0u;
            float fr = 0.0f, fg = 0.0f, fb = 0.0f; // terminal colour of the float-shading extension
            if (best < 0) {
                sky(ray.dy, cr, cg, cb);
                if (float_shading)
                    sky_float(ray.dy, fr, fg, fb);
                terminated = true;
            } else {
                const rt_sphere *sp = a.spheres + best;
                const bool dielectric = EXT && (a.flags & RT_FLAG_DIELECTRIC) != 0u && material == RT_MAT_GLASS;
                if (dielectric) {
                    // EXTENSION (RT_FLAG_DIELECTRIC; the reference's Glass is flat white, :594):
                    // index 1.5, Schlick reflectance against one RandomFloat() draw, glm::reflect /
                    // glm::refract (func_geometric.inl:110-127) on the normalised direction.
                    float px, py, pz, nx, ny, nz;
                    bool inside;
                    hit_frame<FILTER != kFilterExact>(ray, closest, sp->center[0], sp->center[1], sp->center[2], sp->radius,
                                                      px, py, pz, nx, ny, nz, inside);
                    const float inv = 1.0f / sqrtf(ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz);
                    const float ux = ray.dx * inv, uy = ray.dy * inv, uz = ray.dz * inv;
                    const float ratio = inside ? 1.5f : (1.0f / 1.5f);
                    const float nu = nx * ux + ny * uy + nz * uz;
                    float cos_theta = -nu;
                    if (cos_theta > 1.0f)
                        cos_theta = 1.0f;
                    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                    float r0 = (1.0f - ratio) / (1.0f + ratio);
                    r0 = r0 * r0;
                    const float m1 = 1.0f - cos_theta;
                    const float m5 = ((m1 * m1) * (m1 * m1)) * m1;
                    const float reflectance = r0 + (1.0f - r0) * m5;
                    uint32_t w[4];
                    philox4x32_10(pixel, sample, call++, a.key0, a.key1, w);
                    float ndx, ndy, ndz;
                    if (ratio * sin_theta > 1.0f || reflectance > unit_float(w[0])) {
                        ndx = ux - (nx * nu) * 2.0f;
                        ndy = uy - (ny * nu) * 2.0f;
                        ndz = uz - (nz * nu) * 2.0f;
                    } else {
                        const float kk = 1.0f - (ratio * ratio) * (1.0f - nu * nu);
                        const float f = ratio * nu + sqrtf(kk);
                        const float keep = kk >= 0.0f ? 1.0f : 0.0f;
                        ndx = (ratio * ux - f * nx) * keep;
                        ndy = (ratio * uy - f * ny) * keep;
                        ndz = (ratio * uz - f * nz) * keep;
                    }
                    push_bounce(stack, bounce_bytes, level * kTraceThreads + tid, (uint32_t)best);
                    ++level;
                    --depth;
                    ray.ox = px; ray.oy = py; ray.oz = pz;
                    ray.dx = ndx; ray.dy = ndy; ray.dz = ndz;
                    ray_prepare(ray);
                    if (depth <= 0)
                        terminated = true;
                } else if (material == RT_MAT_DIFFUSE || material == RT_MAT_METAL) {
                    float px, py, pz, nx, ny, nz;
                    bool inside;
                    RT_TALLY(wc, kSecScatter);
                    hit_frame<FILTER != kFilterExact>(ray, closest, sp->center[0], sp->center[1], sp->center[2], sp->radius,
                                                      px, py, pz, nx, ny, nz, inside);
                    float ndx, ndy, ndz;
                    if (material == RT_MAT_DIFFUSE) {
                        // S = Normal + GeneratePointInUnitSphere(), :566
                        ndx = nx + rx; ndy = ny + ry; ndz = nz + rz;
                    } else {
                        // glm::reflect: I - N * dot(N, I) * 2, then += fuzz * point, :583-584
                        const float d = nx * ray.dx + ny * ray.dy + nz * ray.dz;
                        const float fuzz = sp->fuzz;
                        ndx = ray.dx - (nx * d) * 2.0f;
                        ndy = ray.dy - (ny * d) * 2.0f;
                        ndz = ray.dz - (nz * d) * 2.0f;
                        ndx = ndx + (fuzz * rx); ndy = ndy + (fuzz * ry); ndz = ndz + (fuzz * rz);
                    }
                    push_bounce(stack, bounce_bytes, level * kTraceThreads + tid, (uint32_t)best);
                    ++level;
                    --depth;
                    ray.ox = px; ray.oy = py; ray.oz = pz;
                    ray.dx = ndx; ray.dy = ndy; ray.dz = ndz;
                    ray_prepare(ray);
                    if (depth <= 0) { // the child call returns RGB(0,0,0), :555-558
                        terminated = true;
                    }
                } else {
                    cr = cg = cb = 255; // :594
                    fr = fg = fb = 255.0f;
                    terminated = true;
                }
            }
            if (terminated && float_shading) {
                // EXTENSION (RT_FLAG_FLOAT_SHADING): the same replay with the colour kept in float,
                // limited to [0, 255] per level, truncated to bytes once per sample
                for (int l = level - 1; l >= 0; --l) {
                    const rt_sphere *sp = a.spheres + read_bounce(stack, bounce_bytes, l * kTraceThreads + tid);
                    if (sp->material == RT_MAT_DIFFUSE) {
                        fr = fr * 0.5f; fg = fg * 0.5f; fb = fb * 0.5f;
                    }
                    fr = limit_255(sp->color[0] * fr);
                    fg = limit_255(sp->color[1] * fg);
                    fb = limit_255(sp->color[2] * fb);
                }
                const ConstArgs kt = fresh_args();
                if (kt->direct_add) {
                    uint32_t *dst = kt->jobs[0].sums + (size_t)pixel * 3u;
                    atomicAdd(dst + 0, (uint32_t)to_byte(fr));
                    atomicAdd(dst + 1, (uint32_t)to_byte(fg));
                    atomicAdd(dst + 2, (uint32_t)to_byte(fb));
                } else {
                    atomicAdd(acc + acc_at + 0u, (uint32_t)to_byte(fr) | ((uint32_t)to_byte(fg) << 16));
                    atomicAdd(acc + acc_at + 1u, (uint32_t)to_byte(fb));
                }
                active = false;
            } else if (terminated) {
                if (kDeferUnwind) {
                    // the unwind waits for the wave's next refill, with the other paths that end until then
                    // (bytes of the colour in the path's depth counter, which has no other use from here on)
                    depth = kUnwindPending | cr | (cg << 8) | (cb << 16);
                    deferred = true;
                } else {
                    unwind_and_add(cr, cg, cb);
                }
                active = false;
            }
        }

        RT_CLOCK(wc, kTimeShade, t_mark);
        // ---- chunks whose last path has just terminated: add their sums to the frame
        const unsigned long long term = wave_ballot(terminated);
        idle_lanes |= term;
        retire(kDeferUnwind ? term & ~wave_ballot(deferred) : term);
        RT_CLOCK(wc, kTimeFlush, t_mark);
    }

    // ---- ray counter: one atomic per wave
    unsigned long long total = n_rays;
    if (!kScalarRayCount) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            total += __shfl_down(total, off, 64);
    }
    unsigned long long *counters = a.rays + (size_t)(blockIdx.x % kCounterSlots) * kCounterWords;
    if (lane == 0 && (total | n_rounds) != 0ull) {
        atomicAdd(counters, total);
        atomicAdd(counters + 1, (unsigned long long)n_rounds);
    }
    if (VARIANT == kCount) { // executed work: [2] node visits, [3] discriminants, [4] root evaluations
        unsigned long long w0 = work.nodes, w1 = work.exact, w2 = work.roots;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            w0 += __shfl_down(w0, off, 64);
            w1 += __shfl_down(w1, off, 64);
            w2 += __shfl_down(w2, off, 64);
        }
        if (lane == 0) {
            atomicAdd(counters + 2, w0);
            atomicAdd(counters + 3, w1);
            atomicAdd(counters + 4, w2);
        }
        for (int i = 0; i < kSecCount; ++i) { // [8 + 2i] wave executions, [9 + 2i] lane executions of section i
            unsigned long long sw = work.sec_waves[i], sl = work.sec_lanes[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sw += __shfl_down(sw, off, 64);
                sl += __shfl_down(sl, off, 64);
            }
            if (lane == 0) {
                atomicAdd(counters + 8 + 2 * i, sw);
                atomicAdd(counters + 9 + 2 * i, sl);
            }
        }
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        work.time[kTimeLife] = t_end - t_start;
        if (lane == 0) { // [32 + i] cycles in phase i, [40] waves
            for (int i = 0; i < kTimeCount; ++i)
                atomicAdd(counters + 32 + i, work.time[i]);
            atomicAdd(counters + 40, 1ull);
            // When the waves started, found the queue empty and ended, on the chip-wide clock
            // (its low 40 bits): sums in [41..43], and in the first slot alone (the host adds the
            // slots up) the latest of each in [44..46] and the complement of the earliest in [47..49].
            constexpr unsigned long long kLow = (1ull << 40) - 1ull;
            const unsigned long long t_last = wall_clock64();
            const unsigned long long at[3] = { t_wall & kLow, (t_dry ? t_dry : t_last) & kLow, t_last & kLow };
            for (int i = 0; i < 3; ++i) {
                atomicAdd(counters + 41 + i, at[i]);
                atomicMax(a.rays + 44 + i, at[i]);
                atomicMax(a.rays + 47 + i, kLow - at[i]);
            }
        }
    }
}

// FinalColor / SPP, ToRGB(ivec3), PutPixel -- Source/main.cpp:638-642.
__global__ void __launch_bounds__(256) resolve_kernel(const ResolveArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = (uint32_t)a.w * (uint32_t)a.h;
    if (i >= total)
        return;
    const uint32_t ly = i / (uint32_t)a.w, lx = i - ly * (uint32_t)a.w;
    const size_t loc = ((size_t)(a.x0 + (int)lx) + (size_t)(a.y0 + (int)ly) * a.width) * 3u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t q = a.sums[loc + c] / a.spp_div;
        q = q > 255u ? 255u : q;
        if (a.flags & RT_FLAG_GAMMA2) // EXTENSION: the reference writes the average as is
            q = (uint32_t)to_byte(sqrtf((float)q / 255.0f) * 255.0f);
        a.rgb[loc + c] = (uint8_t)q;
    }
}

// FinalColor / SPP, ToRGB (Source/main.cpp:638-642) for the 8x8 tiles first, first + stride, ... of
// the frame, written tile after tile (192 bytes each, pixel (lx, ly) of a tile at (ly*8 + lx)*3)
// instead of into the frame: the block one device of a tile partition sends to the gathering one.
// One thread per pixel of a tile; pixels of an edge tile that lie outside the frame are written 0.
__global__ void __launch_bounds__(256) pack_tiles_kernel(const TilesArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = i >> 6, p = i & 63u;
    if (k >= a.n_own)
        return;
    const uint32_t t = a.tile_first + k * a.tile_stride;
    const uint32_t ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const uint32_t x = tx * kTileW + (p & 7u), y = ty * kTileH + (p >> 3);
    uint8_t *dst = a.tiles + (size_t)i * 3u;
    if (x >= a.width || y >= a.height) {
        dst[0] = dst[1] = dst[2] = 0;
        return;
    }
    const size_t loc = ((size_t)x + (size_t)y * a.width) * 3u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t q = a.sums[loc + c] / a.spp_div;
        q = q > 255u ? 255u : q;
        if (a.flags & RT_FLAG_GAMMA2)
            q = (uint32_t)to_byte(sqrtf((float)q / 255.0f) * 255.0f);
        dst[c] = (uint8_t)q;
    }
}

// The gathered blocks of all `n_blocks` devices, block b = tiles b, b + n_blocks, ... as
// pack_tiles_kernel wrote them, each block block_bytes long: PutPixel (Source/main.cpp:273-281) of
// every pixel into the frame.  One thread per pixel of the frame's tile grid.
__global__ void __launch_bounds__(256) unpack_tiles_kernel(const TilesArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = i >> 6, p = i & 63u;
    if (t >= a.tiles_x * a.tiles_y)
        return;
    const uint32_t ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const uint32_t x = tx * kTileW + (p & 7u), y = ty * kTileH + (p >> 3);
    if (x >= a.width || y >= a.height)
        return;
    const uint32_t b = t % a.n_blocks, k = t / a.n_blocks;
    const uint8_t *src = a.tiles_in + (size_t)b * a.block_bytes + ((size_t)k * 64u + p) * 3u;
    uint8_t *dst = a.rgb + ((size_t)x + (size_t)y * a.width) * 3u;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// FinalColor += ... over devices (Source/main.cpp:633-635 is the per-sample add): integer sums, so
// the order of the operands does not matter.  HBM/xGMI-bound; four words per lane and step.
__global__ void __launch_bounds__(256) add_sums_kernel(const AddArgs a)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t quads = a.count / 4u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) {
        uint4 v = reinterpret_cast<const uint4 *>(a.dst)[i];
        for (uint32_t s = 0; s < a.n_src; ++s) {
            const uint4 w = reinterpret_cast<const uint4 *>(a.src[s])[i];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        reinterpret_cast<uint4 *>(a.dst)[i] = v;
    }
    for (size_t i = quads * 4u + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count; i += stride) {
        uint32_t v = a.dst[i];
        for (uint32_t s = 0; s < a.n_src; ++s)
            v += a.src[s][i];
        a.dst[i] = v;
    }
}

// Zeroes `count` 16-byte units.  Used in place of hipMemsetAsync on a stream under graph capture:
// replays of a captured memset of the frame's sums were seen to leave them as they were (every
// replay after the first doubled the picture), a captured kernel runs every time.
__global__ void __launch_bounds__(256) zero_kernel(uint4 *dst, size_t count)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        dst[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Deterministic primary-ray closest-hit map (jitter 0.5): one lane per pixel,
// the whole workgroup walks the sphere tiles together.
__global__ void __launch_bounds__(256) primary_hits_kernel(const HitsArgs a, uint32_t tile_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    Geom *tile_f = reinterpret_cast<Geom *>(smem);
    Geom *tile_e = tile_f + tile_cap;

    // 64x4 pixel footprint per workgroup
    const int px = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int py = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    const bool inside = px < (int)a.width && py < (int)a.height;

    RayState ray = {};
    if (inside)
        camera_ray(a.cam, (double)(float)a.width, (double)(float)a.height, a.inv_width, a.inv_height, px, py, 0.5, 0.5, ray);

    float closest = INFINITY;
    int best = -1;
    if (a.use_bvh) // smem is the walk's stack
        closest_hit_by_hierarchy(a.bvh, a.exact, a.n, ray, inside, reinterpret_cast<uint32_t *>(smem) + threadIdx.x,
                                 closest, best);
    for (uint32_t t0 = 0u; !a.use_bvh && t0 < a.n; t0 += tile_cap) {
        const uint32_t cnt = min(tile_cap, a.n - t0);
        __syncthreads();
        load_tile(tile_f, tile_e, a.geom, a.exact, t0, round_up_group(cnt));
        __syncthreads();
        if (inside) {
            if (a.geom)
                scan_tile<kFilterConservative>(tile_f, tile_e, cnt, t0, ray, 0.001f, closest, best);
            else
                scan_tile<kFilterExact>(tile_f, tile_e, cnt, t0, ray, 0.001f, closest, best);
        }
    }
    if (inside)
        a.hits[(size_t)px + (size_t)py * a.width] = best;
}

// IntersectSceneSpheres for caller-supplied rays, with the hit record.
__global__ void __launch_bounds__(256) intersect_kernel(const IntersectArgs a, uint32_t tile_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    Geom *tile_f = reinterpret_cast<Geom *>(smem);
    Geom *tile_e = tile_f + tile_cap;

    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inside = i < a.n_rays;
    RayState ray = {};
    if (inside) {
        const float *p = a.rays + (size_t)i * 6u;
        ray.ox = p[0]; ray.oy = p[1]; ray.oz = p[2];
        ray.dx = p[3]; ray.dy = p[4]; ray.dz = p[5];
        ray_prepare(ray);
    }
    float closest = INFINITY;
    int best = -1;
    if (a.use_bvh)
        closest_hit_by_hierarchy(a.bvh, a.exact, a.n, ray, inside, reinterpret_cast<uint32_t *>(smem) + threadIdx.x,
                                 closest, best);
    for (uint32_t t0 = 0u; !a.use_bvh && t0 < a.n; t0 += tile_cap) {
        const uint32_t cnt = min(tile_cap, a.n - t0);
        __syncthreads();
        load_tile(tile_f, tile_e, a.geom, a.exact, t0, round_up_group(cnt));
        __syncthreads();
        if (inside) {
            if (a.geom)
                scan_tile<kFilterConservative>(tile_f, tile_e, cnt, t0, ray, 0.001f, closest, best);
            else
                scan_tile<kFilterExact>(tile_f, tile_e, cnt, t0, ray, 0.001f, closest, best);
        }
    }
    if (!inside)
        return;
    int32_t *oi = reinterpret_cast<int32_t *>(a.out + (size_t)i * 32u);
    float *of = reinterpret_cast<float *>(oi);
    oi[0] = best;
    if (best >= 0) {
        const rt_sphere *sp = a.spheres + best;
        float px, py, pz, nx, ny, nz;
        bool inside;
        hit_frame(ray, closest, sp->center[0], sp->center[1], sp->center[2], sp->radius, px, py, pz,
                  nx, ny, nz, inside);
        of[1] = closest;
        of[2] = px; of[3] = py; of[4] = pz;
        of[5] = nx; of[6] = ny; of[7] = nz;
    } else {
#pragma unroll
        for (int k = 1; k < 8; ++k)
            of[k] = 0.0f;
    }
}

} // namespace

// ---------------------------------------------------------------------------
// Static scratch of the entry point that allocates nothing
// ---------------------------------------------------------------------------

// Per-pixel colour sums of a frame of up to kStaticMaxPixels pixels, then the work queue's heads.
// Every GPU the library runs on holds its own copy of this array, at its own address, until the
// process ends.  In host code the array's NAME is only a handle for hipGetSymbolAddress: the
// kernels get the address that call returns, as an argument, like any other buffer.
static __device__ __align__(256) unsigned char scratch[kStaticSumsBytes + kStaticQueueBytes];

hipError_t static_scratch_address(void **ptr)
{
    *ptr = nullptr;
    return hipGetSymbolAddress(ptr, HIP_SYMBOL(scratch));
}

// ---------------------------------------------------------------------------
// Launch wrappers
// ---------------------------------------------------------------------------

uint32_t trace_aux_bytes(int filter, uint32_t bvh_levels, uint32_t bvh_lds_nodes)
{
    if (filter == kFilterBvh)
        return (bvh_levels ? bvh_levels : 1u) * kTraceThreads * (bvh_lds_nodes ? 2u : 4u) + bvh_lds_nodes * 64u + 4u * 64u;
    // matrix scan: 4 waves x 9 x 64 floats of operand staging; the other list scans: the point helper's 64 bytes per wave
    return filter == kFilterMatrix ? 4u * 9u * 64u * (uint32_t)sizeof(float) : 4u * 64u;
}

size_t trace_lds_bytes(int filter, uint32_t tile_cap, uint32_t aux_bytes, int32_t max_depth, uint32_t stack_entry_bytes)
{
    const size_t levels = max_depth > 0 ? (size_t)max_depth : 1u;
    return (size_t)tile_cap * (filter == kFilterMatrix ? 1u : 2u) * sizeof(Geom) + aux_bytes +
           levels * kTraceThreads * stack_entry_bytes + 4u * sizeof(uint32_t) +
           4u * 2u * 64u * 2u * sizeof(uint32_t);
}

hipError_t launch_trace(const TraceArgs &a, uint32_t n_workgroups, hipStream_t stream)
{
    (void)hipGetLastError(); // a failure of some earlier, unrelated runtime call is not this launch\'s
    const size_t lds = trace_lds_bytes(a.filter, a.tile_cap, a.aux_bytes, a.max_depth, a.stack_entry_bytes);
    // the kernels' SPLIT: a list in several tiles, or a hierarchy whose nodes stay in memory
    const bool multi = a.filter == kFilterBvh ? a.bvh.lds_nodes == 0u : a.filter != kFilterShort && a.n > a.tile_cap;
    const dim3 grid(n_workgroups), block(kTraceThreads);
    const bool glass = (a.flags & (RT_FLAG_DIELECTRIC | RT_FLAG_FLOAT_SHADING)) != 0u; // the extended kernels
    const bool count = a.count_work != 0u && !glass; // tallies exist for the reference's behaviour only
#define RT_LAUNCH_TRACE(F)                                                                        \
    do {                                                                                          \
        if (multi && glass)                                                                       \
            hipLaunchKernelGGL((trace_kernel<true, F, kExt>), grid, block, lds, stream, a);       \
        else if (multi && count)                                                                  \
            hipLaunchKernelGGL((trace_kernel<true, F, kCount>), grid, block, lds, stream, a);     \
        else if (multi)                                                                           \
            hipLaunchKernelGGL((trace_kernel<true, F, kPlain>), grid, block, lds, stream, a);     \
        else if (glass)                                                                           \
            hipLaunchKernelGGL((trace_kernel<false, F, kExt>), grid, block, lds, stream, a);      \
        else if (count)                                                                           \
            hipLaunchKernelGGL((trace_kernel<false, F, kCount>), grid, block, lds, stream, a);    \
        else                                                                                      \
            hipLaunchKernelGGL((trace_kernel<false, F, kPlain>), grid, block, lds, stream, a);    \
    } while (0)
    switch (a.filter) {
    case kFilterExact: RT_LAUNCH_TRACE(kFilterExact); break;
    case kFilterConservative: RT_LAUNCH_TRACE(kFilterConservative); break;
    case kFilterMatrix: RT_LAUNCH_TRACE(kFilterMatrix); break;
    case kFilterBvh: RT_LAUNCH_TRACE(kFilterBvh); break;
    case kFilterShort: RT_LAUNCH_TRACE(kFilterShort); break;
    default:
        return hipErrorInvalidValue;
    }
#undef RT_LAUNCH_TRACE
    return hipGetLastError();
}

hipError_t trace_occupancy(int filter, bool counting, size_t lds_bytes, int *workgroups_per_cu)
{
    // (the instantiations of one scan share their launch bound, the counting builds apart)
#define RT_OCCUPANCY(FILTER)                                                                                   \
    (counting ? hipOccupancyMaxActiveBlocksPerMultiprocessor(workgroups_per_cu, trace_kernel<false, FILTER, kCount>, \
                                                             kTraceThreads, lds_bytes)                          \
              : hipOccupancyMaxActiveBlocksPerMultiprocessor(workgroups_per_cu, trace_kernel<false, FILTER, kPlain>, \
                                                             kTraceThreads, lds_bytes))
    if (filter == kFilterBvh)
        return RT_OCCUPANCY(kFilterBvh);
    if (filter == kFilterMatrix)
        return RT_OCCUPANCY(kFilterMatrix);
    if (filter == kFilterShort)
        return RT_OCCUPANCY(kFilterShort);
    return RT_OCCUPANCY(kFilterConservative);
#undef RT_OCCUPANCY
}

hipError_t launch_zero(void *dst, size_t bytes, hipStream_t stream)
{
    (void)hipGetLastError();
    if (bytes == 0u)
        return hipSuccess;
    if (((uintptr_t)dst | bytes) & 15u)
        return hipErrorInvalidValue;
    const size_t count = bytes / 16u;
    const size_t want = (count + 255u) / 256u;
    hipLaunchKernelGGL(zero_kernel, dim3((uint32_t)(want > 2048u ? 2048u : want)), dim3(256), 0, stream,
                       reinterpret_cast<uint4 *>(dst), count);
    return hipGetLastError();
}

hipError_t launch_resolve(const ResolveArgs &a, hipStream_t stream)
{
    (void)hipGetLastError(); // a failure of some earlier, unrelated runtime call is not this launch\'s
    const uint32_t total = (uint32_t)a.w * (uint32_t)a.h;
    if (total == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((total + 255u) / 256u), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_pack_tiles(const TilesArgs &a, hipStream_t stream)
{
    (void)hipGetLastError(); // a failure of some earlier, unrelated runtime call is not this launch\'s
    if (a.n_own == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((a.n_own * 64u + 255u) / 256u), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_unpack_tiles(const TilesArgs &a, hipStream_t stream)
{
    (void)hipGetLastError(); // a failure of some earlier, unrelated runtime call is not this launch\'s
    const uint32_t n_tiles = a.tiles_x * a.tiles_y;
    if (n_tiles == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3((n_tiles * 64u + 255u) / 256u), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_add_sums(const AddArgs &a, hipStream_t stream)
{
    (void)hipGetLastError(); // a failure of some earlier, unrelated runtime call is not this launch\'s
    if (a.count == 0u || a.n_src == 0u)
        return hipSuccess;
    const size_t want = (a.count / 4u + 255u) / 256u;
    const uint32_t blocks = (uint32_t)(want < 1u ? 1u : (want > 4096u ? 4096u : want));
    hipLaunchKernelGGL(add_sums_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

static uint32_t scan_tile_cap(uint32_t n)
{
    // spheres per LDS tile for the one-lane-per-ray kernels: up to 32 KiB
    const uint32_t n_pad = (n + (kGroup - 1)) & ~(uint32_t)(kGroup - 1);
    return n_pad < 2048u ? (n_pad ? n_pad : kGroup) : 2048u;
}

hipError_t launch_primary_hits(const HitsArgs &a, hipStream_t stream)
{
    (void)hipGetLastError(); // a failure of some earlier, unrelated runtime call is not this launch\'s
    if (a.width == 0u || a.height == 0u)
        return hipSuccess;
    const uint32_t cap = scan_tile_cap(a.n);
    dim3 grid((a.width + 63u) / 64u, (a.height + 3u) / 4u);
    const size_t lds = a.use_bvh ? (size_t)a.bvh.levels * 256u * sizeof(uint32_t) : (size_t)cap * 2u * sizeof(Geom);
    hipLaunchKernelGGL(primary_hits_kernel, grid, dim3(256), lds, stream, a, cap);
    return hipGetLastError();
}

hipError_t launch_intersect(const IntersectArgs &a, hipStream_t stream)
{
    (void)hipGetLastError(); // a failure of some earlier, unrelated runtime call is not this launch\'s
    if (a.n_rays == 0u)
        return hipSuccess;
    const uint32_t cap = scan_tile_cap(a.n);
    const size_t lds = a.use_bvh ? (size_t)a.bvh.levels * 256u * sizeof(uint32_t) : (size_t)cap * 2u * sizeof(Geom);
    hipLaunchKernelGGL(intersect_kernel, dim3((a.n_rays + 255u) / 256u), dim3(256), lds, stream, a, cap);
    return hipGetLastError();
}

} // namespace rtamd
