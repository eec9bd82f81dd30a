// rt_bvh_walk.h -- IntersectSceneSpheres as a per-ray walk over the bounding-box hierarchy built by
// rt_bvh.cpp; the spheres in the leaves it reaches get the reference's exact test.
#pragma once

#include "rt_arith.h"
#include "rt_list_scan.h"

namespace rtamd {
namespace {

// First half of resolve_unordered for a leaf record: the discriminant, and whether the root
// evaluation can be skipped.  Besides the reference's own `Discriminant < 0` early-out, a sphere
// with B > 0 and C > 0 (origin outside, centre behind) is dropped without a square root: then
// 4AC > 0, so fl(B*B - 4AC) <= fl(B*B), its correctly rounded root is at most sqrt(fl(B*B)) = |B|,
// both numerators -B -+ sq are <= 0 and the reference rejects both roots against tmin > 0.
// Only tame rays walk the tree: B is kept as d = B / 2 and the discriminant at a quarter (tame_root).
__device__ __forceinline__ bool leaf_candidate(const RayState &r, const Geom &s, float &d, float &disc)
{
    float C;
    disc = tame_discriminant(r, s, d, C);
    return !(disc < 0.0f) && !(d > 0.0f && C > 0.0f);
}

// Second half: the effective root (Source/main.cpp:490-499) and the closest-hit update; the
// list index is fetched only when the root is in range.
__device__ __forceinline__ void leaf_root(const RayState &r, float d, float disc, const uint32_t *index, float tmin,
                                          float &closest, int &best)
{
    const float root = tame_root(r, d, disc, tmin);
    if (root >= tmin && root <= closest) {
        const int idx = (int)*index;
        if (root < closest || idx > best) {
            closest = root;
            best = idx;
        }
    }
}

// The reference's exact test for records [first, first + count) of the leaf-ordered list, four at
// a time: up to four records in flight before the first is used, their discriminants as one block
// of independent arithmetic (all four unconditionally -- the record array ends in three records of
// padding -- and the bits of records beyond `count` dropped afterwards); then every lane walks its
// own candidates, so the wave pays for the most candidates any lane has, not for every record some
// lane hits.
__device__ __forceinline__ void test_records(const BvhArgs &b, const RayState &r, uint32_t first, uint32_t count,
                                             float tmin, float &closest, int &best, WorkCount *wc)
{
    for (uint32_t i0 = 0u; i0 < count; i0 += 4u) {
        Geom g[4];
        RT_TALLY(wc, kSecLeafGroup);
#pragma unroll
        for (uint32_t j = 0u; j < 4u; ++j)
            g[j] = *reinterpret_cast<const Geom *>(reinterpret_cast<const char *>(b.leaf_geom) + ((first + i0) << 4) + 16u * j);
        float Bj[4], dj[4];
        uint32_t cand = 0u;
#pragma unroll
        for (uint32_t j = 0u; j < 4u; ++j)
            if (leaf_candidate(r, g[j], Bj[j], dj[j]))
                cand |= 1u << j;
        cand &= (1u << min(4u, count - i0)) - 1u;
        if (wc) {
            wc->exact += min(4u, count - i0);
            wc->roots += (uint32_t)__popc(cand);
        }
        while (cand != 0u) {
            const uint32_t j = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1u;
            RT_TALLY(wc, kSecRoot);
            const float Bs = j == 0u ? Bj[0] : j == 1u ? Bj[1] : j == 2u ? Bj[2] : Bj[3];
            const float ds = j == 0u ? dj[0] : j == 1u ? dj[1] : j == 2u ? dj[2] : dj[3];
            leaf_root(r, Bs, ds, b.leaf_index + (first + i0 + j), tmin, closest, best);
        }
    }
}

// ---------------------------------------------------------------------------
// Hierarchy scan: a per-ray walk over boxes instead of a pass over the list
// ---------------------------------------------------------------------------
//
// A conservative filter like the two above, with the ray deciding which part
// of the list it looks at: spheres in leaves the walk does not reach are those
// the reference cannot accept, the others go through resolve_unordered.
//
// Why skipping a box is safe.  Write oc for the reference's rounded O - C,
// q = |oc|^2, and take its acceptance condition from filter_sign: the hit branch
// is taken only if the line through C + oc along D passes within
//     dist^2 <= r^2 + u (19.8 q + 5.2 r^2),                       u = 2^-24,
// of the centre, and the root T it then accepts lies (with the roundings of A, B,
// C, the square root and the division: |P - C|^2 <= r^2 + u (41 q + 25 r^2) for
// P = O + T D) at a point whose distance from C is below
//     r_eff = sqrt(r^2 (1 + 64u) + 64u q).
// So the segment [tmin, closest] of the line meets the ball of radius r_eff at the
// accepted T, hence also every box that contains the sphere's box once it is
// padded by r_eff - |r|, which is largest for the smallest radius in the tree:
//     pad >= sqrt(rmin^2 (1 + 64u) + 64u qmax) - rmin,   qmax = (|O - center| + radius)^2 >= q.
// The slab test runs in coordinates relative to the tree's centre; its own
// roundings move a plane by less than 8u (|O - center| + radius) and scale a t by
// less than 1 + 4u, which an extra 2^-10 r_eff of padding (r_eff >= sqrt(64u qmax),
// so this is at least 2^-19 sqrt(qmax) = 32u sqrt(qmax)), the factor 1.001 on the
// whole and the 2^-20 relative slack on [tmin, closest] cover.  A direction component
// too small to divide by is replaced by 2^-50 of the largest one: over any distance
// inside the tree that moves the ray by less than 2^-49 of the tree's size.  Only
// tame rays walk the tree.
//
// The walk is resumable.  Its per-ray state is (node, sp, closest, best) plus the ray's stack
// column in LDS; a wave stops walking when no more than `suspend` of its lanes are still at it
// (unless that was already so when it started), the finished rays go on to shading, and the
// stragglers pick their walk up again in the wave's next pass, next to freshly started rays.
// That cuts the tail of iterations a wave would spend with a handful of lanes active.
// node: >= 0 inner node to visit, negative leaf code to process, kBvhDone when finished.
constexpr int kBvhDone = INT_MIN;

// max(|x|, m) in one instruction (written with fmaxf and fabsf, the compiler first spends one to
// canonicalise |x|).
__device__ __forceinline__ float max_abs(float x, float m)
{
    float o;
    asm("v_max_f32_e64 %0, |%1|, %2" : "=v"(o) : "v"(x), "v"(m));
    return o;
}

// The reference to node `index` of a tree copied to LDS at `lds_nodes`: the node's LDS address (inner
// nodes' references are >= 0 either way; leaf codes and kBvhDone are negative).
__device__ __forceinline__ int lds_node_reference(const float4 *lds_nodes, int index)
{
    return (int)(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)lds_nodes + (index << 6);
}

// The LDS address of a lane's (empty) walk stack: what bvh_begin's `stack` and bvh_walk's `sp` start from.
__device__ __forceinline__ uint32_t lds_stack_address(const void *stk)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)stk;
}

// Start of a ray's walk: the always-tested spheres, then the root (`root`: its reference; `stack`: the
// lane's stack, lds_stack_address()).
__device__ __forceinline__ void bvh_begin(const BvhArgs &b, const RayState &r, float tmin, int &node, uint32_t &sp,
                                          float &closest, int &best, uint32_t stack, WorkCount *wc = nullptr, int root = 0)
{
    closest = INFINITY;
    best = -1;
    RT_TALLY(wc, kSecWalkBegin);
    // the same records for every lane, never written while a kernel runs: read with scalar loads
    // (constant address space) instead of one vector load per lane and sphere
    typedef const __attribute__((address_space(4))) float *ConstFloats;
    typedef const __attribute__((address_space(4))) uint32_t *ConstWords;
    const ConstFloats lg = (ConstFloats)b.leaf_geom;
    const ConstWords li = (ConstWords)b.leaf_index;
    for (uint32_t i = 0; i < b.n_large; ++i) {
        Geom g;
        g.cx = lg[4u * i + 0u]; g.cy = lg[4u * i + 1u]; g.cz = lg[4u * i + 2u]; g.r2 = lg[4u * i + 3u];
        resolve_unordered_tame(r, g, (int)li[i], tmin, closest, best, wc);
    }
    node = b.n_nodes ? root : kBvhDone;
    sp = stack;
}

// Must be called by all 64 lanes.  The lanes whose node is not kBvhDone walk (they hold a tame ray
// and the state bvh_begin or an earlier call left); every other lane must hold kBvhDone.
// Stack entries are child references: four bytes, or two (sign-extended when read) for a tree in LDS,
// whose host side makes sure that every reference fits (kBvhLdsRecords).
template <bool NODES_IN_LDS, typename StackEntry>
__device__ __forceinline__ void bvh_walk(const BvhArgs &b, const float4 *lds_nodes, const RayState &r,
                                         StackEntry *stk, float tmin, int &node, uint32_t &sp,
                                         float &closest, int &best, WorkCount *wc = nullptr)
{
    const uint32_t n_start = (uint32_t)__popcll(wave_ballot(node != kBvhDone));
    if (n_start == 0u)
        return;
    const uint32_t suspend = n_start > b.suspend_lanes ? b.suspend_lanes : 0u;

    if (node != kBvhDone)
        RT_TALLY(wc, kSecWalkSetup);
    const float ox = r.ox - b.center[0], oy = r.oy - b.center[1], oz = r.oz - b.center[2];
    // reach^2 = (|o| + radius)^2 <= 2 (|o|^2 + radius^2); the hardware square root (1 ulp) is
    // inside the 1.001.  Since reff^2 >= 2^-18 reach^2, the slab roundings' 40u reach is below
    // 2^-10 reff.
    const float reach2 = 2.0f * __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, __builtin_fmaf(ox, ox, b.radius2)));
    const float reff = __builtin_amdgcn_sqrtf(__builtin_fmaf(reach2, 0x1p-18f * (1.0f + 0x1p-10f), b.rmin2w));
    const float pad = __builtin_fmaf(reff, 0x1.01p-10f, reff - b.rmin) * 1.001f;

    const float dmax = fmaxf(fmaxf(fabsf(r.dx), fabsf(r.dy)), fabsf(r.dz));
    const float dlim = dmax * 0x1p-50f;
    const float ddx = copysignf(max_abs(r.dx, dlim), r.dx);
    const float ddy = copysignf(max_abs(r.dy, dlim), r.dy);
    const float ddz = copysignf(max_abs(r.dz, dlim), r.dz);
    const float ix = __builtin_amdgcn_rcpf(ddx), iy = __builtin_amdgcn_rcpf(ddy), iz = __builtin_amdgcn_rcpf(ddz);
    // t at plane p: (p - o) / d = fma(p, i, -o i).  The box is padded, so the plane the ray meets
    // first lies pad further back and the one it meets last pad further on, whichever way it runs:
    // the entry planes' constant is -o i - pad |i|, the exit planes' -o i + pad |i| (a product and a
    // fused multiply-add each: two roundings of relative size u on |o i| + pad |i|, as counted above).
    const float mx = -(ox * ix), my = -(oy * iy), mz = -(oz * iz);
    const float ncx = __builtin_fmaf(-pad, fabsf(ix), mx), fcx = __builtin_fmaf(pad, fabsf(ix), mx);
    const float ncy = __builtin_fmaf(-pad, fabsf(iy), my), fcy = __builtin_fmaf(pad, fabsf(iy), my);
    const float ncz = __builtin_fmaf(-pad, fabsf(iz), mz), fcz = __builtin_fmaf(pad, fabsf(iz), mz);
    // Which of the two planes of an axis the ray meets first depends on the sign of d.  Instead
    // of min/max per box (half-rate instructions), each t is fma(lo, a, fma(hi, b, c)) with one
    // of a, b zero: the same single rounding, on full-rate instructions only.
    // entry plane: fma(lo, na, fma(hi, nb, nc)); exit plane: the same with na and nb swapped
    const float nax = fmaxf(ix, 0.0f), nbx = ix - nax;
    const float nay = fmaxf(iy, 0.0f), nby = iy - nay;
    const float naz = fmaxf(iz, 0.0f), nbz = iz - naz;
    const float fax = nbx, fbx = nax, fay = nby, fby = nay, faz = nbz, fbz = naz;
    const float t_lo = tmin * (1.0f - 0x1p-20f);
    float t_hi = closest * (1.0f + 0x1p-20f);

    // sp: LDS address of the lane's next free stack entry (entries of one lane lie kStackStep bytes apart)
    typedef __attribute__((address_space(3))) StackEntry *LdsStack;
    constexpr uint32_t kStackStep = kTraceThreads * (uint32_t)sizeof(StackEntry);
    const uint32_t stk_base = lds_stack_address(stk);

    constexpr int kDone = kBvhDone;
    while ((uint32_t)__popcll(wave_ballot(node != kDone)) > suspend) {
        while (node >= 0) {
            if (wc)
                ++wc->nodes;
            RT_TALLY(wc, kSecNode);
            // (a tree in memory: the node's byte offset as an unsigned 32-bit number, which the load
            // takes next to the array's address in scalar registers -- no 64-bit address arithmetic;
            // a tree in LDS: the reference IS the node's LDS address, see lds_node_reference)
            // (the second box's quadwords end in padding: three words each)
            typedef float f32x3 __attribute__((ext_vector_type(3)));
            f32x4 l0, h0;
            f32x3 l1, h1;
            if (NODES_IN_LDS) {
                typedef const __attribute__((address_space(3))) char *LdsNode;
                const LdsNode np = (LdsNode)(uintptr_t)(uint32_t)node;
                l0 = *(const __attribute__((address_space(3))) f32x4 *)(np);
                h0 = *(const __attribute__((address_space(3))) f32x4 *)(np + 16);
                l1 = *(const __attribute__((address_space(3))) f32x3 *)(np + 32);
                h1 = *(const __attribute__((address_space(3))) f32x3 *)(np + 48);
            } else {
                const char *np = reinterpret_cast<const char *>(b.nodes) + ((uint32_t)node << 6);
                l0 = *reinterpret_cast<const f32x4 *>(np);
                h0 = *reinterpret_cast<const f32x4 *>(np + 16);
                l1 = *reinterpret_cast<const f32x3 *>(np + 32);
                h1 = *reinterpret_cast<const f32x3 *>(np + 48);
            }
#define RT_SLAB(lo, hi, na, nb, nc) __builtin_fmaf(lo, na, __builtin_fmaf(hi, nb, nc))
            const float n0 = fmaxf(fmaxf(fmaxf(RT_SLAB(l0.x, h0.x, nax, nbx, ncx), RT_SLAB(l0.y, h0.y, nay, nby, ncy)),
                                         RT_SLAB(l0.z, h0.z, naz, nbz, ncz)), t_lo);
            const float f0 = fminf(fminf(fminf(RT_SLAB(l0.x, h0.x, fax, fbx, fcx), RT_SLAB(l0.y, h0.y, fay, fby, fcy)),
                                         RT_SLAB(l0.z, h0.z, faz, fbz, fcz)), t_hi);
            const float n1 = fmaxf(fmaxf(fmaxf(RT_SLAB(l1.x, h1.x, nax, nbx, ncx), RT_SLAB(l1.y, h1.y, nay, nby, ncy)),
                                         RT_SLAB(l1.z, h1.z, naz, nbz, ncz)), t_lo);
            const float f1 = fminf(fminf(fminf(RT_SLAB(l1.x, h1.x, fax, fbx, fcx), RT_SLAB(l1.y, h1.y, fay, fby, fcy)),
                                         RT_SLAB(l1.z, h1.z, faz, fbz, fcz)), t_hi);
#undef RT_SLAB
            const bool hit0 = n0 <= f0, hit1 = n1 <= f1;
            const int c0 = __float_as_int(l0.w), c1 = __float_as_int(h0.w);
            if (hit0 && hit1) {
                const bool first0 = n0 <= n1;
                *(LdsStack)(uintptr_t)sp = (StackEntry)(first0 ? c1 : c0);
                sp += kStackStep;
                node = first0 ? c0 : c1;
            } else if (hit0) {
                node = c0;
            } else if (hit1) {
                node = c1;
            } else if (sp == stk_base) {
                node = kDone;
            } else {
                sp -= kStackStep;
                node = (int)*(LdsStack)(uintptr_t)sp;
            }
        }
        if (node == kDone)
            continue;
        const uint32_t code = (uint32_t)~node;
        test_records(b, r, code >> 4, code & 15u, tmin, closest, best, wc);
        t_hi = closest * (1.0f + 0x1p-20f);
        if (sp == stk_base) {
            node = kDone;
        } else {
            sp -= kStackStep;
            node = (int)*(LdsStack)(uintptr_t)sp;
        }
    }
}

// One whole walk for the one-ray-per-lane kernels (all 64 lanes call; `have` = this lane holds a
// ray).  Rays outside the walk's preconditions take the reference's in-order loop from memory.
__device__ __forceinline__ void closest_hit_by_hierarchy(const BvhArgs &b, const Geom *exact, uint32_t n,
                                                         const RayState &r, bool have, uint32_t *stk,
                                                         float &closest, int &best)
{
    int node = kBvhDone;
    uint32_t sp = 0u;
    if (have) {
        if (r.tame)
            bvh_begin(b, r, 0.001f, node, sp, closest, best, lds_stack_address(stk));
        else
            scan_tile<kFilterExact>(exact, exact, n, 0u, r, 0.001f, closest, best);
    }
    bvh_walk<false>(b, nullptr, r, stk, 0.001f, node, sp, closest, best);
}


} // namespace
} // namespace rtamd
