// rt_list_scan.h -- IntersectSceneSpheres as a pass over the whole sphere list in LDS tiles
// (Source/main.cpp:534-546): a conservative first phase (vector ALU or FP32 matrix cores) marks the
// spheres whose `Discriminant < 0` early-out can fail, the second resolves them exactly.
#pragma once

#include "rt_arith.h"

namespace rtamd {
namespace {

// Conservative form of the `Discriminant < 0` early-out, Source/main.cpp:481.
//
// Returns a float whose SIGN BIT is set only if the reference's test certainly
// rejects the sphere.  With oc = O - C (the same three rounded subtractions as
// the reference), d* = oc.D, q* = oc.oc and A = fl(D.D) as exact real numbers,
// the reference takes the hit branch only if
//     d*^2 - A (q* - r^2)  >=  -u A (12.4 q* + 2.1 r^2),      u = 2^-24,
// (its 3-term dot products carry a relative error of at most 3u on the sum of
// absolute terms, |d*| <= sqrt(A_true q*), and each of B*B, 4A*C, q - r^2 adds
// one rounding).  The filter evaluates
//     s = d~^2 - A(1 - 2^-17) (q~ - r^2 (1 + 2^-16))
// with fused chains whose errors are bounded the same way, which gives
//     s >= [d*^2 - A (q* - r^2)] + u A (128 q* + 63.9 r^2) - u A (10.4 q* + 4.2 r^2),
// so whenever the reference accepts, s >= u A (105 q* + 57 r^2) > 0: no false
// reject.  False accepts (relative width 1e-5) only cost one exact evaluation.
// Preconditions, checked per ray (RayState::tame) and per scene (host side):
// every magnitude below 2^28 (no overflow: all intermediates stay under 2^127)
// and A, r^2 >= 2^-40 (the margin exceeds any underflow loss by > 10^10).
// Rays or scenes outside them take every sphere as a candidate.
__device__ __forceinline__ float filter_sign(const RayState &r, const Geom &s)
{
    const float ocx = r.ox - s.cx, ocy = r.oy - s.cy, ocz = r.oz - s.cz;
    const float d = __builtin_fmaf(ocz, r.dz, __builtin_fmaf(ocy, r.dy, ocx * r.dx));
    // s.r2 holds r^2 (1 + 2^-16) in the filter's copy of the geometry
    const float c = __builtin_fmaf(ocx, ocx, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocz, ocz, -s.r2)));
    return __builtin_fmaf(d, d, -(r.a_filter * c));
}

// IntersectSceneSpheres over one LDS tile (Source/main.cpp:534-546), one lane
// per ray.  `filt` and `exact` hold `count` spheres padded to a multiple of
// kBlock; `base` is the list index of record 0.  Phase 1 (uniform, branch
// free): 32 early-out decisions per mask word.  Phase 2 (divergent, rare):
// candidates in list order with the reference's arithmetic.
template <int FILTER>
__device__ __forceinline__ void scan_tile(const Geom *__restrict__ filt, const Geom *__restrict__ exact,
                                          uint32_t count, uint32_t base, const RayState &r, float tmin,
                                          float &closest, int &best, WorkCount *wc = nullptr)
{
    for (uint32_t b = 0; b < count; b += kBlock) {
        uint32_t mask = 0u;
        if (FILTER == kFilterExact) {
            if (wc) // phase 1 is the reference's own discriminant for every sphere of the block
                wc->exact += min((uint32_t)kBlock, count - b);
#pragma unroll
            for (int j = 0; j < kBlock; ++j) {
                float B;
                const float disc = discriminant(r, exact[b + j], B);
                // the reference takes the hit branch unless disc < 0 (so also for NaN)
                mask = (mask << 1) | (disc < 0.0f ? 0u : 1u);
            }
        } else {
            uint32_t reject = 0u;
#pragma unroll
            for (int j = 0; j < kBlock; ++j) // shift the sign bit in from the right
                reject = __builtin_amdgcn_alignbit(reject, __float_as_uint(filter_sign(r, filt[b + j])), 31);
            mask = r.tame ? ~reject : 0xFFFFFFFFu;
        }
        if (b + kBlock > count) // padding records of the last block
            mask &= ~(0xFFFFFFFFu >> (count - b));
        while (mask) {
            const int j = __clz((int)mask);
            mask &= ~(0x80000000u >> j);
            if (resolve_candidate(r, exact[b + j], tmin, closest, FILTER == kFilterExact ? nullptr : wc))
                best = (int)(base + b + j);
            if (FILTER == kFilterExact && wc)
                ++wc->roots; // the discriminant was counted in phase 1
        }
    }
}

// ---------------------------------------------------------------------------
// Matrix-core scan: phase 1 on the FP32 MFMA pipe
// ---------------------------------------------------------------------------
//
// The conservative test is bilinear in (ray, sphere) once expanded about a
// common origin T (the scene's median centre): with o = O - T, c = C - T,
//     d  = (o - c).D            = [c.x c.y c.z 1] . [-D.x -D.y -D.z  o.D]
//     e  = -A' (|o - c|^2 - r2w) + m A' (|o|^2 + |c|^2)
//        = [c.x c.y c.z K] . [2A'o.x 2A'o.y 2A'o.z -A'] - A' |o|^2 (1 - m),
//     K  = |c|^2 (1 - m) - r2w,   A' = A (1 - 2^-17),  r2w = r^2 (1 + 2^-16),  m = 2^-18,
// and the sphere is a candidate unless s = d^2 + e < 0.  Both 4-term products
// are exactly what v_mfma_f32_16x16x4_f32 computes (a k-ordered fmaf chain,
// the ray constant entering as the C operand), for 16 spheres x 16 rays per
// instruction, on a pipe that runs beside the vector ALU.  The VALU is left
// with one fma and one alignbit per test.
//
// Error budget (u = 2^-24, L = |o| + |c|): the fused chains and the roundings of
// o, c, K, 2A'o and the constant perturb d^2 + e by at most
// u A (16.7 L^2 + 2.1 L^2 + 16.3 (|o|^2+|c|^2) + 6.1 r2w) <= 53.8 u A' (|o|^2+|c|^2) + 6.2 u A r2w,
// the reference accepts only if d*^2 - A (q* - r^2) >= -u A (16.7 q* + 2.1 r^2)
// (its own roundings, see filter_sign), and the widening adds
// 128 u A q* + 128 u A r^2 + 64 u A' (|o|^2 + |c|^2): every bracket stays
// positive, so a sphere the reference accepts is never rejected.  Preconditions:
// magnitudes below 2^28 after the shift, A and r^2 at least 2^-40; other rays
// are scanned by the exact in-order loop instead, other scenes never use this
// kernel.  Padding records are (0, 0, 0, K = 1e30): always rejected.
//
// Candidates of one ray come back spread over four lanes (the MFMA's row
// groups); one ds_bpermute round per 128 spheres brings them to the ray's own
// lane.  They are then resolved in whatever order the words hold them; the
// reference's in-order rule (accept root <= closest) equals "smallest root,
// then largest index", which is what resolve_unordered applies.

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGroup = 128; // spheres per candidate word set (8 MFMA row tiles)

struct MatrixOperands {
    float b1[4], b2[4]; // per 16-ray column tile t: this lane's B elements of the two products
    f32x4 c2[4];        // and the second product's C operand (the ray constant in all four rows)
};

// Publishes every lane's ray in matrix-operand form through the wave's LDS
// staging area (9 x 64 floats) and reads back this lane's B/C elements.
__device__ __forceinline__ void matrix_operands(const RayState &r, bool use, float tx, float ty, float tz,
                                                float *stage, int lane, MatrixOperands &m)
{
    float v[9];
    if (use) {
        const float ox = r.ox - tx, oy = r.oy - ty, oz = r.oz - tz;
        const float od = __builtin_fmaf(oz, r.dz, __builtin_fmaf(oy, r.dy, ox * r.dx));
        const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
        const float ap = r.a_filter;       // A (1 - 2^-17)
        const float ap2 = ap + ap;
        v[0] = -r.dx; v[1] = -r.dy; v[2] = -r.dz; v[3] = od;
        v[4] = ap2 * ox; v[5] = ap2 * oy; v[6] = ap2 * oz; v[7] = -ap;
        v[8] = -(ap * oo) * (1.0f - 0x1p-18f);
    } else { // idle or untame ray: d = 0, e = -1: every sphere rejected
#pragma unroll
        for (int q = 0; q < 8; ++q)
            v[q] = 0.0f;
        v[8] = -1.0f;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q)
        stage[q * 64 + lane] = v[q];
    __builtin_amdgcn_wave_barrier();
    const int c = lane & 15, k = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        m.b1[t] = stage[k * 64 + 16 * t + c];
        m.b2[t] = stage[(4 + k) * 64 + 16 * t + c];
        const float cc = stage[8 * 64 + 16 * t + c];
        m.c2[t] = f32x4{ cc, cc, cc, cc };
    }
    __builtin_amdgcn_wave_barrier();
}

// Whole-wave scan of one LDS tile.  `filt` holds {c.x, c.y, c.z, K} records and
// `exact` the reference's {C, r*r}, both padded to a multiple of kGroup.  Must
// be called by all 64 lanes; `mine` says whether this lane's own ray takes part.
__device__ __forceinline__ void scan_tile_matrix(const float *__restrict__ filt,
                                                 const Geom *__restrict__ exact, uint32_t count_padded,
                                                 uint32_t base, const MatrixOperands &m, const RayState &r,
                                                 bool mine, int lane, float tmin, float &closest, int &best,
                                                 WorkCount *wc = nullptr)
{
    const int c = lane & 15, k = lane >> 4;
    const f32x4 zero = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (uint32_t g0 = 0; g0 < count_padded; g0 += kGroup) {
        uint32_t reject[4] = { 0u, 0u, 0u, 0u };
        // Software pipeline, one stage deep: the two MFMAs of step i are issued, then the
        // four fma + alignbit pairs of step i-1 run on results that are already complete, so
        // the vector ALU never waits for the matrix pipe (they share the FP32 lanes: issue
        // slots, not latency, are what is scarce).  A[i = c][k] of the next row tile is
        // fetched one tile ahead (the last prefetch reads the array behind the tile; unused).
        float a_next = filt[(g0 + c) * 4 + k];
        f32x4 d_prev = zero, e_prev = zero;
#pragma unroll 1
        for (int st = 0; st < kGroup / 16; ++st) {
            const float a2 = a_next;
            a_next = filt[(g0 + 16 * (st + 1) + c) * 4 + k];
            const float a1 = (k == 3) ? 1.0f : a2;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, m.b1[t], zero, 0, 0, 0);
                const f32x4 e = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, m.b2[t], m.c2[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // the previous step's results: column tile t - 1 of this row tile, or column
                // tile 3 of the previous row tile; row 4k + q, column c
                const int tp = (t + 3) & 3;
                if (t > 0 || st > 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        reject[tp] = __builtin_amdgcn_alignbit(
                            reject[tp], __float_as_uint(__builtin_fmaf(d_prev[q], d_prev[q], e_prev[q])), 31);
                }
                __builtin_amdgcn_sched_barrier(0);
                d_prev = d;
                e_prev = e;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) // drain: column tile 3 of the last row tile
            reject[3] = __builtin_amdgcn_alignbit(
                reject[3], __float_as_uint(__builtin_fmaf(d_prev[q], d_prev[q], e_prev[q])), 31);
        // bit 31 - (4 st + q) of w[t]: sphere g0 + 16 st + 4k + q, ray 16 t + c
        const uint32_t w[4] = { ~reject[0], ~reject[1], ~reject[2], ~reject[3] };
        // bring the four row-group words of ray `lane` (column tile k, column c) to this lane
        uint32_t own[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tsel = (k - j) & 3; // the column tile whose owner asks this lane in round j
            const uint32_t pub = tsel == 0 ? w[0] : tsel == 1 ? w[1] : tsel == 2 ? w[2] : w[3];
            const int src = c + 16 * ((k + j) & 3);
            own[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(src * 4, (int)pub);
        }
        if (mine) {
            // one divergent loop for the four words: a lane with candidates in several words
            // does not make the wave pay one pass per word
            uint32_t o0 = own[0], o1 = own[1], o2 = own[2], o3 = own[3];
            while ((o0 | o1 | o2 | o3) != 0u) {
                const int j = o0 ? 0 : o1 ? 1 : o2 ? 2 : 3;
                const uint32_t word = o0 ? o0 : o1 ? o1 : o2 ? o2 : o3;
                const int pos = __clz((int)word);
                const uint32_t rest = word & ~(0x80000000u >> pos);
                if (j == 0) o0 = rest; else if (j == 1) o1 = rest; else if (j == 2) o2 = rest; else o3 = rest;
                const uint32_t i = g0 + 16u * (uint32_t)(pos >> 2) + 4u * (uint32_t)((k + j) & 3) + (uint32_t)(pos & 3);
                resolve_unordered_tame(r, exact[i], (int)(base + i), tmin, closest, best, wc); // (`mine` rays are tame)
            }
        }
    }
}

// Copies records [t0, t0+cnt_padded) of the filter's and of the exact geometry
// list into the LDS tile (`filt` may be null when phase 1 reads the exact list).
__device__ __forceinline__ void load_tile(Geom *tile_f, Geom *tile_e, const Geom *__restrict__ filt,
                                          const Geom *__restrict__ exact, uint32_t t0, uint32_t cnt_padded)
{
    if (tile_e) { // (null: the scan reads its exact records from memory)
        const float4 *src_e = reinterpret_cast<const float4 *>(exact + t0);
        float4 *dst_e = reinterpret_cast<float4 *>(tile_e);
        for (uint32_t i = threadIdx.x; i < cnt_padded; i += blockDim.x)
            dst_e[i] = src_e[i];
    }
    if (filt) {
        const float4 *src_f = reinterpret_cast<const float4 *>(filt + t0);
        float4 *dst_f = reinterpret_cast<float4 *>(tile_f);
        for (uint32_t i = threadIdx.x; i < cnt_padded; i += blockDim.x)
            dst_f[i] = src_f[i];
    }
}

// Tiles are padded to whole candidate groups (a multiple of kBlock as well).
__device__ __forceinline__ uint32_t round_up_group(uint32_t n)
{
    return (n + (kGroup - 1)) & ~(uint32_t)(kGroup - 1);
}


} // namespace
} // namespace rtamd
