// rt_kernels.hip -- gfx950 kernels for the per-pixel ray-tracing path.
//
// What the kernels compute is fixed by the reference, operation by operation
// (Source/main.cpp, cited per function); how they compute it is CDNA4-specific:
//
//   * one lane = one light path.  A wave keeps its 64 lanes busy by refilling
//     finished lanes from an 8x8-pixel chunk of (pixel, sample) items it pulls
//     from a global queue (ballot + mbcnt ranks, no per-lane atomics), so waves
//     never idle on short paths while long ones finish;
//   * a ray finds the spheres that need the reference's exact root evaluation through a
//     CONSERVATIVE first phase, which may pass a sphere the reference rejects but never the
//     reverse; whatever it passes is then resolved with the reference's exact, unfused
//     arithmetic and root selection, so results are bit-identical whichever phase 1 runs:
//       - a walk over a bounding-box hierarchy built on the host (boxes padded per ray by the
//         reference's own rounding slack; bvh_closest_hit), the default from 64 spheres up;
//       - the whole list, streamed through LDS tiles of {cx, cy, cz, r*r}, with a widened
//         discriminant test on the FP32 matrix cores (scan_tile_matrix) or with fused
//         multiply-adds on the vector ALU (filter_sign), collecting one bit per test;
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
#include "rt_device.h"

#include <limits.h>
#include <math.h>

namespace rtamd {
namespace {

// ---------------------------------------------------------------------------
// Counter-based random numbers
// ---------------------------------------------------------------------------

// Philox4x32-10 block for counter (c0, c1, c2, 0).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
    uint32_t c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32), not separate high and low halves
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// RandomFloat(), Source/main.cpp:87-92, as libstdc++ maps one 32-bit engine
// word to [0,1): float(w) / 2^32, and 1.0f replaced by the float below it.
__device__ __forceinline__ float unit_float(uint32_t w)
{
    return fminf((float)w * 0x1p-32f, 0.99999994f);
}

// RandomFloat(-1.0f, 1.0f), Source/main.cpp:94-97: -1 + 2*r evaluated in double, then
// narrowed.  2*r - 1 is exact in double for every float r in [0,1), so the narrowing is
// the only rounding -- which is exactly what one float fma computes.
__device__ __forceinline__ float range_m1_p1(uint32_t w)
{
    return __builtin_fmaf(2.0f, unit_float(w), -1.0f);
}

// ---------------------------------------------------------------------------
// Bytes
// ---------------------------------------------------------------------------

// static_cast<int>(float) as the reference's x86-64 build executes it (cvttss2si: truncate,
// INT_MIN for NaN and out-of-range), then glm::clamp to 0..255 (Source/main.cpp:117-133).
// INT_MIN clamps to 0, so the result is 0 for NaN, for v < 1 and for v >= 2^31, else
// min(trunc(v), 255).  v_cvt_u32_f32 truncates and saturates (NaN and negatives to 0, large
// values to 0xFFFFFFFF); read as signed, everything from 2^31 up is negative and clamps to 0.
__device__ __forceinline__ int to_byte(float v)
{
    uint32_t u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(v));
    return min(max((int)u, 0), 255);
}

// GetGradientColorAtRay + Lerp, Source/main.cpp:461-467, 157-160.
__device__ __forceinline__ void sky(float dy, int &r, int &g, int &b)
{
    const float t = dy * 1.8f;
    const float omt = 1.0f - t;
    r = to_byte(omt * 255.0f + t * 128.0f);
    g = to_byte(omt * 255.0f + t * 178.0f);
    b = to_byte(omt * 255.0f + t * 255.0f);
}

// EXTENSION (RT_FLAG_FLOAT_SHADING): colour limited to ToRGB's range but not truncated; NaN -> 0.
__device__ __forceinline__ float limit_255(float v)
{
    return v > 0.0f ? (v < 255.0f ? v : 255.0f) : 0.0f;
}

__device__ __forceinline__ void sky_float(float dy, float &r, float &g, float &b)
{
    const float t = dy * 1.8f;
    const float omt = 1.0f - t;
    r = limit_255(omt * 255.0f + t * 128.0f);
    g = limit_255(omt * 255.0f + t * 178.0f);
    b = limit_255(omt * 255.0f + t * 255.0f);
}

// ---------------------------------------------------------------------------
// Rays
// ---------------------------------------------------------------------------

struct RayState {
    float ox, oy, oz, dx, dy, dz;
    float four_a, two_a; // 4*dot(D,D), 2*dot(D,D)
    float a_filter;      // dot(D,D) * (1 - 2^-17), the conservative filter's A
    bool tame;           // all components finite and small enough for the filter's bound
};

// Magnitude limits under which the conservative filter's arithmetic can neither
// overflow nor lose its margin to underflow (see filter_sign).
constexpr float kTameMax = 0x1p28f;   // |coordinate|, |direction component|
constexpr float kTameMinA = 0x1p-40f; // dot(D,D)

__device__ __forceinline__ void ray_prepare(RayState &r)
{
    const float A = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz; // dot(): products summed left to right
    r.four_a = 4.0f * A;
    r.two_a = 2.0f * A;
    r.a_filter = __builtin_fmaf(-0x1p-17f, A, A);
    // the sum of magnitudes is NaN or Inf as soon as one component is, and comparisons with NaN
    // are false: one test covers "finite" and "small enough" (it is a little stricter than a test
    // per component, which only sends a few more rays to the exact scan)
    const float m = ((fabsf(r.ox) + fabsf(r.oy)) + (fabsf(r.oz) + fabsf(r.dx))) + (fabsf(r.dy) + fabsf(r.dz));
    r.tame = (m < kTameMax) && (A >= kTameMinA);
}

// Camera::GetRay, Source/main.cpp:440-444, for pixel (i, j) with jitter (ju, jv)
// in [0,1): u and v are formed in double and narrowed (Source/main.cpp:627-628).
__device__ __forceinline__ void camera_ray(const rt_camera &cam, uint32_t width, uint32_t height,
                                           int i, int j, double ju, double jv, RayState &r)
{
    const float u = (float)(((double)(float)i + ju) / (double)(float)width);
    const float v = (float)(((double)(float)j + jv) / (double)(float)height);
    r.ox = cam.origin[0]; r.oy = cam.origin[1]; r.oz = cam.origin[2];
    r.dx = ((cam.bottom_left[0] + (cam.horizontal[0] * u)) + (v * cam.vertical[0])) - cam.origin[0];
    r.dy = ((cam.bottom_left[1] + (cam.horizontal[1] * u)) + (v * cam.vertical[1])) - cam.origin[1];
    r.dz = ((cam.bottom_left[2] + (cam.horizontal[2] * u)) + (v * cam.vertical[2])) - cam.origin[2];
    ray_prepare(r);
}

// Discriminant of RaySphereIntersectionTest, Source/main.cpp:475-479, with the
// per-ray factors (A) and the per-sphere factor (R*R) computed once: the value
// is bit-identical because each hoisted factor is a single rounded product.
__device__ __forceinline__ float discriminant(const RayState &r, const Geom &s, float &B)
{
    const float ocx = r.ox - s.cx, ocy = r.oy - s.cy, ocz = r.oz - s.cz;
    const float d = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    B = d + d; // float(2.0 * d): doubling is exact
    const float C = (ocx * ocx + ocy * ocy + ocz * ocz) - s.r2;
    return B * B - r.four_a * C;
}

// Root selection of RaySphereIntersectionTest, Source/main.cpp:481-500, against
// the shrinking ClosestDistance of IntersectSceneSpheres (:539-545).  Returns
// true when the sphere becomes the closest hit (an equal T is accepted, so the
// later sphere wins a tie).
__device__ __forceinline__ bool resolve_candidate(const RayState &r, const Geom &s, float tmin,
                                                  float &closest)
{
    float B;
    const float disc = discriminant(r, s, B);
    if (disc < 0.0f)
        return false;
    const float sq = sqrtf(disc);
    float root = (-B - sq) / r.two_a;
    if (root < tmin || root > closest) {
        root = (-B + sq) / r.two_a;
        if (root < tmin || root > closest)
            return false;
    }
    closest = root;
    return true;
}

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
                                          float &closest, int &best)
{
    for (uint32_t b = 0; b < count; b += kBlock) {
        uint32_t mask = 0u;
        if (FILTER == kFilterExact) {
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
            if (resolve_candidate(r, exact[b + j], tmin, closest))
                best = (int)(base + b + j);
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

// Effective root of RaySphereIntersectionTest (Source/main.cpp:490-499) for a
// ray without NaNs: the near root if it is >= tmin, else the far root.
__device__ __forceinline__ void resolve_unordered(const RayState &r, const Geom &s, int idx, float tmin,
                                                  float &closest, int &best)
{
    float B;
    const float disc = discriminant(r, s, B);
    if (disc < 0.0f)
        return;
    const float sq = sqrtf(disc);
    float root = (-B - sq) / r.two_a;
    if (root < tmin)
        root = (-B + sq) / r.two_a;
    if (root >= tmin && (root < closest || (root == closest && idx > best))) {
        closest = root;
        best = idx;
    }
}

// First half of resolve_unordered for a leaf record: the discriminant, and whether the root
// evaluation can be skipped.  Besides the reference's own `Discriminant < 0` early-out, a sphere
// with B > 0 and C > 0 (origin outside, centre behind) is dropped without a square root: then
// 4AC > 0, so fl(B*B - 4AC) <= fl(B*B), its correctly rounded root is at most sqrt(fl(B*B)) = |B|,
// both numerators -B -+ sq are <= 0 and the reference rejects both roots against tmin > 0.
__device__ __forceinline__ bool leaf_candidate(const RayState &r, const Geom &s, float &B, float &disc)
{
    const float ocx = r.ox - s.cx, ocy = r.oy - s.cy, ocz = r.oz - s.cz;
    const float d = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    B = d + d;
    const float C = (ocx * ocx + ocy * ocy + ocz * ocz) - s.r2;
    disc = B * B - r.four_a * C;
    return !(disc < 0.0f) && !(B > 0.0f && C > 0.0f);
}

// Second half: the effective root (Source/main.cpp:490-499) and the closest-hit update; the
// list index is fetched only when the root is in range.
__device__ __forceinline__ void leaf_root(const RayState &r, float B, float disc, const uint32_t *index, float tmin,
                                          float &closest, int &best)
{
    const float sq = sqrtf(disc);
    float root = (-B - sq) / r.two_a;
    if (root < tmin)
        root = (-B + sq) / r.two_a;
    if (root >= tmin && root <= closest) {
        const int idx = (int)*index;
        if (root < closest || idx > best) {
            closest = root;
            best = idx;
        }
    }
}

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
                                                 bool mine, int lane, float tmin, float &closest, int &best)
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
                resolve_unordered(r, exact[i], (int)(base + i), tmin, closest, best);
            }
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

// Start of a ray's walk: the always-tested spheres, then the root.
__device__ __forceinline__ void bvh_begin(const BvhArgs &b, const RayState &r, float tmin, int &node, uint32_t &sp,
                                          float &closest, int &best)
{
    closest = INFINITY;
    best = -1;
    for (uint32_t i = 0; i < b.n_large; ++i)
        resolve_unordered(r, b.leaf_geom[i], (int)b.leaf_index[i], tmin, closest, best);
    node = b.n_nodes ? 0 : kBvhDone;
    sp = 0u;
}

// Must be called by all 64 lanes; `walking` lanes hold a tame ray whose node is not kBvhDone.
template <bool NODES_IN_LDS>
__device__ __forceinline__ void bvh_walk(const BvhArgs &b, const float4 *lds_nodes, const RayState &r,
                                         uint32_t *stk, bool walking, float tmin, int &node, uint32_t &sp,
                                         float &closest, int &best)
{
    if (!walking)
        node = kBvhDone;
    const uint32_t n_start = (uint32_t)__popcll(__ballot(walking));
    if (n_start == 0u)
        return;
    const uint32_t suspend = n_start > b.suspend_lanes ? b.suspend_lanes : 0u;

    const float ox = r.ox - b.center[0], oy = r.oy - b.center[1], oz = r.oz - b.center[2];
    // reach^2 = (|o| + radius)^2 <= 2 (|o|^2 + radius^2); the hardware square root (1 ulp) is
    // inside the 1.001.  Since reff^2 >= 2^-18 reach^2, the slab roundings' 40u reach is below
    // 2^-10 reff.
    const float reach2 = 2.0f * __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, __builtin_fmaf(ox, ox, b.radius2)));
    const float reff = __builtin_amdgcn_sqrtf(__builtin_fmaf(reach2, 0x1p-18f * (1.0f + 0x1p-10f), b.rmin2w));
    const float pad = __builtin_fmaf(reff, 0x1.01p-10f, reff - b.rmin) * 1.001f;

    const float dmax = fmaxf(fmaxf(fabsf(r.dx), fabsf(r.dy)), fabsf(r.dz));
    const float dlim = dmax * 0x1p-50f;
    const float ddx = fabsf(r.dx) < dlim ? copysignf(dlim, r.dx) : r.dx;
    const float ddy = fabsf(r.dy) < dlim ? copysignf(dlim, r.dy) : r.dy;
    const float ddz = fabsf(r.dz) < dlim ? copysignf(dlim, r.dz) : r.dz;
    const float ix = __builtin_amdgcn_rcpf(ddx), iy = __builtin_amdgcn_rcpf(ddy), iz = __builtin_amdgcn_rcpf(ddz);
    // t at plane p: (p - o) / d = fma(p, i, n); lower planes are moved by -pad, upper by +pad
    const float nlx = -((ox + pad) * ix), nly = -((oy + pad) * iy), nlz = -((oz + pad) * iz);
    const float nhx = -((ox - pad) * ix), nhy = -((oy - pad) * iy), nhz = -((oz - pad) * iz);
    // Which of the two planes of an axis the ray meets first depends on the sign of d.  Instead
    // of min/max per box (half-rate instructions), each t is fma(lo, a, fma(hi, b, c)) with one
    // of a, b zero: the same single rounding, on full-rate instructions only.
    const bool px = ix >= 0.0f, py = iy >= 0.0f, pz = iz >= 0.0f;
    // entry plane: fma(lo, na, fma(hi, nb, nc)); exit plane: the same with na and nb swapped
    const float nax = px ? ix : 0.0f, nbx = ix - nax, ncx = px ? nlx : nhx, fcx = px ? nhx : nlx;
    const float nay = py ? iy : 0.0f, nby = iy - nay, ncy = py ? nly : nhy, fcy = py ? nhy : nly;
    const float naz = pz ? iz : 0.0f, nbz = iz - naz, ncz = pz ? nlz : nhz, fcz = pz ? nhz : nlz;
    const float fax = nbx, fbx = nax, fay = nby, fby = nay, faz = nbz, fbz = naz;
    const float t_lo = tmin * (1.0f - 0x1p-20f);
    float t_hi = closest * (1.0f + 0x1p-20f);

    constexpr int kDone = kBvhDone;
    while ((uint32_t)__popcll(__ballot(node != kDone)) > suspend) {
        while (node >= 0) {
            const float4 *np = (NODES_IN_LDS ? lds_nodes : b.nodes) + (size_t)node * 4u;
            const float4 l0 = np[0], h0 = np[1], l1 = np[2], h1 = np[3];
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
                stk[sp * kTraceThreads] = (uint32_t)(first0 ? c1 : c0);
                ++sp;
                node = first0 ? c0 : c1;
            } else if (hit0) {
                node = c0;
            } else if (hit1) {
                node = c1;
            } else if (sp == 0u) {
                node = kDone;
            } else {
                --sp;
                node = (int)stk[sp * kTraceThreads];
            }
        }
        if (node == kDone)
            continue;
        const uint32_t code = (uint32_t)~node, first = code >> 4, count = code & 15u;
        for (uint32_t i0 = 0u; i0 < count; i0 += 4u) {
            // up to four records in flight before the first is used; then every lane walks its
            // own candidates, so the wave pays for the most candidates any lane has, not for
            // every record some lane hits
            Geom g[4];
#pragma unroll
            for (uint32_t j = 0u; j < 4u; ++j)
                if (i0 + j < count)
                    g[j] = b.leaf_geom[first + i0 + j];
            float Bj[4], dj[4];
            uint32_t cand = 0u;
#pragma unroll
            for (uint32_t j = 0u; j < 4u; ++j) {
                Bj[j] = dj[j] = 0.0f;
                if (i0 + j < count && leaf_candidate(r, g[j], Bj[j], dj[j]))
                    cand |= 1u << j;
            }
            while (cand != 0u) {
                const uint32_t j = (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1u;
                const float Bs = j == 0u ? Bj[0] : j == 1u ? Bj[1] : j == 2u ? Bj[2] : Bj[3];
                const float ds = j == 0u ? dj[0] : j == 1u ? dj[1] : j == 2u ? dj[2] : dj[3];
                leaf_root(r, Bs, ds, b.leaf_index + (first + i0 + j), tmin, closest, best);
            }
        }
        t_hi = closest * (1.0f + 0x1p-20f);
        if (sp == 0u) {
            node = kDone;
        } else {
            --sp;
            node = (int)stk[sp * kTraceThreads];
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
            bvh_begin(b, r, 0.001f, node, sp, closest, best);
        else
            scan_tile<kFilterExact>(exact, exact, n, 0u, r, 0.001f, closest, best);
    }
    bvh_walk<false>(b, nullptr, r, stk, have && node != kBvhDone, 0.001f, node, sp, closest, best);
}

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

// Hit point and normal, Source/main.cpp:503-514.
__device__ __forceinline__ void hit_frame(const RayState &r, float t, float cx, float cy, float cz,
                                          float radius, float &px, float &py, float &pz, float &nx,
                                          float &ny, float &nz, bool &inside)
{
    px = r.ox + (r.dx * t); py = r.oy + (r.dy * t); pz = r.oz + (r.dz * t);
    nx = (px - cx) / radius; ny = (py - cy) / radius; nz = (pz - cz) / radius;
    inside = (r.dx * nx + r.dy * ny + r.dz * nz) > 0.0f;
    if (inside) {
        nx = -nx; ny = -ny; nz = -nz;
    }
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m);

// GeneratePointInUnitSphere (Source/main.cpp:398-414) for the lanes of a wave that `want` a point,
// one Philox block per attempt, with the lanes helping
// each other.  A lane's attempts are the Philox blocks call, call + 1, ... of its own (pixel,
// sample) stream, and it must take the FIRST one inside the sphere; which lane evaluates an
// attempt does not matter.  Every lane tries its attempt 0 itself (52 % succeed).  The rest publish
// their stream in LDS, and in each further round all 64 lanes evaluate `per` consecutive attempts
// for each of them (per = the power of two that fits, at most 16); a lane takes its first success
// and advances its counter past it, or past all `per` attempts if none succeeded.  Two rounds
// usually do where the plain loop makes the wave wait for its unluckiest lane (five or six
// attempts).  Must be called by all 64 lanes; `coop` = 192 words of LDS owned by the wave.
__device__ __forceinline__ void unit_sphere_points_wave(bool want, uint32_t pixel, uint32_t sample, uint32_t &call,
                                                        uint32_t k0, uint32_t k1, uint32_t *coop, int lane,
                                                        float &x, float &y, float &z)
{
    bool need = want;
    if (need) {
        uint32_t w[4];
        philox4x32_10(pixel, sample, call++, k0, k1, w);
        x = range_m1_p1(w[0]);
        y = range_m1_p1(w[1]);
        z = range_m1_p1(w[2]);
        need = !(((x * x) + (y * y) + (z * z)) < 1.0f);
    }
    unsigned long long nm = __ballot(need);
    while (nm != 0ull) {
        const uint32_t n = (uint32_t)__popcll(nm);
        uint32_t lg = 31u - (uint32_t)__builtin_clz(64u / n);
        lg = lg > 4u ? 4u : lg;
        const uint32_t per = 1u << lg;
        const uint32_t k = lane_rank(nm);
        if (need) {
            coop[k] = pixel;
            coop[64u + k] = sample;
            coop[128u + k] = call;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t t = (uint32_t)lane >> lg, at = (uint32_t)lane & (per - 1u);
        float hx = 0.0f, hy = 0.0f, hz = 0.0f;
        bool ok = false;
        if (t < n) {
            uint32_t w[4];
            philox4x32_10(coop[t], coop[64u + t], coop[128u + t] + at, k0, k1, w);
            hx = range_m1_p1(w[0]);
            hy = range_m1_p1(w[1]);
            hz = range_m1_p1(w[2]);
            ok = ((hx * hx) + (hy * hy) + (hz * hz)) < 1.0f;
        }
        const unsigned long long okm = __ballot(ok);
        __builtin_amdgcn_wave_barrier();
        uint32_t src = (uint32_t)lane;
        bool got = false;
        if (need) {
            const uint32_t bits = (uint32_t)(okm >> (k << lg)) & ((1u << per) - 1u);
            if (bits != 0u) {
                const uint32_t first = (uint32_t)__builtin_ctz(bits);
                src = (k << lg) + first;
                call += first + 1u;
                got = true;
                need = false;
            } else {
                call += per;
            }
        }
        const float gx = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src * 4u), __float_as_int(hx)));
        const float gy = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src * 4u), __float_as_int(hy)));
        const float gz = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src * 4u), __float_as_int(hz)));
        if (got) {
            x = gx; y = gy; z = gz;
        }
        nm = __ballot(need);
    }
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m)
{
    // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Copies records [t0, t0+cnt_padded) of the filter's and of the exact geometry
// list into the LDS tile (`filt` may be null when phase 1 reads the exact list).
__device__ __forceinline__ void load_tile(Geom *tile_f, Geom *tile_e, const Geom *__restrict__ filt,
                                          const Geom *__restrict__ exact, uint32_t t0, uint32_t cnt_padded)
{
    const float4 *src_e = reinterpret_cast<const float4 *>(exact + t0);
    float4 *dst_e = reinterpret_cast<float4 *>(tile_e);
    for (uint32_t i = threadIdx.x; i < cnt_padded; i += blockDim.x)
        dst_e[i] = src_e[i];
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

// ---------------------------------------------------------------------------
// The path kernel: TraceThreadFunction's sample loop + GetRayColor
// (Source/main.cpp:551-598, 613-645), iterative.
// ---------------------------------------------------------------------------

template <bool MULTI_TILE, int FILTER, bool EXT>
__global__ void __launch_bounds__(kTraceThreads, 4) trace_kernel(const TraceArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    Geom *tile_f = reinterpret_cast<Geom *>(smem);
    Geom *tile_e = tile_f + a.tile_cap;
    // matrix scan: 4 waves x 9 x 64 floats of operand staging; hierarchy scan: the walk's
    // stack, levels x 256 node references
    float *stage = reinterpret_cast<float *>(tile_e + a.tile_cap);
    uint32_t *stack = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(stage) + a.aux_bytes);
    const int stack_levels = a.max_depth > 0 ? a.max_depth : 1;
    uint32_t *wave_flags = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(stack) +
                                                        (size_t)stack_levels * kTraceThreads * a.stack_entry_bytes);
    uint32_t *acc_base = wave_flags + 4; // 4 waves x 2 chunk slots x 64 pixels x 3 channels
    const Geom *filt_src = FILTER == kFilterExact ? nullptr : a.geom;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t n_pad = round_up_group(a.n);

    if (!MULTI_TILE && FILTER != kFilterBvh) {
        load_tile(tile_f, tile_e, filt_src, a.exact, 0u, n_pad);
        __syncthreads();
    }
    // hierarchy scan: the nodes of a small tree live in LDS, behind the walk's stack
    float4 *bvh_lds = reinterpret_cast<float4 *>(stage) + (size_t)a.bvh.levels * (kTraceThreads / 4);
    // 192 words per wave for unit_sphere_points_wave: the head of the wave's own operand staging
    // slice in the list scans (idle while paths are shaded), or a region behind the hierarchy's nodes
    uint32_t *coop = FILTER == kFilterBvh
                         ? reinterpret_cast<uint32_t *>(bvh_lds + (size_t)a.bvh.lds_nodes * 4u) + (threadIdx.x >> 6) * 192
                         : reinterpret_cast<uint32_t *>(stage) + (threadIdx.x >> 6) * (9 * 64);
    if (FILTER == kFilterBvh && a.bvh.lds_nodes) {
        for (uint32_t i = threadIdx.x; i < a.bvh.lds_nodes * 4u; i += blockDim.x)
            bvh_lds[i] = a.bvh.nodes[i];
        __syncthreads();
    }

    // Per-wave colour sums of the (at most two) chunks the wave has paths in flight for.  A
    // chunk's 64 pixels belong to this wave alone during the launch, so its sums are added
    // to the frame once, with plain loads and stores, when its last path has terminated.
    uint32_t *acc = acc_base + wave * (2 * 64 * 3);
    for (int i = lane; i < 2 * 64 * 3; i += 64)
        acc[i] = 0u;

    // wave-uniform queue state; slot s is free while s_total[s] == 0
    uint32_t w_next = 0u, w_end = 0u;
    int cur = 0;
    uint32_t s_total0 = 0u, s_total1 = 0u, s_done0 = 0u, s_done1 = 0u;
    int s_x0[2] = { 0, 0 }, s_y0[2] = { 0, 0 }, s_tw[2] = { 1, 1 }, s_npx[2] = { 1, 1 };
    bool queue_empty = (a.sample_count == 0u);

    // per-lane path state
    bool active = false;
    RayState ray = {};
    uint32_t pixel = 0u, sample = 0u, call = 0u;
    uint32_t acc_at = 0u; // index of this path's pixel in acc: slot * 192 + pixel-in-chunk * 3
    int depth = 0, level = 0;
    uint32_t n_rays = 0u;
    uint32_t n_rounds = 0u; // wave-uniform: passes over the sphere list
    // hierarchy scan: state of a walk that continues in the next pass
    int w_node = kBvhDone, w_best = -1;
    uint32_t w_sp = 0u;
    float w_closest = INFINITY;
    bool w_pending = false;

    for (;;) {
        // ---- refill idle lanes from the wave's chunk, pulling new chunks as needed
        unsigned long long need = __ballot(!active);
        while (need != 0ull && !queue_empty) {
            if (w_next >= w_end) {
                // a new chunk needs a free accumulator slot; with both in use the idle lanes
                // wait until one of the two chunks in flight has drained
                const int fs = s_total0 == 0u ? 0 : (s_total1 == 0u ? 1 : -1);
                if (fs < 0)
                    break;
                uint32_t c = 0u;
                if (lane == 0)
                    c = atomicAdd(a.chunk_counter, 1u);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= a.n_chunks) {
                    queue_empty = true;
                    break;
                }
                const uint32_t t = a.tile_first + c * a.tile_stride;
                const uint32_t ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
                const int x0 = a.x0 + (int)tx * kTileW, y0 = a.y0 + (int)ty * kTileH;
                const int tw = min(kTileW, a.x0 + a.w - x0), th = min(kTileH, a.y0 + a.h - y0);
                cur = fs;
                s_x0[fs] = x0; s_y0[fs] = y0; s_tw[fs] = tw; s_npx[fs] = tw * th;
                w_next = 0u;
                w_end = (uint32_t)(tw * th) * a.sample_count;
                if (fs == 0) { s_total0 = w_end; s_done0 = 0u; } else { s_total1 = w_end; s_done1 = 0u; }
            }
            const int ch_x0 = s_x0[cur], ch_y0 = s_y0[cur], ch_tw = s_tw[cur], ch_npx = s_npx[cur];
            const uint32_t avail = w_end - w_next;
            const uint32_t rank = lane_rank(need);
            const bool got = ((need >> lane) & 1ull) && rank < avail;
            if (got) {
                const uint32_t item = w_next + rank;
                const uint32_t s = item / (uint32_t)ch_npx;
                const uint32_t p = item - s * (uint32_t)ch_npx;
                const int ly = (int)(p / (uint32_t)ch_tw);
                const int lx = (int)p - ly * ch_tw;
                const int px = ch_x0 + lx, py = ch_y0 + ly;
                pixel = (uint32_t)px + (uint32_t)py * a.width;
                sample = a.sample_begin + s;
                acc_at = (uint32_t)cur * 192u + p * 3u;
                call = 0u;
                uint32_t w[4];
                philox4x32_10(pixel, sample, call++, a.key0, a.key1, w);
                camera_ray(a.cam, a.width, a.height, px, py, (double)unit_float(w[0]),
                           (double)unit_float(w[1]), ray);
                depth = a.max_depth; // >= 1: the host does not launch for RAY_DEPTH <= 0
                level = 0;
                active = true;
            }
            const uint32_t want = (uint32_t)__popcll(need);
            w_next += want < avail ? want : avail;
            need &= ~__ballot(got);
        }

        // ---- termination
        const bool wave_active = __any(active);
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

        // ---- IntersectSceneSpheres(ray, 0.001f, inf), Source/main.cpp:562
        float closest = INFINITY;
        int best = -1;
        MatrixOperands mo;
        const bool mine = active && ray.tame; // this lane's ray goes through the matrix filter
        if (FILTER == kFilterMatrix && wave_active)
            matrix_operands(ray, mine, a.shift[0], a.shift[1], a.shift[2], stage + wave * (9 * 64), lane, mo);
        bool shade = active; // the ray's closest hit is known after this pass
        if (FILTER == kFilterBvh) {
            if (active && !w_pending) { // a ray that starts its scan in this pass
                w_pending = true;
                if (ray.tame) {
                    bvh_begin(a.bvh, ray, 0.001f, w_node, w_sp, w_closest, w_best);
                } else { // the reference's in-order loop over the whole list, from memory
                    w_closest = INFINITY;
                    w_best = -1;
                    w_node = kBvhDone;
                    scan_tile<kFilterExact>(a.exact, a.exact, a.n, 0u, ray, 0.001f, w_closest, w_best);
                }
            }
            if (wave_active) {
                uint32_t *stk = reinterpret_cast<uint32_t *>(stage) + tid;
                const bool walking = active && w_node != kBvhDone;
                if (a.bvh.lds_nodes)
                    bvh_walk<true>(a.bvh, bvh_lds, ray, stk, walking, 0.001f, w_node, w_sp, w_closest, w_best);
                else
                    bvh_walk<false>(a.bvh, nullptr, ray, stk, walking, 0.001f, w_node, w_sp, w_closest, w_best);
            }
            if (active && w_node == kBvhDone) {
                w_pending = false;
                closest = w_closest;
                best = w_best;
            } else {
                shade = false; // still walking: continues in the next pass
            }
        }
        for (uint32_t t0 = 0u; FILTER != kFilterBvh && t0 < (MULTI_TILE ? a.n : 1u); t0 += a.tile_cap) {
            const uint32_t cnt = MULTI_TILE ? min(a.tile_cap, a.n - t0) : a.n;
            if (MULTI_TILE) {
                __syncthreads(); // everyone is done with the previous tile (and with wave_flags)
                load_tile(tile_f, tile_e, filt_src, a.exact, t0, round_up_group(cnt));
                __syncthreads();
            }
            if (FILTER == kFilterMatrix) {
                if (wave_active)
                    scan_tile_matrix(reinterpret_cast<const float *>(tile_f), tile_e, round_up_group(cnt), t0, mo,
                                     ray, mine, lane, 0.001f, closest, best);
                if (active && !ray.tame) // NaN, Inf, zero or huge rays: the reference's in-order loop
                    scan_tile<kFilterExact>(tile_f, tile_e, cnt, t0, ray, 0.001f, closest, best);
            } else if (active) {
                scan_tile<FILTER>(tile_f, tile_e, cnt, t0, ray, 0.001f, closest, best);
            }
        }
        if (wave_active)
            ++n_rounds;

        // ---- the random point of the Diffuse and Metal branches (:566, :584), drawn for the
        // whole wave at once; nothing else draws between the hit and this point
        int material = -1;
        if (shade && best >= 0)
            material = a.spheres[best].material;
        const bool wants_point = material == RT_MAT_DIFFUSE || material == RT_MAT_METAL;
        float rx = 0.0f, ry = 0.0f, rz = 0.0f;
        if (__any(wants_point))
            unit_sphere_points_wave(wants_point, pixel, sample, call, a.key0, a.key1, coop, lane, rx, ry, rz);

        // ---- GetRayColor's branches, Source/main.cpp:562-597
        bool terminated = false;
        if (shade) {
            ++n_rays;
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
                    hit_frame(ray, closest, sp->center[0], sp->center[1], sp->center[2], sp->radius,
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
                    push_bounce(stack, a.stack_entry_bytes, level * kTraceThreads + tid, (uint32_t)best);
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
                    hit_frame(ray, closest, sp->center[0], sp->center[1], sp->center[2], sp->radius,
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
                    push_bounce(stack, a.stack_entry_bytes, level * kTraceThreads + tid, (uint32_t)best);
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
                    const rt_sphere *sp = a.spheres + read_bounce(stack, a.stack_entry_bytes, l * kTraceThreads + tid);
                    if (sp->material == RT_MAT_DIFFUSE) {
                        fr = fr * 0.5f; fg = fg * 0.5f; fb = fb * 0.5f;
                    }
                    fr = limit_255(sp->color[0] * fr);
                    fg = limit_255(sp->color[1] * fg);
                    fb = limit_255(sp->color[2] * fb);
                }
                atomicAdd(acc + acc_at + 0u, (uint32_t)to_byte(fr));
                atomicAdd(acc + acc_at + 1u, (uint32_t)to_byte(fg));
                atomicAdd(acc + acc_at + 2u, (uint32_t)to_byte(fb));
                active = false;
            } else if (terminated) {
                // unwind the recursion: per level, bytes -> float -> bytes (:569-578, :587-591)
                for (int l = level - 1; l >= 0; --l) {
                    const rt_sphere *sp = a.spheres + read_bounce(stack, a.stack_entry_bytes, l * kTraceThreads + tid);
                    if (sp->material == RT_MAT_DIFFUSE) {
                        cr >>= 1; cg >>= 1; cb >>= 1;
                    }
                    cr = to_byte(sp->color[0] * (float)cr);
                    cg = to_byte(sp->color[1] * (float)cg);
                    cb = to_byte(sp->color[2] * (float)cb);
                }
                // FinalColor += ray_color, :633-635
                atomicAdd(acc + acc_at + 0u, (uint32_t)cr);
                atomicAdd(acc + acc_at + 1u, (uint32_t)cg);
                atomicAdd(acc + acc_at + 2u, (uint32_t)cb);
                active = false;
            }
        }

        // ---- chunks whose last path has just terminated: add their sums to the frame
        const unsigned long long term = __ballot(terminated);
        if (term != 0ull) {
            const unsigned long long term1 = __ballot(terminated && acc_at >= 192u);
            s_done1 += (uint32_t)__popcll(term1);
            s_done0 += (uint32_t)__popcll(term & ~term1);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const uint32_t total = sl ? s_total1 : s_total0, done = sl ? s_done1 : s_done0;
                if (total != 0u && done == total) {
                    if (lane < s_npx[sl]) {
                        const int ly = lane / s_tw[sl], lx = lane - ly * s_tw[sl];
                        uint32_t *dst = a.sums + ((size_t)(s_x0[sl] + lx) + (size_t)(s_y0[sl] + ly) * a.width) * 3u;
                        uint32_t *src = acc + sl * 192 + lane * 3;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            dst[ch] += src[ch];
                            src[ch] = 0u;
                        }
                    }
                    if (sl) s_total1 = 0u; else s_total0 = 0u;
                }
            }
        }
    }

    // ---- ray counter: one atomic per wave
    unsigned long long total = n_rays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        total += __shfl_down(total, off, 64);
    if (lane == 0 && total != 0ull) {
        atomicAdd(a.rays, total);
        atomicAdd(a.rays + 1, (unsigned long long)n_rounds);
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
        camera_ray(a.cam, a.width, a.height, px, py, 0.5, 0.5, ray);

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
// Launch wrappers
// ---------------------------------------------------------------------------

uint32_t trace_aux_bytes(int filter, uint32_t bvh_levels, uint32_t bvh_lds_nodes)
{
    if (filter == kFilterBvh)
        return (bvh_levels ? bvh_levels : 1u) * kTraceThreads * (uint32_t)sizeof(uint32_t) + bvh_lds_nodes * 64u +
               4u * 192u * (uint32_t)sizeof(uint32_t);
    return 4u * 9u * 64u * (uint32_t)sizeof(float);
}

size_t trace_lds_bytes(uint32_t tile_cap, uint32_t aux_bytes, int32_t max_depth, uint32_t stack_entry_bytes)
{
    const size_t levels = max_depth > 0 ? (size_t)max_depth : 1u;
    return (size_t)tile_cap * 2u * sizeof(Geom) + aux_bytes +
           levels * kTraceThreads * stack_entry_bytes + 4u * sizeof(uint32_t) +
           4u * 2u * 64u * 3u * sizeof(uint32_t);
}

hipError_t launch_trace(const TraceArgs &a, uint32_t n_workgroups, hipStream_t stream)
{
    const size_t lds = trace_lds_bytes(a.tile_cap, a.aux_bytes, a.max_depth, a.stack_entry_bytes);
    const bool multi = a.filter != kFilterBvh && a.n > a.tile_cap;
    const dim3 grid(n_workgroups), block(kTraceThreads);
    const bool glass = (a.flags & (RT_FLAG_DIELECTRIC | RT_FLAG_FLOAT_SHADING)) != 0u; // the extended kernels
#define RT_LAUNCH_TRACE(F)                                                                        \
    do {                                                                                          \
        if (multi && glass)                                                                       \
            hipLaunchKernelGGL((trace_kernel<true, F, true>), grid, block, lds, stream, a);       \
        else if (multi)                                                                           \
            hipLaunchKernelGGL((trace_kernel<true, F, false>), grid, block, lds, stream, a);      \
        else if (glass)                                                                           \
            hipLaunchKernelGGL((trace_kernel<false, F, true>), grid, block, lds, stream, a);      \
        else                                                                                      \
            hipLaunchKernelGGL((trace_kernel<false, F, false>), grid, block, lds, stream, a);     \
    } while (0)
    switch (a.filter) {
    case kFilterExact: RT_LAUNCH_TRACE(kFilterExact); break;
    case kFilterConservative: RT_LAUNCH_TRACE(kFilterConservative); break;
    case kFilterMatrix: RT_LAUNCH_TRACE(kFilterMatrix); break;
    case kFilterBvh:
        if (glass)
            hipLaunchKernelGGL((trace_kernel<false, kFilterBvh, true>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((trace_kernel<false, kFilterBvh, false>), grid, block, lds, stream, a);
        break;
    default:
        return hipErrorInvalidValue;
    }
#undef RT_LAUNCH_TRACE
    return hipGetLastError();
}

hipError_t trace_occupancy(int filter, size_t lds_bytes, int *workgroups_per_cu)
{
    if (filter == kFilterBvh)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(
            workgroups_per_cu, trace_kernel<false, kFilterBvh, false>, kTraceThreads, lds_bytes);
    if (filter == kFilterMatrix)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(
            workgroups_per_cu, trace_kernel<false, kFilterMatrix, false>, kTraceThreads, lds_bytes);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(
        workgroups_per_cu, trace_kernel<false, kFilterConservative, false>, kTraceThreads, lds_bytes);
}

hipError_t launch_resolve(const ResolveArgs &a, hipStream_t stream)
{
    const uint32_t total = (uint32_t)a.w * (uint32_t)a.h;
    if (total == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((total + 255u) / 256u), dim3(256), 0, stream, a);
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
    if (a.n_rays == 0u)
        return hipSuccess;
    const uint32_t cap = scan_tile_cap(a.n);
    const size_t lds = a.use_bvh ? (size_t)a.bvh.levels * 256u * sizeof(uint32_t) : (size_t)cap * 2u * sizeof(Geom);
    hipLaunchKernelGGL(intersect_kernel, dim3((a.n_rays + 255u) / 256u), dim3(256), lds, stream, a, cap);
    return hipGetLastError();
}

} // namespace rtamd
