// rt_arith.h -- the reference's arithmetic, operation by operation (Source/main.cpp, cited per
// function): random words to floats, bytes, rays, the discriminant and the root selection of
// RaySphereIntersectionTest, the hit frame.  Device code; part of rt_kernels.hip's translation unit.
//
// Everything here relies on -ffp-contract=off: every float multiply and add rounds on its own, as
// in the reference's x86-64 build; sqrt and divide are IEEE.
#pragma once

#include <limits.h>
#include <math.h>

#include "rt_device.h"

namespace rtamd {
namespace {

// The lanes for which `p` holds, as a mask: the compare that produced `p` already left exactly this
// in a scalar register pair, and the builtin hands it over.  (HIP's __ballot(int) and __any(int) first
// turn the condition into a 0/1 integer per lane and compare that again: two half-rate vector
// instructions per vote, a dozen votes per pass.)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Per-lane tally of the work a scan executes, kept only by the counting build of the trace kernel
// (RT_OPT_COUNT_WORK); everywhere else the pointer is a compile-time null and the code vanishes.
struct WorkCount {
    uint32_t nodes; // inner nodes of the hierarchy visited (two box tests each)
    uint32_t exact; // evaluations of the reference's discriminant (Source/main.cpp:475-479)
    uint32_t roots; // of those, evaluations that went on to the square root and the division (:490-499)
    // Per section of the kernel (kSec*): how often a wave executed it, and with how many lanes enabled.
    uint32_t sec_waves[12], sec_lanes[12];
    // Shader-clock cycles a wave spent in a phase of its passes (kTime*), summed over its lifetime.
    unsigned long long time[8];
};

enum { kTimeLife = 0, kTimeFetch, kTimeRefill, kTimeScan, kTimePoints, kTimeShade, kTimeFlush, kTimeCount };

// Adds the cycles since `since` to phase `i` and restarts the clock (counting build only).
#define RT_CLOCK(wc, i, since)                                         \
    do {                                                               \
        if (wc) {                                                      \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            (wc)->time[i] += now_ - (since);                           \
            (since) = now_;                                            \
        }                                                              \
    } while (0)

enum { kSecPass = 0, kSecRefill, kSecWalkBegin, kSecNode, kSecLeafGroup, kSecRoot, kSecPointFirst, kSecPointHelp,
       kSecScatter, kSecReplayLevel, kSecWalkSetup, kSecShade, kSecCount };

// Tallies one execution of section `i` by the lanes that are enabled here.
#define RT_TALLY(wc, i)                                                                             \
    do {                                                                                            \
        if (wc) {                                                                                   \
            const unsigned long long ex_ = __builtin_amdgcn_read_exec();                            \
            ++(wc)->sec_lanes[i];                                                                   \
            if (__builtin_amdgcn_mbcnt_hi((uint32_t)(ex_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex_, 0u)) == 0u) \
                ++(wc)->sec_waves[i];                                                               \
        }                                                                                           \
    } while (0)

// ---------------------------------------------------------------------------
// Counter-based random numbers
// ---------------------------------------------------------------------------

// Philox4x32-10 block for counter (c0, c1, c2, 0).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
    uint32_t c3 = 0u;
    // The key is wave-uniform; keep the compiler from precomputing the ten round keys once per
    // kernel (twenty scalar registers held, and spilled, across the whole loop): formed here, they
    // are ten scalar additions beside the vector work.
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32), not separate high and low halves
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // a ^ b ^ c as one three-input bit operation (truth table 0x96)
        c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
        c1 = lo1;
        c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
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
    float a, inv_a;      // dot(D,D) and, for a tame ray, its correctly rounded reciprocal (see tame_root)
    float a_filter;      // dot(D,D) * (1 - 2^-17), the conservative filter's A
    bool tame;           // all components finite and small enough for the filter's bound
};

// Magnitude limits under which the conservative filter's arithmetic can neither
// overflow nor lose its margin to underflow (see filter_sign), and under which the halved form of
// the root evaluation (tame_root) rounds like the reference's.
constexpr float kTameMax = 0x1p28f;   // |coordinate|, |direction component|
constexpr float kTameMinA = 0x1p-30f; // dot(D,D)

// ---------------------------------------------------------------------------
// Correctly rounded division and square root in a few instructions
// ---------------------------------------------------------------------------
//
// The reference divides and takes square roots with x86-64's divss and sqrtss: IEEE-754, rounded to
// nearest.  The compiler's expansions of `/` and sqrtf give the same results in 12 and 18 instructions
// (range scaling, a refined reciprocal per division, special-case fix-ups).  The sequences below give
// the same results in 3 and 5 for operands inside the stated ranges -- not by an error analysis but by
// enumeration on the MI355X itself (tools/exact_arith.hip): every float of the range for the
// reciprocal and the square root, all 2^46 pairs of significands for the quotient (scaling an operand
// by a power of two scales every intermediate exactly, and the sequences are odd in their operands).
// Outside the ranges the callers use `/` and sqrtf.

// 1 / x, correctly rounded, for 2^-100 <= |x| < 2^126: one correction of the hardware's approximation.
__device__ __forceinline__ float exact_rcp(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

// a / b, correctly rounded, from y = exact_rcp(b): the product is within an ulp or two of the
// quotient, the residual a - b q is exact in one fused multiply-add and a second one rounds q + r y
// to the float nearest a / b.  Needs b, y, the quotient and the residual (about 2^-24 |a|) in the
// normal range.
__device__ __forceinline__ float exact_div(float a, float b, float y)
{
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// sqrt(x), correctly rounded, for x >= 0: from 2^-100 up, one correction of the hardware's
// approximation by its residual and half the approximate reciprocal root; below (zero, subnormals,
// where those approximations are not good enough), sqrtf -- taken by the whole wave as soon as one
// of its lanes needs it.
__device__ __forceinline__ float exact_sqrt(float x)
{
    if (wave_any(x < 0x1p-100f))
        return sqrtf(x);
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(r, h, s0);
}

__device__ __forceinline__ void ray_prepare(RayState &r)
{
    const float A = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz; // dot(): products summed left to right
    r.four_a = 4.0f * A;
    r.two_a = 2.0f * A;
    r.a = A;
    r.inv_a = exact_rcp(A); // (meaningless for an untame ray, and never used for one)
    r.a_filter = __builtin_fmaf(-0x1p-17f, A, A);
    // the sum of magnitudes is NaN or Inf as soon as one component is, and comparisons with NaN
    // are false: one test covers "finite" and "small enough" (it is a little stricter than a test
    // per component, which only sends a few more rays to the exact scan)
    const float m = ((fabsf(r.ox) + fabsf(r.oy)) + (fabsf(r.oz) + fabsf(r.dx))) + (fabsf(r.dy) + fabsf(r.dz));
    r.tame = (m < kTameMax) && (A >= kTameMinA);
}

// Camera::GetRay, Source/main.cpp:440-444, for pixel (i, j) with jitter (ju, jv)
// in [0,1): u and v are formed in double and narrowed (Source/main.cpp:627-628).
// a / b correctly rounded, from rb = 1/b correctly rounded: q = a*rb is within one ulp of the
// quotient, the residual a - b*q is then exact in one fused multiply-add, and a second one rounds
// q + r*rb to the nearest double of a/b (Markstein's theorem; no overflow or underflow for the
// frame coordinates this is used on).  Three full-rate instructions in place of a division sequence.
__device__ __forceinline__ double divide_by_reciprocal(double a, double b, double rb)
{
    const double q = a * rb;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, rb, q);
}

// u, v of Source/main.cpp:627-628 (double sums and quotients, narrowed to float) and Camera::GetRay
// (:440-444).  inv_width / inv_height are 1/(double)(float)width and height, rounded on the host.
// width_d / height_d are (double)(float)width and height.
__device__ __forceinline__ void camera_ray(const rt_camera &cam, double width_d, double height_d,
                                           double inv_width, double inv_height,
                                           int i, int j, double ju, double jv, RayState &r)
{
    const float u = (float)divide_by_reciprocal((double)(float)i + ju, width_d, inv_width);
    const float v = (float)divide_by_reciprocal((double)(float)j + jv, height_d, inv_height);
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
                                                  float &closest, WorkCount *wc = nullptr)
{
    float B;
    const float disc = discriminant(r, s, B);
    if (wc)
        ++wc->exact;
    if (disc < 0.0f)
        return false;
    if (wc)
        ++wc->roots;
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

// Effective root of RaySphereIntersectionTest (Source/main.cpp:490-499) for a
// ray without NaNs: the near root if it is >= tmin, else the far root.
__device__ __forceinline__ void resolve_unordered(const RayState &r, const Geom &s, int idx, float tmin,
                                                  float &closest, int &best, WorkCount *wc = nullptr)
{
    float B;
    const float disc = discriminant(r, s, B);
    if (wc)
        ++wc->exact;
    if (disc < 0.0f)
        return;
    if (wc)
        ++wc->roots;
    const float sq = sqrtf(disc);
    float root = (-B - sq) / r.two_a;
    if (root < tmin)
        root = (-B + sq) / r.two_a;
    if (root >= tmin && (root < closest || (root == closest && idx > best))) {
        closest = root;
        best = idx;
    }
}

// The same for a tame ray (finite, magnitudes below 2^28, A = dot(D,D) at least 2^-30), with every
// quantity at half or a quarter of the reference's: d = B / 2, disc = Discriminant / 4, and roots
// (-d -+ sqrt(disc)) / A.  Scaling by a power of two commutes with every rounding as long as nothing
// leaves the normal range, so each value is the reference's, scaled, and the root is the reference's,
// bit for bit.  Where something does leave the normal range the root is below tmin on both sides: a
// root of at least tmin = 0.001 needs |d| or sqrt(disc) of at least 2^-41 (A >= 2^-30), so the
// larger of d*d and disc is at least 2^-82, and a product below 2^-126 next to it is lost in the
// reference's rounding of the difference as well.  (The early-out `disc < 0` can differ from the
// reference's only when both products are subnormal, i.e. when no root reaches tmin.)
__device__ __forceinline__ float tame_discriminant(const RayState &r, const Geom &s, float &d, float &C)
{
    const float ocx = r.ox - s.cx, ocy = r.oy - s.cy, ocz = r.oz - s.cz;
    d = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    C = (ocx * ocx + ocy * ocy + ocz * ocz) - s.r2;
    return d * d - r.a * C;
}

// Effective root (Source/main.cpp:490-499) for a tame ray without NaNs: the near root if it is >=
// tmin, else the far root; both are formed (three instructions each) rather than branched on.
__device__ __forceinline__ float tame_root(const RayState &r, float d, float disc, float tmin)
{
    const float sq = exact_sqrt(disc);
    const float near_root = exact_div(-d - sq, r.a, r.inv_a);
    const float far_root = exact_div(sq - d, r.a, r.inv_a);
    return near_root < tmin ? far_root : near_root;
}

// resolve_candidate for a tame ray: the in-order rule (an equal T is accepted, so the later sphere
// wins a tie), both roots formed rather than branched on.
__device__ __forceinline__ bool resolve_candidate_tame(const RayState &r, const Geom &s, float tmin, float &closest,
                                                       WorkCount *wc = nullptr)
{
    float d, C;
    const float disc = tame_discriminant(r, s, d, C);
    if (wc)
        ++wc->exact;
    if (disc < 0.0f)
        return false;
    if (wc)
        ++wc->roots;
    const float sq = exact_sqrt(disc);
    const float near_root = exact_div(-d - sq, r.a, r.inv_a);
    const float far_root = exact_div(sq - d, r.a, r.inv_a);
    const bool near_out = near_root < tmin || near_root > closest;
    const float root = near_out ? far_root : near_root;
    if (root < tmin || root > closest)
        return false;
    closest = root;
    return true;
}

// resolve_unordered for a tame ray.
__device__ __forceinline__ void resolve_unordered_tame(const RayState &r, const Geom &s, int idx, float tmin,
                                                       float &closest, int &best, WorkCount *wc = nullptr)
{
    float d, C;
    const float disc = tame_discriminant(r, s, d, C);
    if (wc)
        ++wc->exact;
    if (disc < 0.0f)
        return;
    if (wc)
        ++wc->roots;
    const float root = tame_root(r, d, disc, tmin);
    if (root >= tmin && (root < closest || (root == closest && idx > best))) {
        closest = root;
        best = idx;
    }
}

// Hit point and normal, Source/main.cpp:503-514.  TAME_SCENE: every radius of the scene has a
// magnitude in [2^-20, 2^28) (the precondition of every scan but the exact one); then the three
// divisions by the radius go through its reciprocal whenever all three numerators of all the wave's
// rays have magnitudes in [2^-80, 2^80] (zero, subnormal, huge or NaN ones send the wave to `/`).
template <bool TAME_SCENE = false>
__device__ __forceinline__ void hit_frame(const RayState &r, float t, float cx, float cy, float cz,
                                          float radius, float &px, float &py, float &pz, float &nx,
                                          float &ny, float &nz, bool &inside)
{
    px = r.ox + (r.dx * t); py = r.oy + (r.dy * t); pz = r.oz + (r.dz * t);
    const float x0 = px - cx, x1 = py - cy, x2 = pz - cz;
    bool slow = true;
    if (TAME_SCENE) {
        const float lo = fminf(fminf(fabsf(x0), fabsf(x1)), fabsf(x2));
        const float hi = fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fabsf(x2));
        slow = wave_any(!(lo >= 0x1p-80f && hi <= 0x1p80f));
    }
    if (slow) {
        nx = x0 / radius; ny = x1 / radius; nz = x2 / radius;
    } else {
        const float inv = exact_rcp(radius);
        nx = exact_div(x0, radius, inv); ny = exact_div(x1, radius, inv); nz = exact_div(x2, radius, inv);
    }
    inside = (r.dx * nx + r.dy * ny + r.dz * nz) > 0.0f;
    if (inside) {
        nx = -nx; ny = -ny; nz = -nz;
    }
}


} // namespace
} // namespace rtamd
