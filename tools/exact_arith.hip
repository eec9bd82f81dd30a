// exact_arith.hip -- proof by enumeration, on the GPU, that the kernels' short division and
// square-root sequences return the IEEE-754 correctly rounded result (what the reference's
// x86-64 build computes with divss / sqrtss, Source/main.cpp:490-499, 503-514).
//
//   division   a / b from y = RN(1/b):  q = a*y;  r = fma(-b, q, a);  q' = fma(r, y, q)
//              checked for ALL 2^23 x 2^23 pairs of significands (a, b in [1, 2)); scaling either
//              operand by a power of two scales every intermediate exactly, and the sequence is odd
//              in a and in b, so this covers every pair of normal floats whose intermediates stay
//              in the normal range.
//   sqrt       s from the hardware's approximations (v_sqrt_f32, v_rsq_f32) and one or two fused
//              corrections, checked for every float in [2^-100, 2^127].
//
// Build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -o tools/exact_arith tools/exact_arith.hip
// Usage: tools/exact_arith [div_fraction_log2]   (0 = the whole division space, k = every 2^k-th b)
// Prints one JSON line per check; the reference side of every comparison is the compiler's own
// correctly rounded `/` and `sqrtf` (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(call)                                                                      \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                   \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

struct Tally {
    unsigned long long bad[6];
    unsigned int example[6][2];
};

__device__ __forceinline__ float quotient_from_reciprocal(float a, float b, float y)
{
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// One thread per divisor significand in [b_first, b_first + count * b_step) and sixteenth of the
// dividends (blockIdx.y), so that every divisor meets all 2^23 dividends.
__global__ void __launch_bounds__(256) div_kernel(unsigned b_first, unsigned b_step, unsigned count, Tally *t)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const unsigned mb = b_first + i * b_step;
    const float b = __uint_as_float(0x3F800000u | mb);
    const float y = 1.0f / b; // correctly rounded
    unsigned bad = 0u, first = 0u;
    const unsigned a_first = blockIdx.y << 19;
#pragma unroll 4
    for (unsigned ma = a_first; ma < a_first + (1u << 19); ++ma) {
        const float a = __uint_as_float(0x3F800000u | ma);
        const float want = a / b;
        const float got = quotient_from_reciprocal(a, b, y);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (bad == 0u)
                first = ma;
            ++bad;
        }
    }
    if (bad != 0u) {
        atomicAdd(&t->bad[0], (unsigned long long)bad);
        t->example[0][0] = first;
        t->example[0][1] = mb;
    }
}

// Square-root candidates on every float of a range of bit patterns.
//   1: s0 = v_sqrt(x); r = fma(-s0, s0, x); s = fma(r, 0.5 * v_rsq(x), s0)
//   2: g = x * v_rsq(x), h = 0.5 * v_rsq(x); two coupled corrections (Markstein)
//   3: s0 = v_sqrt(x); h = 0.5 * v_rcp(s0); r = fma(-s0, s0, x); s = fma(r, h, s0)
//   4: candidate 1 applied twice
//   5: 1 / x from v_rcp and one correction (not a square root: the reciprocal the division needs)
__global__ void __launch_bounds__(256) sqrt_kernel(unsigned first_bits, unsigned long long count, Tally *t)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned n_bad[6] = { 0u, 0u, 0u, 0u, 0u, 0u }, ex[6][2] = {};
#define note(t_, k_, x_, y_) do { if (n_bad[k_]++ == 0u) { ex[k_][0] = (x_); ex[k_][1] = (y_); } } while (0)
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const unsigned bits = first_bits + (unsigned)i;
        const float x = __uint_as_float(bits);
        const float want = sqrtf(x);
        const float rs = __builtin_amdgcn_rsqf(x);
        const float s0 = __builtin_amdgcn_sqrtf(x);
        {
            const float r = __builtin_fmaf(-s0, s0, x);
            const float s = __builtin_fmaf(r, 0.5f * rs, s0);
            if (__float_as_uint(s) != __float_as_uint(want))
                note(t, 1, bits, __float_as_uint(s));
            const float r2 = __builtin_fmaf(-s, s, x);
            const float s2 = __builtin_fmaf(r2, 0.5f * rs, s);
            if (__float_as_uint(s2) != __float_as_uint(want))
                note(t, 4, bits, __float_as_uint(s2));
        }
        {
            const float g = x * rs, h = 0.5f * rs;
            const float r = __builtin_fmaf(-g, h, 0.5f);
            const float g1 = __builtin_fmaf(g, r, g), h1 = __builtin_fmaf(h, r, h);
            const float d = __builtin_fmaf(-g1, g1, x);
            const float s = __builtin_fmaf(d, h1, g1);
            if (__float_as_uint(s) != __float_as_uint(want))
                note(t, 2, bits, __float_as_uint(s));
        }
        {
            const float h = 0.5f * __builtin_amdgcn_rcpf(s0);
            const float r = __builtin_fmaf(-s0, s0, x);
            const float s = __builtin_fmaf(r, h, s0);
            if (__float_as_uint(s) != __float_as_uint(want))
                note(t, 3, bits, __float_as_uint(s));
        }
        {
            const float y0 = __builtin_amdgcn_rcpf(x);
            const float e = __builtin_fmaf(-x, y0, 1.0f);
            const float y = __builtin_fmaf(e, y0, y0);
            const float inv = 1.0f / x;
            if (__float_as_uint(y) != __float_as_uint(inv))
                note(t, 5, bits, __float_as_uint(y));
        }
    }
#undef note
    for (int k = 1; k < 6; ++k)
        if (n_bad[k] != 0u && atomicAdd(&t->bad[k], (unsigned long long)n_bad[k]) == 0ull) {
            t->example[k][0] = ex[k][0];
            t->example[k][1] = ex[k][1];
        }
}

int main(int argc, char **argv)
{
    const int skip_log2 = argc > 1 ? atoi(argv[1]) : 0;
    Tally *t = nullptr;
    CHECK(hipMalloc((void **)&t, sizeof(Tally)));
    Tally h;

    // ---- square roots: every float from 2^-100 up to the largest finite one
    {
        CHECK(hipMemset(t, 0, sizeof(Tally)));
        const unsigned first = 0x0D800000u; // 2^-100
        const unsigned long long count = 0x7F800000ull - first;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(sqrt_kernel, dim3(256 * 64), dim3(256), 0, 0, first, count, t);
        CHECK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CHECK(hipMemcpy(&h, t, sizeof h, hipMemcpyDeviceToHost));
        const char *names[6] = { "", "sqrt_rsq_one_correction", "sqrt_markstein_coupled", "sqrt_rcp_one_correction",
                                 "sqrt_rsq_two_corrections", "reciprocal_one_correction" };
        for (int k = 1; k <= 5; ++k)
            printf("{\"check\": \"%s\", \"inputs\": %llu, \"range\": \"every float in [2^-100, FLT_MAX]\", \"mismatches\": %llu, "
                   "\"example_bits\": [\"%08x\", \"%08x\"], \"seconds\": %.2f}\n",
                   names[k], count, h.bad[k], h.example[k][0], h.example[k][1], s);
        fflush(stdout);
    }

    // ---- divisions: all pairs of significands, in slices so that progress shows
    {
        CHECK(hipMemset(t, 0, sizeof(Tally)));
        const unsigned step = 1u << skip_log2;
        const unsigned total_b = (1u << 23) >> skip_log2;
        const unsigned slice = 1u << 17; // divisors per launch
        unsigned long long pairs = 0ull;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned done = 0u; done < total_b; done += slice) {
            const unsigned n = total_b - done < slice ? total_b - done : slice;
            // (with a step, the divisors of a slice are spread over the whole range, offset by the slice number)
            hipLaunchKernelGGL(div_kernel, dim3((n + 255u) / 256u, 16u), dim3(256), 0, 0, done * step + (step > 1u ? (done / slice) % step : 0u),
                               step, n, t);
            CHECK(hipDeviceSynchronize());
            pairs += (unsigned long long)n << 23;
            CHECK(hipMemcpy(&h, t, sizeof h, hipMemcpyDeviceToHost));
            fprintf(stderr, "division: %u of %u divisors, %llu mismatches, %.1f s\n", done + n, total_b, h.bad[0],
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"check\": \"quotient_from_reciprocal\", \"pairs\": %llu, \"range\": \"a, b in [1, 2): %s\", "
               "\"mismatches\": %llu, \"example_significands\": [\"%06x\", \"%06x\"], \"seconds\": %.2f}\n",
               pairs, skip_log2 ? "all dividends, a sample of the divisors" : "all 2^46 pairs of significands", h.bad[0],
               h.example[0][0], h.example[0][1], s);
    }
    CHECK(hipFree(t));
    return 0;
}
