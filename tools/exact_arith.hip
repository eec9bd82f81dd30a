// exact_arith.hip -- proof by enumeration, on the GPU, that the kernels' short division and
// square-root sequences return the IEEE-754 correctly rounded result (what the reference's
// x86-64 build computes with divss / sqrtss, Source/main.cpp:490-499, 503-514).
//
//   division   a / b from y = 1/b correctly rounded:  q = a*y;  r = fma(-b, q, a);  q' = fma(r, y, q)
//              checked for ALL 2^23 x 2^23 pairs of significands (a, b in [1, 2)); scaling either
//              operand by a power of two scales every intermediate exactly, and the sequence is odd
//              in a and in b, so this covers every pair of normal floats whose intermediates stay
//              in the normal range.
//   sqrt       s from the hardware's approximations (v_sqrt_f32, v_rsq_f32) and one fused correction,
//              checked for every non-negative finite float.
//   1 / x      from v_rcp_f32 and one fused correction, for every float of magnitude in [2^-100, 2^126).
//
// The functions under test are the library's own (rt_arith.h).  Other candidates that were tried: the
// square root corrected with half the RECIPROCAL of the approximation (v_rcp_f32) is wrong for 114
// inputs; the reciprocal sequence is wrong from 2^126 up (its result is subnormal there).
//
// Build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Iray-tracer_amd/csrc -o tools/exact_arith tools/exact_arith.hip
// Usage: tools/exact_arith [div_fraction_log2]   (0 = the whole division space, k = every 2^k-th b)
// Prints one JSON line per check; the reference side of every comparison is the compiler's own
// correctly rounded `/` and `sqrtf` (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_arith.h" // the library's own exact_rcp, exact_div, exact_sqrt

using namespace rtamd;

#define CHECK(call)                                                                      \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                   \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

struct Tally {
    unsigned long long bad[4];
    unsigned int example[4][2];
};

// One thread per divisor significand in [b_first, b_first + count * b_step) and sixteenth of the
// dividends (blockIdx.y), so that every divisor meets all 2^23 dividends.
__global__ void __launch_bounds__(256) div_kernel(unsigned b_first, unsigned b_step, unsigned count, Tally *t)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const unsigned mb = b_first + i * b_step;
    const float b = __uint_as_float(0x3F800000u | mb);
    const float y = exact_rcp(b);
    unsigned bad = 0u, first = 0u;
    const unsigned a_first = blockIdx.y << 19;
#pragma unroll 4
    for (unsigned ma = a_first; ma < a_first + (1u << 19); ++ma) {
        const float a = __uint_as_float(0x3F800000u | ma);
        const float want = a / b;
        const float got = exact_div(a, b, y);
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

// exact_sqrt on every bit pattern of [first_bits, first_bits + count) (non-negative floats), exact_rcp on
// those of them inside its range and on their negatives.
__global__ void __launch_bounds__(256) unary_kernel(unsigned first_bits, unsigned long long count, Tally *t)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned n_bad[4] = { 0u, 0u, 0u, 0u }, ex[4][2] = {};
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const unsigned bits = first_bits + (unsigned)i;
        const float x = __uint_as_float(bits);
        const float s = exact_sqrt(x);
        if (__float_as_uint(s) != __float_as_uint(sqrtf(x)) && n_bad[1]++ == 0u) {
            ex[1][0] = bits;
            ex[1][1] = __float_as_uint(s);
        }
        if (x >= 0x1p-100f && x < 0x1p126f) {
            const float y = exact_rcp(x), yn = exact_rcp(-x);
            if ((__float_as_uint(y) != __float_as_uint(1.0f / x) || __float_as_uint(yn) != __float_as_uint(1.0f / -x)) &&
                n_bad[2]++ == 0u) {
                ex[2][0] = bits;
                ex[2][1] = __float_as_uint(y);
            }
        }
    }
    for (int k = 1; k < 3; ++k)
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

    // ---- square roots of every non-negative finite float (zero and subnormals included: they take the
    // sqrtf path inside exact_sqrt), reciprocals of every float of magnitude in [2^-100, 2^126)
    {
        CHECK(hipMemset(t, 0, sizeof(Tally)));
        const unsigned long long count = 0x7F800000ull;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(unary_kernel, dim3(256 * 64), dim3(256), 0, 0, 0u, count, t);
        CHECK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CHECK(hipMemcpy(&h, t, sizeof h, hipMemcpyDeviceToHost));
        printf("{\"check\": \"exact_sqrt\", \"inputs\": %llu, \"range\": \"every float in [0, FLT_MAX]\", \"mismatches\": %llu, "
               "\"example_bits\": [\"%08x\", \"%08x\"], \"seconds\": %.2f}\n", count, h.bad[1], h.example[1][0], h.example[1][1], s);
        printf("{\"check\": \"exact_rcp\", \"inputs\": %llu, \"range\": \"every float x and -x with x in [2^-100, 2^126)\", "
               "\"mismatches\": %llu, \"example_bits\": [\"%08x\", \"%08x\"], \"seconds\": %.2f}\n",
               2ull * (0x7E800000ull - 0x0D800000ull), h.bad[2], h.example[2][0], h.example[2][1], s);
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
            hipLaunchKernelGGL(div_kernel, dim3((n + 255u) / 256u, 16u), dim3(256), 0, 0,
                               done * step + (step > 1u ? (done / slice) % step : 0u), step, n, t);
            CHECK(hipDeviceSynchronize());
            pairs += (unsigned long long)n << 23;
            CHECK(hipMemcpy(&h, t, sizeof h, hipMemcpyDeviceToHost));
            fprintf(stderr, "division: %u of %u divisors, %llu mismatches, %.1f s\n", done + n, total_b, h.bad[0],
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"check\": \"exact_div\", \"pairs\": %llu, \"range\": \"a, b in [1, 2): %s, y = exact_rcp(b)\", "
               "\"mismatches\": %llu, \"example_significands\": [\"%06x\", \"%06x\"], \"seconds\": %.2f}\n",
               pairs, skip_log2 ? "all dividends, a sample of the divisors" : "all 2^46 pairs of significands", h.bad[0],
               h.example[0][0], h.example[0][1], s);
    }
    CHECK(hipFree(t));
    return 0;
}
