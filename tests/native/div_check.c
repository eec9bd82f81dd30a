/* Checks on the host that a/b formed as  q = a*rb;  r = fma(-b, q, a);  q + r*rb (one fma)
 * with rb = 1/b correctly rounded equals the correctly rounded quotient a/b, for the operands the
 * sample loop forms: a = (double)(float)i + (double)ju with 0 <= i < 2^24 and ju a float in [0,1),
 * b = (double)(float)width (Source/main.cpp:627-628).  Prints the number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static uint64_t s = 0x9e3779b97f4a7c15ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

static double div_by_reciprocal(double a, double b, double rb)
{
    const double q = a * rb;
    const double r = fma(-b, q, a);
    return fma(r, rb, q);
}

int main(int argc, char **argv)
{
    const long per_width = argc > 1 ? atol(argv[1]) : 2000;
    long bad = 0, n = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t n_widths = pass == 0 ? 4096u : 20000u;
        for (uint32_t k = 1; k <= n_widths; ++k) {
            const uint32_t w = pass == 0 ? k : (uint32_t)(rnd() % 16777216u) + 1u;
            const double b = (double)(float)w, rb = 1.0 / b;
            for (long t = 0; t < per_width; ++t) {
                const uint32_t i = (uint32_t)(rnd() % (uint64_t)(w < 16777216u ? w : 16777216u));
                /* jitter: a float in [0,1) on a 2^-24 grid as the samplers produce, or any float */
                const float ju = (t & 1) ? (float)(rnd() >> 40) * 0x1p-24f
                                         : ldexpf((float)(rnd() >> 40) * 0x1p-24f, -(int)(rnd() % 30u));
                const double a = (double)(float)i + (double)ju;
                const double want = a / b, got = div_by_reciprocal(a, b, rb);
                ++n;
                if (want != got || (float)want != (float)got)
                    ++bad;
            }
            /* the pixel centres of the deterministic hit map: i + 0.5 */
            const double a = (double)(float)(w - 1u) + 0.5;
            if (a / b != div_by_reciprocal(a, b, rb))
                ++bad;
        }
    }
    printf("%ld %ld\n", bad, n);
    return 0;
}
