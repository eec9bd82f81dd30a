// valu_rate.hip -- measures the issue rate of unfused / fused, scalar / packed
// FP32 vector arithmetic on one MI355X at several occupancies.  Diagnostic tool
// (not part of the library): its numbers justify the instruction mix chosen for
// the ray-sphere discriminant loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float *out, float a, float b, int iters)
{
    // 8 independent chains
    float x[8];
    float2v p[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i; p[i] = float2v{a + i, a - i}; }
    const float2v pb = {b, b * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { // v_mul_f32 + v_add_f32
                asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x[i]) : "v"(b), "v"(a));
            } else if (MODE == 1) { // v_fma_f32 x2
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
            } else if (MODE == 2) { // v_pk_mul_f32 + v_pk_add_f32
                asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            } else if (MODE == 3) { // v_pk_fma_f32 x2
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
            } else if (MODE == 4) { // v_cmp + v_addc (mask shift-in)
                asm volatile("v_cmp_nlt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x[i]) : "v"(b), "v"(a) : "vcc");
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int wg_per_cu, int iters, float *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0001f, 0.9999f, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions issued per SIMD = waves/SIMD * iters * 16
    const double inst_per_simd = (double)wg_per_cu * iters * 16.0;
    return inst_per_simd / (ms * 1e-3); // wave-instructions per second per SIMD
}

int main()
{
    float *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    const char *names[5] = {"v_mul+v_add (2 flop/lane/pair)", "v_fma x2", "v_pk_mul+v_pk_add", "v_pk_fma x2", "v_cmp+v_addc"};
    const int iters = 200000;
    printf("mode, waves/SIMD, G wave-inst/s/SIMD, cycles/inst @2.4GHz\n");
    for (int wg = 1; wg <= 8; wg *= 2) {
        double r[5];
        r[0] = run<0>(wg, iters, d_out);
        r[1] = run<1>(wg, iters, d_out);
        r[2] = run<2>(wg, iters, d_out);
        r[3] = run<3>(wg, iters, d_out);
        r[4] = run<4>(wg, iters, d_out);
        for (int m = 0; m < 5; ++m)
            printf("%-32s %d  %.3f  %.2f\n", names[m], wg, r[m] * 1e-9, 2.4e9 / r[m]);
    }
    return 0;
}
