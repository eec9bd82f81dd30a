// valu_rate.hip -- measures the issue rate of unfused / fused, scalar / packed
// FP32 vector arithmetic on one MI355X at several occupancies.  Diagnostic tool
// (not part of the library): its numbers justify the instruction mix chosen for
// the ray-sphere discriminant loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float *out, float a, float b, int iters, unsigned long long *clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
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
    if (clk && threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;      // shader-clock cycles
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;  // 100 MHz ticks
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int wg_per_cu, int iters, float *d_out, unsigned long long *d_clk, double *cyc_per_inst, double *ghz)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0001f, 0.9999f, 16, nullptr);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0001f, 0.9999f, iters, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions issued per SIMD = waves/SIMD * iters * 16
    unsigned long long clk[2];
    hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
    const double inst_per_simd = (double)wg_per_cu * iters * 16.0;
    *cyc_per_inst = (double)clk[0] / inst_per_simd; // block 0's cycles over the SIMD's instruction count
    *ghz = (double)clk[0] / ((double)clk[1] * 10.0); // cycles per ns
    return inst_per_simd / (ms * 1e-3); // wave-instructions per second per SIMD
}

int main()
{
    float *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    unsigned long long *d_clk;
    hipMalloc(&d_clk, 16);
    const char *names[5] = {"v_mul+v_add (2 flop/lane/pair)", "v_fma x2", "v_pk_mul+v_pk_add", "v_pk_fma x2", "v_cmp+v_addc"};
    const int iters = 200000;
    printf("mode, waves/SIMD, G wave-inst/s/SIMD, shader cycles per wave-inst, shader GHz\n");
    for (int wg = 1; wg <= 8; wg *= 2) {
        double r[5], c[5], g[5];
        r[0] = run<0>(wg, iters, d_out, d_clk, &c[0], &g[0]);
        r[1] = run<1>(wg, iters, d_out, d_clk, &c[1], &g[1]);
        r[2] = run<2>(wg, iters, d_out, d_clk, &c[2], &g[2]);
        r[3] = run<3>(wg, iters, d_out, d_clk, &c[3], &g[3]);
        r[4] = run<4>(wg, iters, d_out, d_clk, &c[4], &g[4]);
        for (int m = 0; m < 5; ++m)
            printf("%-32s %d  %.3f  %.2f  %.2f\n", names[m], wg, r[m] * 1e-9, c[m], g[m]);
    }
    return 0;
}
