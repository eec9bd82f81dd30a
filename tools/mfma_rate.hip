// mfma_rate.hip -- does the FP32 MFMA pipe run beside the vector ALU on gfx950?
// Diagnostic tool: times loops of (a) MFMA only, (b) VALU only, (c) both interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters)
{
    f32x4 acc[4] = {{a, a, a, a}, {b, b, b, b}, {a, b, a, b}, {b, a, b, a}};
    f32x16 big = {};
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) { // 4 independent 16x16x4 f32 MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) { // 16 independent-ish VALU fmas
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
        }
        if (MODE == 4 || MODE == 5) { // 4 independent 16x16x32 bf16 MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
        }
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
        }
        if (MODE == 3) { // 2 x 32x32x2 f32 MFMA (same flops as 4 x 16x16x4... 2*4096 vs 4*2048)
            big = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big, 0, 0, 0);
            big = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, big, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += big[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int wg, int iters, float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(256), 0, 0, d, 1.0001f, 0.9999f, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(256), 0, 0, d, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double)wg * iters); // seconds per loop iteration per SIMD
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 100000;
    printf("ns per iteration per SIMD (x ~2.2 GHz = cycles):  4xMFMA16x16x4f32 | 16xv_fma | both | 2xMFMA32x32x2f32 | 4xMFMA16x16x32bf16 | bf16+16xv_fma\n");
    for (int wg = 1; wg <= 8; wg *= 2)
        printf("waves/SIMD %d: %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n", wg, run<0>(wg, iters, d) * 1e9, run<1>(wg, iters, d) * 1e9,
               run<2>(wg, iters, d) * 1e9, run<3>(wg, iters, d) * 1e9, run<4>(wg, iters, d) * 1e9, run<5>(wg, iters, d) * 1e9);
    return 0;
}
