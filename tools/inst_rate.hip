// inst_rate.hip -- issue cost of individual vector instructions on gfx950 (8 waves/SIMD,
// 8 independent chains per wave).  Diagnostic tool.
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, ASM)                                                                        \
    __global__ void __launch_bounds__(256) NAME(unsigned *out, unsigned a, unsigned b, int iters, unsigned long long *clk) \
    {                                                                                             \
        unsigned x[8];                                                                            \
        for (int i = 0; i < 8; ++i) x[i] = a + i;                                                 \
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < iters; ++it) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM "\n" ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
        }                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; } \
        unsigned s = 0; for (int i = 0; i < 8; ++i) s += x[i];                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, %1, 5")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_mad_u64, "v_mad_u64_u32 v[20:21], vcc, %0, %1, v[20:21]")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_pk_fma, "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[24:25]")
KERNEL(k_pk_mul, "v_pk_mul_f32 v[20:21], v[20:21], v[22:23]")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_xor3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")

typedef void (*kern_t)(unsigned *, unsigned, unsigned, int, unsigned long long *);

static void run(const char *name, kern_t k, unsigned *d, unsigned long long *dclk)
{
    const int wg = 8, iters = 50000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256 * wg), dim3(256), 0, 0, d, 3u, 5u, 8, dclk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256 * wg), dim3(256), 0, 0, d, 3u, 5u, iters, dclk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2]; hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    const double inst = (double)wg * iters * 16.0;
    printf("%-16s %6.2f cycles per wave-instruction per SIMD (%.2f GHz)\n", name, ms * 1e-3 * ghz * 1e9 / inst, ghz);
}

int main()
{
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *dclk; hipMalloc(&dclk, 16);
#define RUN(K) run(#K, K, d, dclk)
    RUN(k_fma); RUN(k_alignbit); RUN(k_lshl_or); RUN(k_cndmask); RUN(k_add); RUN(k_xor); RUN(k_mullo); RUN(k_mulhi);
    RUN(k_sqrt); RUN(k_rcp); RUN(k_bfe); RUN(k_and_or); RUN(k_max3); RUN(k_cvt); RUN(k_cmp);
    RUN(k_min); RUN(k_max); RUN(k_mul); RUN(k_sub); RUN(k_min3); RUN(k_med3); RUN(k_cndmask64); RUN(k_lshl_add);
    RUN(k_mad_u64); RUN(k_perm); RUN(k_cvt_ub); RUN(k_bfi); RUN(k_pk_fma); RUN(k_pk_mul); RUN(k_mov); RUN(k_lshlrev);
    RUN(k_and); RUN(k_xor3); RUN(k_add3); RUN(k_fmac);
    return 0;
}
