// valu_rates.hip -- issue rate of the vector instructions the mcorb kernels lean on (gfx950): cycles per wave-instruction
// per SIMD with 1, 2 and 4 waves per SIMD.  Each kernel runs a long unrolled chain of independent instructions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define N_IT 256
#define UNROLL 16
#define KERNEL(NAME, BODY)                                                              \
    __global__ void NAME(uint32_t *out, uint32_t seed)                                   \
    {                                                                                   \
        uint32_t a[UNROLL], b = seed + threadIdx.x, c = seed * 7 + 3;                    \
        for (int u = 0; u < UNROLL; u++) a[u] = seed + u * 977 + threadIdx.x;            \
        for (int it = 0; it < N_IT; it++) {                                              \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { BODY; }                 \
        }                                                                               \
        uint32_t s = 0;                                                                 \
        for (int u = 0; u < UNROLL; u++) s ^= a[u];                                      \
        if (s == 0x12345) out[0] = s;                                                   \
    }
KERNEL(k_add, a[u] = a[u] + b)
KERNEL(k_mad24, a[u] = __umul24(a[u], b) + c)
KERNEL(k_mul_lo, a[u] = a[u] * b)
KERNEL(k_dot4, a[u] = __builtin_amdgcn_udot4(a[u], b, c, false))
KERNEL(k_dot2, a[u] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a[u]), __builtin_bit_cast(u16x2, b), c, false))
KERNEL(k_pk_mad, a[u] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a[u]) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c))))
KERNEL(k_pk_max, a[u] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a[u]), __builtin_bit_cast(s16x2, b))))
KERNEL(k_perm, a[u] = __builtin_amdgcn_perm(a[u], b, 0x07020500u))
KERNEL(k_alignbyte, a[u] = __builtin_amdgcn_alignbyte(a[u], b, 1))
KERNEL(k_bcnt, a[u] = __builtin_popcount(a[u]) + b)
KERNEL(k_max3, a[u] = max(max(a[u], b), c))
KERNEL(k_mbcnt, a[u] = __builtin_amdgcn_mbcnt_lo(a[u], b))
KERNEL(k_sad, a[u] = __builtin_amdgcn_sad_u8(a[u], b, c))
KERNEL(k_lshl_add, a[u] = (a[u] << 3) + b)

int main()
{
    uint32_t *out;
    hipMalloc((void **)&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
#define RUN(K)                                                                                              \
    for (int wps = 1; wps <= 4; wps *= 2) {                                                                   \
        const int blocks = 256 * wps, threads = 256; /* 4 waves per block = one per SIMD; wps blocks per CU */  \
        hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, 12345u);                                \
        hipEventRecord(e0, 0);                                                                                \
        for (int r = 0; r < 10; r++) hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, 12345u);   \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                                       \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                           \
        const double instr_per_simd = (double)N_IT * UNROLL * wps * 10;                                        \
        printf("%-12s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", #K, wps, \
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);                                   \
    }
    RUN(k_add) RUN(k_mad24) RUN(k_mul_lo) RUN(k_dot4) RUN(k_dot2) RUN(k_pk_mad) RUN(k_pk_max) RUN(k_perm) RUN(k_alignbyte)
    RUN(k_bcnt) RUN(k_max3) RUN(k_mbcnt) RUN(k_sad) RUN(k_lshl_add)
    return 0;
}
