// valu_rates.hip -- issue rate of the instructions the mcorb kernels lean on (gfx950), as a function of the waves a SIMD
// holds: cycles per wave-instruction per SIMD at 1, 2, 4 and 8 waves per SIMD.
//
// Method.  Every kernel is a loop whose body is 256 instructions of straight-line code: 16 INDEPENDENT chains of one
// instruction, 16 rounds (inline asm volatile: the compiler can neither fold nor reorder it; a first version with 16
// instructions per trip measured the taken branch of a lone wave, not the instruction), 512 trips = 131 072
// wave-instructions per wave.  A workgroup is 256 threads = one wave per SIMD; exactly `wps` workgroups are resident on
// every CU: each asks for 160 KiB / (wps + 0.5) of LDS (wps fit, wps + 1 do not) and 256 * wps are launched.  Every wave
// stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop and records HW_ID.  Reported per test:
//   cyc    = shader cycles per wave-instruction PER SIMD over the whole launch = (latest end - earliest start) x clock x SIMDs
//            / (waves x instructions per wave): the aggregate issue rate, whatever the arbitration between the waves of a SIMD
//   res    = waves that were resident on a SIMD TOGETHER (median over the SIMDs of the largest number of overlapping
//            [start, end] intervals among the waves whose HW_ID / XCC_ID name that SIMD): the waves/SIMD really measured
//   GHz    = shader clock during the loop = s_memtime delta / s_memrealtime delta x 0.1 (median wave)
//   w/a    = median wave's own cycles per instruction (a wave alone: the `1` column)
// Build: make -C tools; run: tools/_build/valu_rates > profiles/rNN_valu_rates.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

#define N_IT 512
#define UNROLL 16

// BODY uses a[u] (chain register), b, c (loop-invariant operands), addr (per-lane LDS byte address)
#define KERNEL(NAME, BODY, TAIL)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, unsigned long long *cyc, uint32_t seed)                \
    {                                                                                                                \
        extern __shared__ uint32_t lds[];                                                                            \
        uint32_t a[UNROLL], b = seed + threadIdx.x, c = seed * 7 + 3;                                                \
        for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i * 2654435761u;                                      \
        __syncthreads();                                                                                             \
        const uint32_t addr = (threadIdx.x & 63) * 4 + ((threadIdx.x >> 6) << 10);                                   \
        (void)addr;                                                                                                  \
        for (int u = 0; u < UNROLL; u++) a[u] = seed + u * 977 + threadIdx.x;                                        \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        _Pragma("unroll 1") for (int it = 0; it < N_IT; it++) {                                                       \
            REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL;                                  \
            REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL;                                  \
            REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL;                                  \
            REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL; REP16(BODY) TAIL;                                  \
        }                                                                                                            \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                              \
        uint32_t s = 0;                                                                                              \
        for (int u = 0; u < UNROLL; u++) s ^= a[u];                                                                  \
        if (s == 0x12345) out[0] = s;                                                                                \
        if ((threadIdx.x & 63) == 0) {                                                                               \
            unsigned long long *o = cyc + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;                         \
            o[0] = t1 - t0; o[1] = r0; o[2] = r1;                                                                   \
            o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        \
        }                                                                                                            \
    }
#define REP16(B) B(0) B(1) B(2) B(3) B(4) B(5) B(6) B(7) B(8) B(9) B(10) B(11) B(12) B(13) B(14) B(15)
#define V2(OP, u) asm volatile(OP " %0, %0, %1" : "+v"(a[u]) : "v"(b))
#define V3(OP, u) asm volatile(OP " %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c))
#define NOTAIL (void)0
#define LGKM asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

#define B_k_add_u32(u) { V2("v_add_u32", u); }
KERNEL(k_add_u32, B_k_add_u32, NOTAIL)
#define B_k_and_b32(u) { V2("v_and_b32", u); }
KERNEL(k_and_b32, B_k_and_b32, NOTAIL)
#define B_k_max_u32(u) { V2("v_max_u32", u); }
KERNEL(k_max_u32, B_k_max_u32, NOTAIL)
#define B_k_xor_b32(u) { V2("v_xor_b32", u); }
KERNEL(k_xor_b32, B_k_xor_b32, NOTAIL)
#define B_k_sub_u32(u) { V2("v_sub_u32", u); }
KERNEL(k_sub_u32, B_k_sub_u32, NOTAIL)
#define B_k_lshlrev_b32(u) { asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[u])); }
KERNEL(k_lshlrev_b32, B_k_lshlrev_b32, NOTAIL)
#define B_k_min_u32(u) { V2("v_min_u32", u); }
KERNEL(k_min_u32, B_k_min_u32, NOTAIL)
#define B_k_max_i32(u) { V2("v_max_i32", u); }
KERNEL(k_max_i32, B_k_max_i32, NOTAIL)
#define B_k_max_u16(u) { V2("v_max_u16", u); }
KERNEL(k_max_u16, B_k_max_u16, NOTAIL)
#define B_k_add_f32(u) { V2("v_add_f32", u); }
KERNEL(k_add_f32, B_k_add_f32, NOTAIL)
#define B_k_fma_f32(u) { V3("v_fma_f32", u); }
KERNEL(k_fma_f32, B_k_fma_f32, NOTAIL)
#define B_k_max_f32(u) { V2("v_max_f32", u); }
KERNEL(k_max_f32, B_k_max_f32, NOTAIL)
#define B_k_cndmask(u) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b)); }
KERNEL(k_cndmask, B_k_cndmask, NOTAIL)
#define B_k_add3_u32(u) { V3("v_add3_u32", u); }
KERNEL(k_add3_u32, B_k_add3_u32, NOTAIL)
#define B_k_and_or_b32(u) { V3("v_and_or_b32", u); }
KERNEL(k_and_or_b32, B_k_and_or_b32, NOTAIL)
#define B_k_pk_add_u16(u) { V2("v_pk_add_u16", u); }
KERNEL(k_pk_add_u16, B_k_pk_add_u16, NOTAIL)
#define B_k_mov_b32(u) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[u]) : "v"(b)); }
KERNEL(k_mov_b32, B_k_mov_b32, NOTAIL)
#define B_k_max3_u32(u) { V3("v_max3_u32", u); }
KERNEL(k_max3_u32, B_k_max3_u32, NOTAIL)
#define B_k_med3_i32(u) { V3("v_med3_i32", u); }
KERNEL(k_med3_i32, B_k_med3_i32, NOTAIL)
#define B_k_pk_max_i16(u) { V2("v_pk_max_i16", u); }
KERNEL(k_pk_max_i16, B_k_pk_max_i16, NOTAIL)
#define B_k_pk_min_u16(u) { V2("v_pk_min_u16", u); }
KERNEL(k_pk_min_u16, B_k_pk_min_u16, NOTAIL)
#define B_k_pk_sub_i16(u) { V2("v_pk_sub_i16", u); }
KERNEL(k_pk_sub_i16, B_k_pk_sub_i16, NOTAIL)
#define B_k_pk_lshr_b16(u) { asm volatile("v_pk_lshrrev_b16 %0, 2, %0" : "+v"(a[u])); }
KERNEL(k_pk_lshr_b16, B_k_pk_lshr_b16, NOTAIL)
#define B_k_perm_b32(u) { V3("v_perm_b32", u); }
KERNEL(k_perm_b32, B_k_perm_b32, NOTAIL)
#define B_k_alignbyte(u) { asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[u]) : "v"(b)); }
KERNEL(k_alignbyte, B_k_alignbyte, NOTAIL)
#define B_k_mbcnt_lo(u) { V2("v_mbcnt_lo_u32_b32", u); }
KERNEL(k_mbcnt_lo, B_k_mbcnt_lo, NOTAIL)
#define B_k_mbcnt_hi(u) { V2("v_mbcnt_hi_u32_b32", u); }
KERNEL(k_mbcnt_hi, B_k_mbcnt_hi, NOTAIL)
#define B_k_bcnt(u) { V2("v_bcnt_u32_b32", u); }
KERNEL(k_bcnt, B_k_bcnt, NOTAIL)
#define B_k_lshl_add(u) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[u]) : "v"(b)); }
KERNEL(k_lshl_add, B_k_lshl_add, NOTAIL)
#define B_k_mad_u32_u24(u) { V3("v_mad_u32_u24", u); }
KERNEL(k_mad_u32_u24, B_k_mad_u32_u24, NOTAIL)
#define B_k_mul_lo_u32(u) { V2("v_mul_lo_u32", u); }
KERNEL(k_mul_lo_u32, B_k_mul_lo_u32, NOTAIL)
#define B_k_dot4_u32_u8(u) { V3("v_dot4_u32_u8", u); }
KERNEL(k_dot4_u32_u8, B_k_dot4_u32_u8, NOTAIL)
#define B_k_dot2_u32_u16(u) { V3("v_dot2_u32_u16", u); }
KERNEL(k_dot2_u32_u16, B_k_dot2_u32_u16, NOTAIL)
#define B_k_add_sdwa(u) { asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[u]) : "v"(b)); }
KERNEL(k_add_sdwa, B_k_add_sdwa, NOTAIL)
// compare into an SGPR pair (what a ballot costs on the vector side); the s_nop-free chain is independent per u
#define B_k_cmp_gt_i16(u) { asm volatile("v_cmp_gt_i16_e64 s[20:21], %0, %1" ::"v"(a[u]), "v"(b) : "s20", "s21"); }
KERNEL(k_cmp_gt_i16, B_k_cmp_gt_i16, NOTAIL)
#define B_k_cmp_gt_u32(u) { asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" ::"v"(a[u]), "v"(b) : "s20", "s21"); }
KERNEL(k_cmp_gt_u32, B_k_cmp_gt_u32, NOTAIL)
#define B_k_readlane(u) { asm volatile("v_readlane_b32 s20, %0, 3" ::"v"(a[u]) : "s20"); }
KERNEL(k_readlane, B_k_readlane, NOTAIL)
// LDS: 16 reads in flight, one wait per 16 (how pass 2's ring fetch behaves); byte reads at stride 4 (conflict-free)
// and at the ring's real pattern (address = lane * 1 + row * 48: neighbouring lanes in one dword = broadcast-free conflicts)
#define B_k_ds_read_u8(u) { asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(a[u]) : "v"(addr), "n"(u * 4)); }
KERNEL(k_ds_read_u8, B_k_ds_read_u8, LGKM)
#define B_k_ds_read_u8_b1(u) { asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(a[u]) : "v"(addr >> 2), "n"(u * 48)); }
KERNEL(k_ds_read_u8_b1, B_k_ds_read_u8_b1, LGKM)
#define B_k_ds_read_b32(u) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[u]) : "v"(addr), "n"(u * 4)); }
KERNEL(k_ds_read_b32, B_k_ds_read_b32, LGKM)
#define B_k_ds_read_b64(u) { unsigned long long q64; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q64) : "v"(addr * 2), "n"(u * 8)); }
KERNEL(k_ds_read_b64, B_k_ds_read_b64, LGKM)
#define B_k_ds_write_b8(u) { asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(addr), "v"(a[u]), "n"(u * 4)); }
KERNEL(k_ds_write_b8, B_k_ds_write_b8, LGKM)
#define B_k_ds_write_b16(u) { asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(addr), "v"(a[u]), "n"(u * 4)); }
KERNEL(k_ds_write_b16, B_k_ds_write_b16, LGKM)
// the FAST pass-1 / score-loop mixes (16 instructions per trip of the same proportions): what the kernel's stream looks like
#define B_k_mix_pass1(u) { if ((u & 3) == 0) { V3("v_perm_b32", u); } else if ((u & 3) == 3 && (u & 4)) { V2("v_mbcnt_lo_u32_b32", u); } else { V2("v_pk_max_i16", u); } }
KERNEL(k_mix_pass1, B_k_mix_pass1, NOTAIL)
#define B_k_mix_score(u) { if (u < 5) { asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(a[u]) : "v"(addr >> 2), "n"(u * 48)); } else { V2("v_pk_min_u16", u); } }
KERNEL(k_mix_score, B_k_mix_score, LGKM)

typedef void (*kern_t)(uint32_t *, unsigned long long *, uint32_t);

__global__ void k_warm(uint32_t *out, int n)
{
    uint32_t a = threadIdx.x;
    for (int i = 0; i < n; i++) a = a * 1664525u + 1013904223u;
    if (a == 0x12345) out[1] = a;
}

int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    (void)hipMalloc((void **)&out, 64);
    (void)hipMalloc((void **)&cyc, 256 * 8 * 4 * 4 * sizeof(unsigned long long));
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int ninstr = N_IT * 256;
    printf("# %s, %d CUs, clockRate %d kHz; %d wave-instructions per wave per launch\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, ninstr);
    printf("# per column: shader cycles per wave-instruction PER SIMD, aggregate (res = waves resident together on a SIMD, shader clock, w/a = the median wave's own cycles per instruction)\n");
    printf("# %-16s %38s %38s %38s %38s\n", "workgroups/CU:", "1", "2", "4", "8");
    struct { const char *name; kern_t k; } tests[] = {
#define T(K) {#K, K}
        T(k_add_u32), T(k_and_b32), T(k_xor_b32), T(k_sub_u32), T(k_lshlrev_b32), T(k_mov_b32), T(k_cndmask), T(k_add_f32), T(k_fma_f32), T(k_max_f32),
        T(k_max_u32), T(k_min_u32), T(k_max_i32), T(k_max_u16), T(k_add3_u32), T(k_and_or_b32), T(k_pk_add_u16), T(k_max3_u32), T(k_med3_i32), T(k_pk_max_i16), T(k_pk_min_u16), T(k_pk_sub_i16),
        T(k_pk_lshr_b16), T(k_perm_b32), T(k_alignbyte), T(k_mbcnt_lo), T(k_mbcnt_hi), T(k_bcnt), T(k_lshl_add), T(k_mad_u32_u24),
        T(k_mul_lo_u32), T(k_dot4_u32_u8), T(k_dot2_u32_u16), T(k_add_sdwa), T(k_cmp_gt_i16), T(k_cmp_gt_u32), T(k_readlane),
        T(k_ds_read_u8), T(k_ds_read_u8_b1), T(k_ds_read_b32), T(k_ds_read_b64), T(k_ds_write_b8), T(k_ds_write_b16),
        T(k_mix_pass1), T(k_mix_score)};
    const int ncu = prop.multiProcessorCount;
    int nsimd = 0;
    hipLaunchKernelGGL(k_warm, dim3(ncu * 8), dim3(256), 0, 0, out, 2000000);   // clocks up before the first measurement
    (void)hipDeviceSynchronize();
    for (auto &t : tests) {
        printf("  %-16s", t.name + 2);
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int blocks = ncu * wps;
            const size_t shm = (size_t)(160.0 * 1024 / (wps + 0.5)) & ~(size_t)255;   // wps workgroups fit a CU, wps + 1 do not
            (void)hipFuncSetAttribute((const void *)t.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), shm, 0, out, cyc, 12345u);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), shm, 0, out, cyc, 12345u);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", t.name); return 1; }
            std::vector<unsigned long long> h((size_t)blocks * 16);
            (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            const size_t nw = (size_t)blocks * 4;
            std::vector<double> cpi(nw), ghz(nw);
            unsigned long long s_min = ~0ull, e_max = 0;
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> simd;   // SIMD -> (time, +1 / -1)
            for (size_t w = 0; w < nw; w++) {
                const unsigned long long dm = h[w * 4], r0 = h[w * 4 + 1], r1 = h[w * 4 + 2], id = h[w * 4 + 3];
                cpi[w] = (double)dm / ninstr;
                ghz[w] = (double)dm / (double)(r1 - r0) * 0.1;
                s_min = std::min(s_min, r0);
                e_max = std::max(e_max, r1);
                const unsigned long long key = ((id >> 32) & 0xf) << 16 | (id & 0xfff0);   // XCC, SE, SH, CU, PIPE, SIMD (wave slot masked out)
                simd[key].push_back({r0, +1});
                simd[key].push_back({r1, -1});
            }
            std::vector<int> resid;
            for (auto &kv : simd) {
                std::sort(kv.second.begin(), kv.second.end());
                int cur = 0, mx = 0;
                for (auto &ev : kv.second) { cur += ev.second; mx = std::max(mx, cur); }
                resid.push_back(mx);
            }
            std::sort(resid.begin(), resid.end());
            std::sort(cpi.begin(), cpi.end());
            std::sort(ghz.begin(), ghz.end());
            const double clk = ghz[nw / 2];
            const double agg = (double)(e_max - s_min) * 10.0 * clk * (double)simd.size() / ((double)nw * ninstr);   // 100 MHz ticks -> shader cycles
            printf("  %5.2f (res %d, %4.2f GHz, w/a %5.2f)", agg, resid[resid.size() / 2], clk, cpi[nw / 2]);
            if (wps == 1 && &t == &tests[0]) nsimd = (int)simd.size();
        }
        printf("\n");
        fflush(stdout);
    }
    printf("# distinct SIMDs seen in HW_ID / XCC_ID: %d\n", nsimd);
    return 0;
}
