// fp4_probe.hip -- what the k-NN kernel needs to know about gfx950's block-scaled FP4 matrix instruction
// (v_mfma_scale_f32_32x32x64_f8f6f4, cbsz = blgp = 4: e2m1 operands) before it is allowed to carry Hamming distances:
//
//  1. EXACTNESS.  Descriptor bits as e2m1 +-1 (nibble 0x2 / 0xA), unit or power-of-two block scales, an f32 accumulator that
//     starts at a small integer (the tie-break term): is the result exactly  S * (256 - 2 * hamming) + C  for every element,
//     for S = 1 (no scales), S = 4096 (both scales 2^6) and C up to 8191?  Checked against the CPU on random and on
//     low-entropy descriptors, all 1024 elements of a 32 x 32 tile, with the C/D layout the i8 form uses.
//  2. RATE.  Cycles per instruction per SIMD for a wave issuing it back to back on independent accumulators, next to
//     v_mfma_i32_32x32x32_i8 (what the kernel used until round 4) -- one wave per SIMD and three.
//  3. CO-ISSUE.  The same loop with k vector instructions (v_max3_f32 / v_med3_f32 / v_max_f32 on independent chains)
//     between two matrix instructions, k = 0 .. 32: where the vector port starts to stretch the loop.
// Build: make -C tools; run: tools/_build/fp4_probe > profiles/rNN_fp4_probe.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// 32 bits -> 32 e2m1 nibbles (bit 1 -> -1.0 = 0xA, bit 0 -> +1.0 = 0x2), low bit in the low nibble of dword 0
__device__ __host__ inline void bits_to_fp4(uint32_t w, uint32_t o[4])
{
    for (int q = 0; q < 4; q++) {
        uint32_t x = (w >> (8 * q)) & 0xffu;
        x = (x | (x << 12)) & 0x000f000fu;
        x = (x | (x << 6)) & 0x03030303u;
        x = (x | (x << 3)) & 0x11111111u;
        o[q] = (x << 3) | 0x22222222u;
    }
}

// one wave: 32 train descriptors (A rows) x 32 query descriptors (B columns), four K-steps of 64 bits
template <int MODE>   // 0: no scales; 1: both scales 2^6 (E8M0 133)
__global__ __launch_bounds__(64) void k_exact(const uint32_t *__restrict__ da, const uint32_t *__restrict__ db, const float *__restrict__ cin,
                                              float *__restrict__ out)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v16f acc;
    for (int e = 0; e < 16; e++) acc[e] = cin[e * 64 + lane];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        uint32_t a[4], b[4];
        bits_to_fp4(da[r * 8 + 2 * s + h], a);
        bits_to_fp4(db[r * 8 + 2 * s + h], b);
        const v8i A = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0}, B = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
        if (MODE == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0x85858585, 0, 0x85858585);
    }
    for (int e = 0; e < 16; e++) out[e * 64 + lane] = acc[e];
}

static int popc256(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
    for (int i = 0; i < 8; i++) d += __builtin_popcount(a[i] ^ b[i]);
    return d;
}

static int run_exact(int mode, int kind, uint32_t seed)
{
    std::vector<uint32_t> da(32 * 8), db(32 * 8);
    std::vector<float> cin(1024), out(1024);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (int i = 0; i < 256; i++) {
        da[i] = rnd(); db[i] = rnd();
        if (kind == 1) { da[i] &= rnd() & rnd() & rnd(); db[i] = da[i] ^ (rnd() & rnd() & rnd() & rnd() & rnd()); }   // near-duplicates: small distances
        if (kind == 2) { da[i] = 0xffffffffu; db[i] = 0; }                                                       // distance 256
        if (kind == 3) { da[i] = db[i] = 0x55555555u; }                                                          // distance 0
    }
    for (int e = 0; e < 16; e++)
        for (int l = 0; l < 64; l++) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            cin[e * 64 + l] = (kind & 1) ? (float)(8191 - row * 251 % 8192) : (float)(31 - row);
        }
    uint32_t *d_a, *d_b; float *d_c, *d_o;
    CK(hipMalloc(&d_a, 1024)); CK(hipMalloc(&d_b, 1024)); CK(hipMalloc(&d_c, 4096)); CK(hipMalloc(&d_o, 4096));
    CK(hipMemcpy(d_a, da.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, db.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c, cin.data(), 4096, hipMemcpyHostToDevice));
    if (mode == 0) k_exact<0><<<1, 64>>>(d_a, d_b, d_c, d_o); else k_exact<1><<<1, 64>>>(d_a, d_b, d_c, d_o);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), d_o, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    const double S = mode == 0 ? 1.0 : 4096.0;
    for (int e = 0; e < 16; e++)
        for (int l = 0; l < 64; l++) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
            const double want = S * (256 - 2 * popc256(&da[row * 8], &db[col * 8])) + cin[e * 64 + l];
            if ((double)out[e * 64 + l] != want) {
                if (bad < 4) printf("    mismatch mode %d kind %d: row %d col %d got %.1f want %.1f\n", mode, kind, row, col, out[e * 64 + l], want);
                bad++;
            }
        }
    CK(hipFree(d_a)); CK(hipFree(d_b)); CK(hipFree(d_c)); CK(hipFree(d_o));
    return bad;
}

// ---- rate / co-issue ----
// NV vector instructions between two matrix instructions; 4 independent accumulator sets; FP4: scaled e2m1, else i8 32x32x32
#define VALU_CHAIN(u) asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_med3_f32 %3, %3, %1, %2" : "+v"(va[u]), "+v"(vb[u]) : "v"(x), "v"(y))
template <int FP4, int NV>
__global__ __launch_bounds__(256) void k_rate(float *out, unsigned long long *cyc, int iters, float x, float y)
{
    v16f facc[4];
    v16i iacc[4];
    for (int i = 0; i < 4; i++)
        for (int e = 0; e < 16; e++) { facc[i][e] = (float)(threadIdx.x + e + i); iacc[i][e] = threadIdx.x + e + i; }
    const v8i A = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    const v8i B = {(int)0xaaaa2222, 0x2222aaaa, 0x2a2a2a2a, 0x22222222, 0, 0, 0, 0};
    const v4i A4 = {0x40c040c0, 0x4040c0c0, (int)0xc0c04040, 0x40404040};
    const v4i B4 = {0x4040c0c0, (int)0xc040c040, 0x40404040, (int)0xc0c0c0c0};
    float va[8], vb[8];
    for (int u = 0; u < 8; u++) { va[u] = x + u; vb[u] = y - u; }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (FP4) facc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, facc[i], 4, 4, 0, 0x85858585, 0, 0x85858585);
            else iacc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, iacc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 2; v++) VALU_CHAIN(v & 7);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; i++)
        for (int e = 0; e < 16; e++) s += FP4 ? facc[i][e] : (float)iacc[i][e];
    for (int u = 0; u < 8; u++) s += va[u] + vb[u];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *o = cyc + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
        o[0] = t1 - t0;
        o[1] = r1 - r0;
    }
}

template <int FP4, int NV>
static void run_rate(int wps)
{
    // wps workgroups of 256 threads resident per CU (one wave per SIMD each): LDS request sized so that exactly wps fit
    const int iters = 2000, ncu = 256, nwg = ncu * wps;
    const size_t lds = (size_t)(160 * 1024 / (wps + 0.5));
    float *d_o; unsigned long long *d_c;
    CK(hipMalloc(&d_o, 64)); CK(hipMalloc(&d_c, (size_t)nwg * 4 * 16));
    CK(hipFuncSetAttribute((const void *)k_rate<FP4, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) k_rate<FP4, NV><<<nwg, 256, lds>>>(d_o, d_c, iters, 1.5f, 2.5f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c((size_t)nwg * 8);
    CK(hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> per, ghz;
    for (int w = 0; w < nwg * 4; w++) { per.push_back((double)c[2 * w] / (iters * 4.0)); ghz.push_back((double)c[2 * w] / (double)c[2 * w + 1] * 0.1); }
    std::sort(per.begin(), per.end()); std::sort(ghz.begin(), ghz.end());
    // a wave's own cycles per matrix instruction; with wps waves sharing the SIMD the SIMD's rate is that / wps
    printf("  %-4s NV=%-2d wps=%d: wave cycles per matrix instruction %7.1f (median)  -> per SIMD %6.1f   clock %.2f GHz\n", FP4 ? "fp4" : "i8", NV, wps,
           per[per.size() / 2], per[per.size() / 2] / wps, ghz[ghz.size() / 2]);
    CK(hipFree(d_o)); CK(hipFree(d_c));
}


// ---- the k-NN loop's skeleton: NCH independent accumulator chains per wave, each K = 256 (four steps, the first one taking a
// constant register set as its C input like the kernel's row term), the A operand of a step optionally read from LDS two steps
// ahead (ds_read_b128, 1 KiB per wave and step), no folds, no barrier: what the matrix pipe sustains in that structure ----
template <int NCH, int LDSA>
__global__ __launch_bounds__(256) void k_skel(float *out, unsigned long long *cyc, int iters)
{
    extern __shared__ uint4 lds4[];
    for (int i = threadIdx.x; i < 2048; i += 256) lds4[i] = uint4{0x22222222u, 0x2a2a2a2au + i, 0xa2a2a2a2u, 0x22aa22aau};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v16f Crow, acc[NCH];
    for (int e = 0; e < 16; e++) Crow[e] = (float)(31 - e - (lane >> 5) * 4);
    v4i Bf[NCH][4];
    for (int u = 0; u < NCH; u++)
        for (int s = 0; s < 4; s++) Bf[u][s] = v4i{0x2a2a2a2a + u, 0x22222222 + s, (int)0xa2a2a2a2, 0x22aa22aa};
    float sink = 0;
    const uint4 *S = lds4 + wave * 512 + lane;
    v4i Af[4];
    Af[0] = __builtin_bit_cast(v4i, S[0]); Af[1] = __builtin_bit_cast(v4i, S[64]);
    Af[2] = Af[0]; Af[3] = Af[1];
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 8; g++) {   // two tiles
            const int s = g & 3;
            if (LDSA) Af[(g + 2) & 3] = __builtin_bit_cast(v4i, S[((g + 2) & 7) * 64]);
            else asm volatile("" : "+v"(Af[(g + 2) & 3][0]));
            const v4i a = Af[g & 3];
            const v8i A = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < NCH; u++) {
                const v8i B = {Bf[u][s][0], Bf[u][s][1], Bf[u][s][2], Bf[u][s][3], 0, 0, 0, 0};
                acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, s == 0 ? Crow : acc[u], 4, 4, 0, 0x85858585, 0, 0x85858585);
            }
            if (s == 3) {
#pragma unroll
                for (int u = 0; u < NCH; u++) asm volatile("" :: "v"(acc[u][0]), "v"(acc[u][15]));   // the tile's result is "used"
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    for (int u = 0; u < NCH; u++) sink += acc[u][3];
    if (sink == 12345.678f) out[0] = sink;
    if (lane == 0) {
        unsigned long long *o = cyc + (size_t)(blockIdx.x * 4 + wave) * 2;
        o[0] = t1 - t0;
        o[1] = r1 - r0;
    }
}

template <int NCH, int LDSA>
static void run_skel(int wps)
{
    const int iters = 500, ncu = 256, nwg = ncu * wps;
    const size_t lds = std::max<size_t>((size_t)(160 * 1024 / (wps + 0.5)), 2048 * 16);
    float *d_o; unsigned long long *d_c;
    CK(hipMalloc(&d_o, 64)); CK(hipMalloc(&d_c, (size_t)nwg * 4 * 16));
    CK(hipFuncSetAttribute((const void *)k_skel<NCH, LDSA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) k_skel<NCH, LDSA><<<nwg, 256, lds>>>(d_o, d_c, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c((size_t)nwg * 8);
    CK(hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> per;
    for (int w = 0; w < nwg * 4; w++) per.push_back((double)c[2 * w] / (iters * 8.0 * NCH));
    std::sort(per.begin(), per.end());
    printf("  skeleton chains=%d A-from-%s wps=%d: wave cycles per matrix instruction %6.1f -> per SIMD %6.1f\n", NCH, LDSA ? "LDS" : "reg", wps,
           per[per.size() / 2], per[per.size() / 2] / wps);
    CK(hipFree(d_o)); CK(hipFree(d_c));
}

int main()
{
    printf("fp4_probe: v_mfma_scale_f32_32x32x64_f8f6f4 (e2m1 x e2m1) as a Hamming-distance engine\n\n1. exactness (mismatching elements of 1024 per case; 0 = exact)\n");
    int total = 0;
    for (int mode = 0; mode < 2; mode++)
        for (int kind = 0; kind < 4; kind++) {
            int bad = 0;
            for (uint32_t seed = 1; seed <= 16; seed++) bad += run_exact(mode, kind, seed);
            printf("  scales %-6s kind %d (%s): %d mismatches in 16 tiles\n", mode ? "2^6,2^6" : "none", kind,
                   kind == 0 ? "random, C = 31 - row" : kind == 1 ? "near-duplicates, C up to 8191" : kind == 2 ? "distance 256" : "distance 0, C up to 8191", bad);
            total += bad;
        }
    printf("  => %s\n\n2./3. rate and co-issue (cycles per matrix instruction; NV = vector instructions between two of them)\n", total ? "NOT EXACT" : "exact in every case");
    run_rate<0, 0>(1); run_rate<1, 0>(1); run_rate<0, 0>(3); run_rate<1, 0>(3);
    run_rate<1, 8>(1);  run_rate<1, 16>(1); run_rate<1, 24>(1); run_rate<1, 32>(1);
    run_rate<1, 8>(3);  run_rate<1, 12>(3); run_rate<1, 16>(3); run_rate<1, 24>(3); run_rate<1, 32>(3);
    run_rate<0, 16>(3); run_rate<0, 32>(3);
    printf("\n4. the k-NN loop's skeleton (K = 256 chains, first step with a constant C input)\n");
    run_skel<1, 0>(1); run_skel<2, 0>(1); run_skel<4, 0>(1); run_skel<1, 0>(3); run_skel<2, 0>(3); run_skel<2, 0>(2); run_skel<4, 0>(2);
    run_skel<2, 1>(1); run_skel<4, 1>(1); run_skel<2, 1>(3); run_skel<2, 1>(2); run_skel<4, 1>(2);
    return total ? 1 : 0;
}
