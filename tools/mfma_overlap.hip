// mfma_overlap.hip -- do v_mfma_i32_32x32x32_i8 and ordinary vector instructions overlap on gfx950?
// Three kernels, each 4 waves per workgroup (one per SIMD), W workgroups per CU:
//   mfma   : two independent accumulate chains of MFMAs
//   valu   : the same number of (v_med3 + v_max) pairs k_knn2 spends per MFMA pair (eight vector instructions per two MFMAs)
//   both   : the two interleaved in program order (2 MFMA, 8 VALU, ...), different registers
// If `both` takes max(mfma, valu) the pipes overlap; if it takes mfma + valu they share the issue port.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define N_IT 2048
template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int seed)
{
    v16i a0 = {0}, a1 = {0};
    v4i A = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, B = {seed * 11, seed + 1, seed + 2, seed + 3};
    int k0[8], k1[8], x = seed + threadIdx.x;
    for (int i = 0; i < 8; i++) { k0[i] = seed + i; k1[i] = seed - i; }
    for (int it = 0; it < N_IT; it++) {
        if (MODE != 1) {
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, a1, 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int m;
                asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(k0[i]), "v"(k1[i]), "v"(x));
                k1[i] = m;
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(m) : "v"(k0[i]), "v"(x));
                k0[i] = m;
            }
        }
        if (MODE == 2) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 16; i++) s ^= a0[i] ^ a1[i];
    for (int i = 0; i < 8; i++) s ^= k0[i] ^ k1[i];
    if (s == 0x1234567) out[0] = s;
}
int main()
{
    int *out;
    hipMalloc((void **)&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"mfma only (2 per iteration)", "valu only (8 per iteration)", "both, interleaved"};
    for (int wps = 1; wps <= 3; wps++)
        for (int mode = 0; mode < 3; mode++) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, 3);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, 3);
                else hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, out, 3);
            };
            launch();
            hipEventRecord(e0, 0);
            for (int r = 0; r < 5; r++) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d  %-30s %.1f cycles per iteration per SIMD (at 2.4 GHz)\n", wps, names[mode], ms * 1e6 * 2.4 / (5.0 * N_IT * wps) / 1.0);
        }
    return 0;
}
