// Launch / completion latency probe (profiling aid, not part of the product): where the time between the host's submit and its
// wake-up goes for a job of ~16 short dependent kernels (one rig frame at a time).
//   a) host submit -> first instruction of a kernel (it raises a flag in host-mapped memory), launched directly / as a graph's first node
//   b) last kernel's flag -> hipEventSynchronize returns (completion detection through the event) vs spinning on the flag itself
//   c) a chain of N dependent ~5 us kernels, launch by launch vs replayed from a graph: submit -> last flag
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <immintrin.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_flag(volatile int *flag, int v, int spin)
{
    if (flag) { *flag = v; __threadfence_system(); }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }   // 100 MHz: spin = 500 -> 5 us
}
__global__ void k_flag_end(volatile int *flag, int v, int spin)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    __threadfence_system();
    if (flag) *flag = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main()
{
    volatile int *flag;
    CHECK(hipHostMalloc((void **)&flag, 4096, hipHostMallocMapped | hipHostMallocPortable));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev;
    CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipEvent_t evt;
    CHECK(hipEventCreate(&evt));
    const int R = 300;
    // a) direct launch -> first instruction
    std::vector<double> a, b1, b2, b3;
    for (int r = 0; r < R + 20; r++) {
        flag[0] = 0; flag[16] = 0;
        const double t0 = now_us();
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, 1, 500);
        hipLaunchKernelGGL(k_flag_end, dim3(1), dim3(64), 0, st, flag + 16, 1, 100);
        CHECK(hipEventRecord(r & 1 ? ev : evt, st));
        while (!flag[0]) _mm_pause();
        const double t1 = now_us();
        while (!flag[16]) _mm_pause();
        const double t2 = now_us();
        CHECK(hipEventSynchronize(r & 1 ? ev : evt));
        const double t3 = now_us();
        if (r >= 20) { a.push_back(t1 - t0); (r & 1 ? b1 : b2).push_back(t3 - t2); b3.push_back(t2 - t1); }
    }
    printf("a) direct launch -> kernel's first instruction seen by the host: %.1f us (two launches + event record included in the submit)\n", med(a));
    printf("b) last kernel's end flag seen -> hipEventSynchronize returns: %.1f us (event without timing), %.1f us (with timing); first flag -> end flag %.1f us (6 us of kernels)\n", med(b1), med(b2), med(b3));
    // b') hipEventQuery polling
    {
        std::vector<double> q;
        for (int r = 0; r < R; r++) {
            flag[16] = 0;
            hipLaunchKernelGGL(k_flag_end, dim3(1), dim3(64), 0, st, flag + 16, 1, 500);
            CHECK(hipEventRecord(ev, st));
            while (!flag[16]) _mm_pause();
            const double t2 = now_us();
            while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause();
            q.push_back(now_us() - t2);
        }
        printf("b') end flag seen -> hipEventQuery says done: %.1f us\n", med(q));
    }
    // c) chains
    for (int N : {1, 8, 16, 24}) {
        std::vector<double> d0, d1, g0, g1, gs;
        for (int r = 0; r < R; r++) {
            flag[0] = 0; flag[16] = 0;
            const double t0 = now_us();
            for (int i = 0; i < N; i++) {
                if (i == N - 1) hipLaunchKernelGGL(k_flag_end, dim3(1), dim3(64), 0, st, flag + 16, 1, 500);
                else hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, i == 0 ? flag : (volatile int *)nullptr, 1, 500);
            }
            if (N > 1) while (!flag[0]) _mm_pause();
            const double t1 = now_us();
            while (!flag[16]) _mm_pause();
            const double t2 = now_us();
            d0.push_back(t1 - t0); d1.push_back(t2 - t0);
            CHECK(hipStreamSynchronize(st));
        }
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) {
            if (i == N - 1) hipLaunchKernelGGL(k_flag_end, dim3(1), dim3(64), 0, st, flag + 16, 1, 500);
            else hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, i == 0 ? flag : (volatile int *)nullptr, 1, 500);
        }
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int r = 0; r < R + 5; r++) {
            flag[0] = 0; flag[16] = 0;
            const double t0 = now_us();
            CHECK(hipGraphLaunch(exec, st));
            const double ts = now_us();
            if (N > 1) while (!flag[0]) _mm_pause();
            const double t1 = now_us();
            while (!flag[16]) _mm_pause();
            const double t2 = now_us();
            if (r >= 5) { g0.push_back(t1 - t0); g1.push_back(t2 - t0); gs.push_back(ts - t0); }
            CHECK(hipStreamSynchronize(st));
        }
        printf("c) chain of %2d x 5 us kernels: launch by launch first flag %.1f us, end flag %.1f us | graph: launch call %.1f us, first flag %.1f us, end flag %.1f us\n",
               N, N > 1 ? med(d0) : 0.0, med(d1), med(gs), N > 1 ? med(g0) : 0.0, med(g1));
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
    }
    return 0;
}
