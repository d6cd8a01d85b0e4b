// fetch_calib.hip -- known-answer kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950.
//
// Every kernel streams a buffer of N bytes (default 1 GiB, far beyond the 256 MiB Infinity Cache and the 32 MiB of L2)
// exactly once, in one of the access shapes the mcorb kernels use:
//   read16     16 B per lane, 16-byte aligned (global_load_dwordx4)         -- k_blur / k_resize window fills
//   read16u4   16 B per lane at 4-byte-aligned, not 16-byte-aligned addresses -- k_fast_cells / k_describe chunks
//   read4      4 B per lane (global_load_dword)
//   read1      1 B per lane (global_load_ubyte), N/16 bytes only
//   rows32     32-byte row segments 1 KiB apart (two lanes per row)          -- k_describe's 27 x 32-B patch rows
//   write16 / write4   streaming stores of N bytes
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and, separately, `--pmc WRITE_SIZE`) and compare the
// counter (KiB) with the byte count each kernel prints.  Build: hipcc -O3 --offload-arch=gfx950 (tools/Makefile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

struct __attribute__((packed, aligned(4))) Chunk { uint32_t a, b, c, d; };

__global__ __launch_bounds__(256) void read16(const uint4 *__restrict__ p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;   // never true for the fill pattern; keeps the loads alive
}

__global__ __launch_bounds__(256) void read16u4(const uint8_t *__restrict__ p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 1 < n16; i += (size_t)gridDim.x * 256) {
        const Chunk v = *reinterpret_cast<const Chunk *>(p + 16 * i + 4);
        acc += v.a ^ v.b ^ v.c ^ v.d;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read4(const uint32_t *__restrict__ p, size_t n4, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read1(const uint8_t *__restrict__ p, size_t n1, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

// lane pair (2r, 2r+1) reads the 32-byte segment at row r * 1024: 32 B used out of every 1 KiB
__global__ __launch_bounds__(256) void rows32(const uint8_t *__restrict__ p, size_t nrows, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * nrows; i += (size_t)gridDim.x * 256) {
        const Chunk v = *reinterpret_cast<const Chunk *>(p + (i >> 1) * 1024 + 16 * (i & 1) + 4);
        acc += v.a ^ v.b ^ v.c ^ v.d;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void write16(uint4 *__restrict__ p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = uint4{(uint32_t)i, 1u, 2u, 3u};
}

__global__ __launch_bounds__(256) void write4(uint32_t *__restrict__ p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)i;
}

int main(int argc, char **argv)
{
    const size_t N = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1024) << 20;   // MiB
    uint8_t *buf = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc((void **)&buf, N + 4096));
    CHECK(hipMalloc((void **)&sink, 64));
    CHECK(hipMemset(buf, 0x5a, N + 4096));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto report = [&](const char *name, size_t bytes, int rep) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s bytes_per_launch %zu  (%.1f KiB)  %.3f ms/launch  %.1f GB/s\n", name, bytes, bytes / 1024.0, ms / rep,
               bytes / (ms / rep * 1e-3) / 1e9);
    };
    const int REP = 3;
#define RUN(NAME, BYTES, ...)                                     \
    do {                                                          \
        CHECK(hipEventRecord(e0, 0));                             \
        for (int r = 0; r < REP; r++) hipLaunchKernelGGL(NAME, dim3(grid), dim3(256), 0, 0, __VA_ARGS__); \
        CHECK(hipEventRecord(e1, 0));                             \
        CHECK(hipEventSynchronize(e1));                           \
        CHECK(hipGetLastError());                                 \
        report(#NAME, BYTES, REP);                                \
    } while (0)
    RUN(read16, N, reinterpret_cast<const uint4 *>(buf), N / 16, sink);
    RUN(read16u4, N - 16, buf, N / 16, sink);
    RUN(read4, N, reinterpret_cast<const uint32_t *>(buf), N / 4, sink);
    RUN(read1, N / 16, buf, N / 16, sink);
    RUN(rows32, (N / 1024) * 32, buf, N / 1024, sink);
    RUN(write16, N, reinterpret_cast<uint4 *>(buf), N / 16);
    RUN(write4, N, reinterpret_cast<uint32_t *>(buf), N / 4);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
