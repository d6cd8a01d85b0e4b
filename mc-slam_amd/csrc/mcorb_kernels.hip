// mcorb_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ORB front-end.
//
// One image = one camera frame; every kernel is launched over a whole batch of
// images (grid.y or grid.z = image index) so that launch cost is shared by all
// cameras and frames of a batch.  All arithmetic is integer and bit-exact with
// the reference's OpenCV CPU path as restated in oracle/ (see DESIGN.md).
// Nothing here is GEMM-shaped: no MFMA.  The levers are LDS tiles, packed
// 16-bit min/max, wave64 ballots and popcounts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcorb_common.h"
#include "mcorb_kernels.h"

namespace mcorb {

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int reflect101(int p, int len)
{
    // cv::borderInterpolate(BORDER_REFLECT_101); |overshoot| < len here
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------
// Frame hand-off: CV_32F [0,1] (1 or 3 channels, BGR) -> u8 gray level-0 plane.
// multiply(img,255) -> convertTo(CV_8U) -> cvtColor(BGR2GRAY)
// (MCSlam/src/MultiCameraFrame.cpp:108-116).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int sat_u8_rne(float v)
{
    int r = __float2int_rn(v);   // cvRound: round-half-even
    return r < 0 ? 0 : (r > 255 ? 255 : r);
}

__global__ __launch_bounds__(256) void k_stage_f32(const float *__restrict__ src, int w, int h, int src_pitch_f,
                                                   int channels, size_t src_img_stride_f, uint8_t *__restrict__ pyr,
                                                   Geom g)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const float *S = src + (size_t)img * src_img_stride_f + (size_t)y * src_pitch_f;
    int v;
    if (channels == 1) {
        v = sat_u8_rne(__fmul_rn(S[x], 255.f));
    } else {
        int b = sat_u8_rne(__fmul_rn(S[3 * x + 0], 255.f));
        int gg = sat_u8_rne(__fmul_rn(S[3 * x + 1], 255.f));
        int r = sat_u8_rne(__fmul_rn(S[3 * x + 2], 255.f));
        v = (b * 1868 + gg * 9617 + r * 4899 + 8192) >> 14;
    }
    pyr[(size_t)img * g.imgBytes + g.lv[0].off + (size_t)y * g.lv[0].pitch + x] = (uint8_t)v;
}

// ---------------------------------------------------------------------------
// Pyramid: level L from level L-1, cv::resize INTER_LINEAR 8UC1 fixed point
// (ComputePyramid, ORBextractor.cpp:1173-1198; coefficients: host tables).
// Thread = 4 consecutive output pixels of one row -> one dword store.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(uint8_t *__restrict__ pyr, Geom g, int level,
                                                const ResizeTap *__restrict__ tabs)
{
    const LevelGeom &D = g.lv[level];
    const LevelGeom &S = g.lv[level - 1];
    const int img = blockIdx.z;
    uint8_t *base = pyr + (size_t)img * g.imgBytes;
    const int dx4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dy >= D.h || dx4 >= D.w) return;
    const ResizeTap ty = tabs[D.ytab + dy];
    const uint8_t *S0 = base + S.off + (size_t)ty.s0 * S.pitch;
    const uint8_t *S1 = base + S.off + (size_t)ty.s1 * S.pitch;
    const int b0 = ty.c0, b1 = ty.c1;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const ResizeTap tx = tabs[D.xtab + dx4 + k];   // table is padded to a multiple of 4 entries
        const int a0 = tx.c0, a1 = tx.c1;
        const int R0 = S0[tx.s0] * a0 + S0[tx.s1] * a1;
        const int R1 = S1[tx.s0] * a0 + S1[tx.s1] * a1;
        const int v = (((b0 * (R0 >> 4)) >> 16) + ((b1 * (R1 >> 4)) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 0xff) << (8 * k);
    }
    if (dx4 + 4 > D.w) out &= 0xffffffffu >> (8 * (dx4 + 4 - D.w));   // keep the row padding zero
    *reinterpret_cast<uint32_t *>(base + D.off + (size_t)dy * D.pitch + dx4) = out;
}

// ---------------------------------------------------------------------------
// FAST-9/16 corner score + cell-local 3x3 non-max suppression, one workgroup
// per 35-px cell (ComputeKeyPointsOctTree detection loop, ORBextractor.cpp:
// 804-871, calling cv::FAST on each cell ROI).
//
// For a pixel v with ring r[0..15], d[k] = v - r[k]:
//   A = max over the 16 contiguous 9-arcs of min(d) (dark) or min(-d) (bright)
// p is a FAST corner at threshold t  <=>  A > t, and cornerScore = A - 1
// independently of t.  So one pass computes A for every pixel; the reference's
// "retry the cell with minThFAST if iniThFAST gave no keypoint" is a second
// NMS pass over the same A map.  The dark and bright arcs are evaluated
// together in packed 16-bit lanes (v_pk_min_i16 / v_pk_max_i16).
// ---------------------------------------------------------------------------
constexpr int kFastMaxIts = 24;   // ceil(70*70/256) = 20

__device__ __forceinline__ int fast_arc_max(const uint8_t *c)
{
    // ring offsets (dx,dy), k = 0..15 (cv::FAST makeOffsets, patternSize 16)
    constexpr int TP = kTilePitch;
    const int v = c[0];
    int r[16];
    r[0] = c[3 * TP];        r[1] = c[3 * TP + 1];   r[2] = c[2 * TP + 2];   r[3] = c[TP + 3];
    r[4] = c[3];             r[5] = c[-TP + 3];      r[6] = c[-2 * TP + 2];  r[7] = c[-3 * TP + 1];
    r[8] = c[-3 * TP];       r[9] = c[-3 * TP - 1];  r[10] = c[-2 * TP - 2]; r[11] = c[-TP - 3];
    r[12] = c[-3];           r[13] = c[TP - 3];      r[14] = c[2 * TP - 2];  r[15] = c[3 * TP - 1];
    s16x2 p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int d = v - r[k];
        p[k] = s16x2{(short)d, (short)(-d)};
    }
    s16x2 m2[16], m4[16], m8[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m2[k] = __builtin_elementwise_min(p[k], p[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m4[k] = __builtin_elementwise_min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m8[k] = __builtin_elementwise_min(m4[k], m4[(k + 4) & 15]);
    s16x2 best = __builtin_elementwise_min(m8[0], p[8]);
#pragma unroll
    for (int k = 1; k < 16; k++) best = __builtin_elementwise_max(best, __builtin_elementwise_min(m8[k], p[(k + 8) & 15]));
    const int a = best.x > best.y ? best.x : best.y;
    return a < 0 ? 0 : a;   // <= 255
}

__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ pyr, Geom g, int iniTh, int minTh,
                                                    uint32_t *__restrict__ cell_kp, int *__restrict__ cell_cnt)
{
    constexpr int TP = kTilePitch;
    __shared__ __attribute__((aligned(16))) uint8_t tile[kMaxRoi * TP];
    __shared__ __attribute__((aligned(16))) uint8_t sc[kMaxRoi * TP];
    __shared__ int wcnt[kFastMaxIts * 4 + 4];
    __shared__ int s_total;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cell = blockIdx.x, img = blockIdx.y;
    int level = 0;
#pragma unroll 1
    for (int l = 1; l < g.nlevels; l++)
        if (cell >= g.lv[l].cell0) level = l;
    const LevelGeom &L = g.lv[level];
    const int cl = cell - L.cell0;
    const int ci = cl / L.nCols, cj = cl - ci * L.nCols;
    int *out_cnt = cell_cnt + (size_t)img * g.cells + cell;

    // cell ROI exactly as the reference builds it (float there, exact in int)
    const int iniY = kMinBorder + ci * L.hCell;
    int maxY = iniY + L.hCell + 6;
    const int iniX = kMinBorder + cj * L.wCell;
    int maxX = iniX + L.wCell + 6;
    const bool skip = (iniY >= L.maxBorderY - 3) || (iniX >= L.maxBorderX - 6);
    if (maxY > L.maxBorderY) maxY = L.maxBorderY;
    if (maxX > L.maxBorderX) maxX = L.maxBorderX;
    const int cols = maxX - iniX, rows = maxY - iniY;
    const int wi = cols - 6, hi = rows - 6;   // FAST evaluates ROI columns 3..cols-4, rows 3..rows-4
    if (skip || wi <= 0 || hi <= 0) {
        if (tid == 0) *out_cnt = 0;
        return;
    }

    // ---- stage the ROI in LDS with aligned dword loads (same byte phase as HBM) ----
    const uint8_t *plane = pyr + (size_t)img * g.imgBytes + L.off;
    const int ph = iniX & 3;
    const int nd = (ph + cols + 3) >> 2;   // dwords per row, <= 20
    {
        const uint8_t *src0 = plane + (size_t)iniY * L.pitch + (iniX - ph);
        uint32_t *t32 = reinterpret_cast<uint32_t *>(tile);
        uint32_t *s32 = reinterpret_cast<uint32_t *>(sc);
        for (int i = tid; i < rows * (TP / 4); i += 256) {
            const int y = i / (TP / 4), d = i - y * (TP / 4);
            s32[i] = 0;
            if (d < nd) t32[i] = *reinterpret_cast<const uint32_t *>(src0 + (size_t)y * L.pitch + 4 * d);
        }
    }
    __syncthreads();

    // ---- arc score A for every interior pixel ----
    const int ni = wi * hi;
    const int nits = (ni + 255) >> 8;
    for (int it = 0; it < nits; it++) {
        const int p = it * 256 + tid;
        if (p < ni) {
            const int y = p / wi, x = p - y * wi;
            const int a = fast_arc_max(tile + (y + 3) * TP + ph + x + 3);
            sc[(y + 3) * TP + x + 3] = (uint8_t)a;
        }
    }
    __syncthreads();

    // ---- NMS at iniTh, then (only if the cell came out empty) at minTh ----
    uint32_t keepmask = 0;
    int T = iniTh;
    for (int attempt = 0; attempt < 2; attempt++) {
        keepmask = 0;
        int mine = 0;
        for (int it = 0; it < nits; it++) {
            const int p = it * 256 + tid;
            bool keep = false;
            if (p < ni) {
                const int y = p / wi, x = p - y * wi;
                const uint8_t *s = sc + (y + 3) * TP + x + 3;
                const int a = s[0];
                if (a > T) {
                    const int e = a - 1;   // cornerScore
#define EFF(q) ((int)(q) > T ? (int)(q)-1 : 0)
                    keep = e > EFF(s[1]) && e > EFF(s[-1]) && e > EFF(s[-TP - 1]) && e > EFF(s[-TP]) &&
                           e > EFF(s[-TP + 1]) && e > EFF(s[TP - 1]) && e > EFF(s[TP]) && e > EFF(s[TP + 1]);
#undef EFF
                }
            }
            const unsigned long long b = __ballot(keep);
            if (lane == 0) wcnt[it * 4 + wave] = __popcll(b);
            if (keep) { keepmask |= 1u << it; mine++; }
        }
        const int total = __syncthreads_count(mine > 0) ? 1 : 0;
        if (total || T == minTh || attempt == 1) break;
        T = minTh;
        __syncthreads();
    }

    // ---- exclusive scan of the (iteration, wave) counts by wave 0 ----
    const int nc = nits * 4;
    if (wave == 0) {
        int carry = 0;
        for (int base = 0; base < nc; base += 64) {
            const int i = base + lane;
            const int v = i < nc ? wcnt[i] : 0;
            int s = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(s, o);
                if (lane >= o) s += t;
            }
            if (i < nc) wcnt[i] = carry + s - v;
            carry += __shfl(s, 63);
        }
        if (lane == 0) s_total = carry;
    }
    __syncthreads();
    const int total = s_total;
    if (tid == 0) *out_cnt = total;
    if (total == 0) return;

    // ---- ordered emission: raster order inside the cell == cv::FAST's output order ----
    uint32_t *dst = cell_kp + ((size_t)img * g.cells + cell) * g.cellCap;
    for (int it = 0; it < nits; it++) {
        const bool keep = (keepmask >> it) & 1u;
        const unsigned long long b = __ballot(keep);
        if (keep) {
            const int pos = wcnt[it * 4 + wave] + __popcll(b & ((1ull << lane) - 1ull));
            const int p = it * 256 + tid;
            const int y = p / wi, x = p - y * wi;
            const int a = sc[(y + 3) * TP + x + 3];
            // keypoint coordinates relative to minBorder: FAST's ROI coordinate + cell origin (:864-865)
            if (pos < g.cellCap) dst[pos] = pack_cand(x + 3 + cj * L.wCell, y + 3 + ci * L.hCell, a - 1);
        }
    }
}

// ---------------------------------------------------------------------------
// Candidate compaction: per-cell lists -> one contiguous list per image in
// (level, cell row, cell col, raster) order == vToDistributeKeys order of every
// level back to back.  Written straight into host-mapped memory; lvl_off gets
// nlevels+1 offsets per image.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(const uint32_t *__restrict__ cell_kp, const int *__restrict__ cell_cnt,
                                                 Geom g, uint32_t *__restrict__ cand, int *__restrict__ lvl_off,
                                                 int *__restrict__ overflow)
{
    __shared__ int red[4];
    const int tid = threadIdx.x;
    const int cell = blockIdx.x, img = blockIdx.y;
    const int *cnt = cell_cnt + (size_t)img * g.cells;
    int s = 0;
    for (int c = tid; c < cell; c += 256) s += cnt[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const int offset = red[0] + red[1] + red[2] + red[3];
    int n = cnt[cell];
    if (n > g.cellCap) n = g.cellCap;
    if (tid == 0) {
        for (int l = 0; l < g.nlevels; l++)
            if (cell == g.lv[l].cell0) lvl_off[(size_t)img * (kMaxLevels + 1) + l] = offset;
        if (cell == g.cells - 1) lvl_off[(size_t)img * (kMaxLevels + 1) + g.nlevels] = offset + n;
        if (offset + n > g.candCap || cnt[cell] > g.cellCap) atomicOr(overflow, 1);
    }
    const uint32_t *src = cell_kp + ((size_t)img * g.cells + cell) * g.cellCap;
    uint32_t *dst = cand + (size_t)img * g.candCap;
    for (int k = tid; k < n; k += 256)
        if (offset + k < g.candCap) dst[offset + k] = src[k];
}

// ---------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, OpenCV >= 4 fixed-point path: taps {18,34,48,56,48,34,18}/256,
// horizontal pass in 8.8 (u16), vertical pass in 16.16, (sum + 32768) >> 16,
// BORDER_REFLECT_101 on the un-bordered level (ORBextractor.cpp:1132-1133).
// Workgroup = 64x16 output tile staged through LDS.
// ---------------------------------------------------------------------------
constexpr int kBlurTW = 64, kBlurTH = 16;

__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, Geom g)
{
    __shared__ uint8_t in[(kBlurTH + 6) * (kBlurTW + 8)];
    __shared__ uint16_t hb[(kBlurTH + 6) * kBlurTW];
    constexpr int IP = kBlurTW + 8;
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    int level = 0;
#pragma unroll 1
    for (int l = 1; l < g.nlevels; l++)
        if ((int)blockIdx.x >= g.lv[l].tile0) level = l;
    const LevelGeom &L = g.lv[level];
    const int t = blockIdx.x - L.tile0;
    const int ty = t / L.tilesX, tx = t - ty * L.tilesX;
    const int x0 = tx * kBlurTW, y0 = ty * kBlurTH;
    const uint8_t *plane = pyr + (size_t)img * g.imgBytes + L.off;

    for (int i = tid; i < (kBlurTH + 6) * (kBlurTW + 6); i += 256) {
        const int r = i / (kBlurTW + 6), c = i - r * (kBlurTW + 6);
        const int sy = reflect101(y0 + r - 3, L.h), sx = reflect101(x0 + c - 3, L.w);
        // tiles at the right/bottom edge over-cover; clamp keeps the address inside the plane
        const int cy = sy < 0 ? 0 : (sy >= L.h ? L.h - 1 : sy);
        const int cx = sx < 0 ? 0 : (sx >= L.w ? L.w - 1 : sx);
        in[r * IP + c] = plane[(size_t)cy * L.pitch + cx];
    }
    __syncthreads();
    for (int i = tid; i < (kBlurTH + 6) * kBlurTW; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint8_t *s = in + r * IP + c;
        hb[i] = (uint16_t)(18 * (s[0] + s[6]) + 34 * (s[1] + s[5]) + 48 * (s[2] + s[4]) + 56 * s[3]);
    }
    __syncthreads();
    const int c = tid & 63, rq = tid >> 6;
    const int x = x0 + c;
    uint8_t *oplane = blur + (size_t)img * g.imgBytes + L.off;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = rq * 4 + k;
        const uint16_t *s = hb + r * kBlurTW + c;
        const uint32_t acc = 18u * (s[0] + s[6 * kBlurTW]) + 34u * (s[kBlurTW] + s[5 * kBlurTW]) +
                             48u * (s[2 * kBlurTW] + s[4 * kBlurTW]) + 56u * s[3 * kBlurTW];
        const int y = y0 + r;
        if (x < L.w && y < L.h) oplane[(size_t)y * L.pitch + x] = (uint8_t)((acc + 32768u) >> 16);
    }
}

// ---------------------------------------------------------------------------
// 256-bit BRIEF, one wavefront per keypoint (computeOrbDescriptor,
// ORBextractor.cpp:105-145).  Lane l evaluates test pairs l, 64+l, 128+l, 192+l;
// a 64-wide ballot of "t0 < t1" IS eight descriptor bytes (bit k of byte i =
// pair 8i+k), so four ballots produce the descriptor with no shuffles.
// With angle = 0 (reference behaviour) the taps are the raw pattern offsets;
// with IC_Angle enabled they are rotated per keypoint as in :110-118.
// ---------------------------------------------------------------------------
__constant__ int8_t c_pattern[1024] = {
#include "brief_pattern_31.inc"
};
__constant__ int c_umax[16];

// cv::fastAtan2 scalar polynomial
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__global__ __launch_bounds__(256) void k_describe(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                  Geom g, const uint32_t *__restrict__ sel,
                                                  const int *__restrict__ nsel, int orientation,
                                                  uint8_t *__restrict__ desc, float *__restrict__ angles)
{
    const int lane = lane_id();
    const int img = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nsel[img]) return;
    const uint32_t s = sel[(size_t)img * g.kcap + k];
    const int level = (int)(s >> 28), ky = (int)((s >> 14) & 0x3fffu), kx = (int)(s & 0x3fffu);
    const LevelGeom &L = g.lv[level];
    const int pitch = L.pitch;
    const size_t coff = (size_t)img * g.imgBytes + L.off + (size_t)ky * pitch + kx;

    float ca = 1.f, sa = 0.f;
    if (orientation) {
        // IC_Angle (ORBextractor.cpp:75-102) on the un-blurred level; lanes share the 31 rows
        const uint8_t *c = pyr + coff;
        int m01 = 0, m10 = 0;
        if (lane < 31) {
            const int u = lane - 15;
            m10 = u * c[u];
        }
        for (int v = 1; v <= 15; v++) {
            const int d = c_umax[v];
            if (lane <= 2 * d) {
                const int u = lane - d;
                const int vp = c[u + v * pitch], vm = c[u - v * pitch];
                m01 += v * (vp - vm);
                m10 += u * (vp + vm);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m01 += __shfl_xor(m01, o);
            m10 += __shfl_xor(m10, o);
        }
        const float ang = fast_atan2_deg((float)m01, (float)m10);
        if (lane == 0) angles[(size_t)img * g.kcap + k] = ang;
        const float rad = __fmul_rn(ang, (float)(3.14159265358979323846 / 180.f));
        ca = cosf(rad);
        sa = sinf(rad);
    }

    const uint8_t *c = blur + coff;
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = (j * 64 + lane) * 4;
        const int x0 = c_pattern[p], y0 = c_pattern[p + 1], x1 = c_pattern[p + 2], y1 = c_pattern[p + 3];
        int ox0 = x0, oy0 = y0, ox1 = x1, oy1 = y1;
        if (orientation) {
            // cvRound(x*b + y*a), cvRound(x*a - y*b) with a = cos, b = sin (:117-118)
            oy0 = __float2int_rn(__fadd_rn(__fmul_rn((float)x0, sa), __fmul_rn((float)y0, ca)));
            ox0 = __float2int_rn(__fsub_rn(__fmul_rn((float)x0, ca), __fmul_rn((float)y0, sa)));
            oy1 = __float2int_rn(__fadd_rn(__fmul_rn((float)x1, sa), __fmul_rn((float)y1, ca)));
            ox1 = __float2int_rn(__fsub_rn(__fmul_rn((float)x1, ca), __fmul_rn((float)y1, sa)));
        }
        const int t0 = c[oy0 * pitch + ox0], t1 = c[oy1 * pitch + ox1];
        bits[j] = __ballot(t0 < t1);
    }
    if (lane < 4) {
        const unsigned long long b = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long *>(desc + ((size_t)img * g.kcap + k) * 32)[lane] = b;
    }
}

// ---------------------------------------------------------------------------
// All-pairs Hamming k-NN, k = 2 (BFMatcher knnMatch, MultiCameraFrame.cpp:
// 1053-1055).  One query per lane (4 x u64 in registers); a chunk of train
// descriptors is staged in LDS and every lane reads the same descriptor
// (broadcast ds_read_b128).  (distance << 16 | trainIdx) as one u32 key makes
// "two smallest keys" exactly knnMatch's order, including its lowest-index
// tie-break, so chunks can be reduced in any order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_knn2(const uint8_t *__restrict__ desc, const int *__restrict__ counts,
                                             const int2 *__restrict__ pairs, int kcap, int nchunks,
                                             uint2 *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) ulonglong4 tr[kKnnChunk];
    const int lane = threadIdx.x;
    const int pair = blockIdx.z, chunk = blockIdx.y;
    const int2 qt = pairs[pair];
    const int nq = counts[qt.x], nt = counts[qt.y];
    const int q = blockIdx.x * 64 + lane;
    const int t0 = chunk * kKnnChunk;
    if (blockIdx.x * 64 >= nq) return;
    uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu;
    if (t0 < nt) {
        const int tn = nt - t0 < kKnnChunk ? nt - t0 : kKnnChunk;
        const ulonglong4 *tsrc = reinterpret_cast<const ulonglong4 *>(desc + ((size_t)qt.y * kcap + t0) * 32);
        for (int i = lane; i < tn; i += 64) tr[i] = tsrc[i];
        __syncthreads();
        ulonglong4 qv = {0, 0, 0, 0};
        if (q < nq) qv = *reinterpret_cast<const ulonglong4 *>(desc + ((size_t)qt.x * kcap + q) * 32);
#pragma unroll 4
        for (int j = 0; j < tn; j++) {
            const ulonglong4 t = tr[j];
            const uint32_t d = __popcll(qv.x ^ t.x) + __popcll(qv.y ^ t.y) + __popcll(qv.z ^ t.z) + __popcll(qv.w ^ t.w);
            const uint32_t key = (d << 16) | (uint32_t)(t0 + j);
            const uint32_t lo = key < k0 ? key : k0;
            const uint32_t hi = key < k0 ? k0 : key;
            k1 = hi < k1 ? hi : k1;
            k0 = lo;
        }
    }
    if (q < nq) part[((size_t)pair * nchunks + chunk) * kcap + q] = uint2{k0, k1};
}

// Merge chunk partials, emit the knnMatch table and BruteForceMatch's accept
// flag: m0.distance < ratio * m1.distance && !(m0.distance > dist_thresh)
// (MultiCameraFrame.cpp:1061-1063), float arithmetic as in the reference.
__global__ __launch_bounds__(256) void k_knn2_finalize(const uint2 *__restrict__ part, const int *__restrict__ counts,
                                                       const int2 *__restrict__ pairs, int kcap, int nchunks,
                                                       float dist_thresh, float ratio, KnnRow *__restrict__ out)
{
    const int pair = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int2 qt = pairs[pair];
    const int nq = counts[qt.x], nt = counts[qt.y];
    if (q >= nq) return;
    uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu;
    const int used = (nt + kKnnChunk - 1) / kKnnChunk;
    for (int c = 0; c < used; c++) {
        const uint2 p = part[((size_t)pair * nchunks + c) * kcap + q];
        // insert p.x then p.y
        uint32_t key = p.x;
        uint32_t lo = key < k0 ? key : k0, hi = key < k0 ? k0 : key;
        k1 = hi < k1 ? hi : k1; k0 = lo;
        key = p.y;
        lo = key < k0 ? key : k0; hi = key < k0 ? k0 : key;
        k1 = hi < k1 ? hi : k1; k0 = lo;
    }
    KnnRow r;
    r.idx0 = k0 == 0xffffffffu ? -1 : (int)(k0 & 0xffffu);
    r.idx1 = k1 == 0xffffffffu ? -1 : (int)(k1 & 0xffffu);
    r.d0 = k0 == 0xffffffffu ? -1 : (int)(k0 >> 16);
    r.d1 = k1 == 0xffffffffu ? -1 : (int)(k1 >> 16);
    int acc = 0;
    if (r.idx0 >= 0 && r.idx1 >= 0) {
        const float f0 = (float)r.d0, f1 = (float)r.d1;
        acc = (f0 < __fmul_rn(ratio, f1)) && !(f0 > dist_thresh);
    }
    r.d1 |= acc << 30;
    out[(size_t)pair * kcap + q] = r;
}

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------
hipError_t upload_umax(const int umax[16]) { return hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax, 16 * sizeof(int)); }

void launch_stage_f32(hipStream_t st, const float *src, int w, int h, int pitch_f, int channels, size_t img_stride_f,
                      uint8_t *pyr, const Geom &g, int nimg)
{
    dim3 grid((w + 255) / 256, h, nimg);
    hipLaunchKernelGGL(k_stage_f32, grid, dim3(256), 0, st, src, w, h, pitch_f, channels, img_stride_f, pyr, g);
}

void launch_pyramid(hipStream_t st, uint8_t *pyr, const Geom &g, const ResizeTap *tabs, int nimg)
{
    for (int l = 1; l < g.nlevels; l++) {
        dim3 grid((g.lv[l].w + 255) / 256, (g.lv[l].h + 3) / 4, nimg);
        hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, st, pyr, g, l, tabs);
    }
}

void launch_fast(hipStream_t st, const uint8_t *pyr, const Geom &g, int iniTh, int minTh, uint32_t *cell_kp,
                 int *cell_cnt, uint32_t *cand, int *lvl_off, int *overflow, int nimg, hipEvent_t ev_mid)
{
    dim3 grid(g.cells, nimg);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, st, pyr, g, iniTh, minTh, cell_kp, cell_cnt);
    if (ev_mid) (void)hipEventRecord(ev_mid, st);
    hipLaunchKernelGGL(k_compact, grid, dim3(256), 0, st, cell_kp, cell_cnt, g, cand, lvl_off, overflow);
}

void launch_blur(hipStream_t st, const uint8_t *pyr, uint8_t *blur, const Geom &g, int nimg)
{
    dim3 grid(g.tiles, nimg);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, st, pyr, blur, g);
}

void launch_describe(hipStream_t st, const uint8_t *pyr, const uint8_t *blur, const Geom &g, const uint32_t *sel,
                     const int *nsel, int orientation, uint8_t *desc, float *angles, int nimg)
{
    dim3 grid((g.kcap + 3) / 4, nimg);
    hipLaunchKernelGGL(k_describe, grid, dim3(256), 0, st, pyr, blur, g, sel, nsel, orientation, desc, angles);
}

void launch_knn2(hipStream_t st, const uint8_t *desc, const int *counts, const int2 *pairs, int npairs, int kcap,
                 uint2 *part, float dist_thresh, float ratio, KnnRow *out, hipEvent_t ev_mid)
{
    const int nchunks = (kcap + kKnnChunk - 1) / kKnnChunk;
    dim3 grid((kcap + 63) / 64, nchunks, npairs);
    hipLaunchKernelGGL(k_knn2, grid, dim3(64), 0, st, desc, counts, pairs, kcap, nchunks, part);
    if (ev_mid) (void)hipEventRecord(ev_mid, st);
    dim3 g2((kcap + 255) / 256, npairs);
    hipLaunchKernelGGL(k_knn2_finalize, g2, dim3(256), 0, st, part, counts, pairs, kcap, nchunks, dist_thresh, ratio, out);
}

}  // namespace mcorb
