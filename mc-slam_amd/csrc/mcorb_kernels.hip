// mcorb_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ORB front-end.
//
// One image = one camera frame; every kernel is launched over a whole batch of
// images (grid.y or grid.z = image index) so that launch cost is shared by all
// cameras and frames of a batch.  All arithmetic is integer and bit-exact with
// the reference's OpenCV CPU path as restated in oracle/ (see DESIGN.md).
// The levers are LDS tiles, packed 16-bit min/max, wave64 ballots and popcounts; one kernel is
// GEMM-shaped: the all-pairs Hamming k-NN has an exact dense formulation and runs on the matrix
// cores (k_knn2: v_mfma_i32_32x32x32_i8 on +-64 expanded descriptors).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcorb_common.h"
#include "mcorb_kernels.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace mcorb {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// number of set bits of a wave mask below this lane, plus acc (v_mbcnt_lo/hi: two instructions)
__device__ __forceinline__ int lane_rank(unsigned long long mask, int acc = 0)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, (uint32_t)acc));
}
// XCD-aware work index: workgroups b and b+8 share an XCD (and its L2) under the observed round-robin
// placement, so give each XCD one contiguous eighth of the work items -- spatial neighbours (cells or
// tiles that re-read the same 64-B lines for their halos) then hit in the same L2.  Bijective for any n;
// placement only changes speed, never results.
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
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
// Workgroup = 256 x 4 output pixels.  The source window it needs (a few rows,
// ~1.2 x 256 columns) is staged in LDS with aligned dword loads; each thread
// then produces 4 consecutive output pixels of one row -> one dword store.
// ---------------------------------------------------------------------------
constexpr int kResizeRows = kResizeTileH;   // output rows per workgroup (4 per thread): amortises the window fill + barrier

template <int NF>
__global__ __launch_bounds__(256) void k_resize(uint8_t *__restrict__ pyr, Geom g, int level,
                                                const ResizeTap *__restrict__ tabs, int srcPitch, int srcRows)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];   // srcRows x srcPitch bytes + 16 spare
    const LevelGeom &D = g.lv[level];
    const LevelGeom &S = g.lv[level - 1];
    const int img = blockIdx.z;
    uint8_t *base = pyr + (size_t)img * g.imgBytes;
    const int tid = threadIdx.x;
    const int bx0 = blockIdx.x * 256, by0 = blockIdx.y * kResizeRows;
    const int bx1 = min(bx0 + 255, D.w - 1), by1 = min(by0 + kResizeRows - 1, D.h - 1);
    // source window of this workgroup (tables are monotone); 16-byte aligned so it is filled with dwordx4 loads
    const int sx0 = tabs[D.xtab + bx0].s0 & ~15, sx1 = tabs[D.xtab + bx1].s1;
    const int sy0 = tabs[D.ytab + by0].s0, sy1 = tabs[D.ytab + by1].s1;
    const int nq = ((sx1 - sx0) >> 4) + 1, nr = sy1 - sy0 + 1;   // nq*16 <= srcPitch, nr <= srcRows by construction
    // this wave's row taps, one per lane, fetched now so that the latency hides behind the window fill (read back with
    // v_readlane below: a load inside the row loop is a vector load the whole wave waits for, eight times in a row)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wy0 = by0 + wave * (kResizeRows / 4);
    uint2 ytap;
    {
        const int l = tid & 63;
        const int dy = min(wy0 + (l < kResizeRows / 4 ? l : 0), D.h - 1);
        ytap = *reinterpret_cast<const uint2 *>(tabs + D.ytab + dy);
    }
    // the four column taps of this thread too (lanes past the row's end read the last group; they leave after the barrier)
    const int dx4 = bx0 + (tid & 63) * 4;
    const int dxc = min(dx4, ((D.w + 3) & ~3) - 4);
    const uint4 t01 = *reinterpret_cast<const uint4 *>(tabs + D.xtab + dxc);       // 4 taps, 8 bytes each
    const uint4 t23 = *reinterpret_cast<const uint4 *>(tabs + D.xtab + dxc + 2);
    {
        const uint8_t *sp = base + S.off + (size_t)sy0 * S.pitch + sx0;
        uint4 *w128 = reinterpret_cast<uint4 *>(win);
        const int pq = srcPitch >> 4;
        const float rcp_pq = 1.0f / (float)pq;
        // rows are 64-byte aligned and padded to a multiple of 64 bytes, so a 16-byte load never leaves the row
        if constexpr (NF > 0) {
            // every load of the fill is issued before the first LDS store (NF >= ceil(window chunks / 256), launch_pyramid).
            // As a loop of load - wait - store the fill was four memory round trips in a row per workgroup, on top of the
            // one for the tap lookups: most of a workgroup's life was spent waiting for them.
            uint4 v[NF];
            bool ok[NF];
#pragma unroll
            for (int k = 0; k < NF; k++) {
                const int i = tid + 256 * k, ic = min(i, nr * pq - 1);
                const int r = (int)(((float)ic + 0.5f) * rcp_pq), q = ic - r * pq;
                ok[k] = i < nr * pq && q < nq;
                // unconditional, from a clamped (always valid) address: a load under a branch is waited for at the branch's end
                v[k] = *reinterpret_cast<const uint4 *>(sp + (size_t)r * S.pitch + 16 * min(q, nq - 1));
            }
            // (stores without a branch either: the optimiser sinks a load into the branch that holds its only use.  Lanes with
            // nothing to store write the spare 16 bytes behind the window.)
            const int spare = (srcPitch >> 4) * srcRows;
#pragma unroll
            for (int k = 0; k < NF; k++) w128[ok[k] ? tid + 256 * k : spare] = v[k];
        } else {
            for (int i = tid; i < nr * pq; i += 256) {
                const int r = (int)(((float)i + 0.5f) * rcp_pq), q = i - r * pq;
                if (q < nq) w128[i] = *reinterpret_cast<const uint4 *>(sp + (size_t)r * S.pitch + 16 * q);
            }
        }
    }
    __syncthreads();
    // The row taps must be in their registers in EVERY lane before lanes leave below: rows are read back with v_readlane
    // from lanes 0..7, which a narrow last column block retires.  (Without this the optimiser sinks the load to its first
    // use, behind the early return, and lanes 6 and 7 never load theirs.)
    asm volatile("" ::"v"(ytap.x), "v"(ytap.y), "v"(t01.x), "v"(t01.y), "v"(t01.z), "v"(t01.w), "v"(t23.x), "v"(t23.y), "v"(t23.z), "v"(t23.w));
    if (dx4 >= D.w) return;
    const uint32_t tw[8] = {t01.x, t01.y, t01.z, t01.w, t23.x, t23.y, t23.z, t23.w};
    // The table is padded to a multiple of four with copies of the row's last tap (build_resize_axis), so the taps of
    // columns past the image's edge are in range like any other; their pixels are masked out of the store.
    // (Reading the second sample at s0 + 1 through the same address register lets the compiler merge the two byte reads
    // into one ds_read_u16 at an odd address; unaligned LDS reads are slow on gfx950: 663 vs 239 us for the seven launches.)
    int s0[4], s1[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        s0[k] = (int)(tw[2 * k] & 0xffff) - sx0;
        s1[k] = (int)(tw[2 * k] >> 16) - sx0;
        a0[k] = (int)(short)(tw[2 * k + 1] & 0xffff);
        a1[k] = (int)(short)(tw[2 * k + 1] >> 16);
    }
    const uint32_t mask = dx4 + 4 > D.w ? 0xffffffffu >> (8 * (dx4 + 4 - D.w)) : 0xffffffffu;   // keep the row padding zero
    // Each wave produces kResizeRows / 4 CONSECUTIVE output rows.  Consecutive rows share source rows (s1 of row j is s0 of
    // row j+1 five times out of six at scale 1.2), so the horizontal two-tap sums R = S[s0]*a0 + S[s1]*a1 are kept per source
    // row in two register sets and computed once (the row indices are wave-uniform: the reuse tests are scalar branches).
    // All products fit 24 bits (coefficients <= 2048, samples <= 255, sums >> 4 <= 32640): v_mul/mad_u32_u24 are exact and
    // full rate (v_mul_lo_u32 is a quarter of that).
    int rowA = -1, rowB = -1;
    uint32_t HA[4], HB[4];
    // The second sample of a tap is the byte after the first -- except at the image's right edge, where the table repeats the
    // first and gives it the coefficient 0 (cv's HResizeLinear tail: S[sx] * ONE), so the byte after is as good there.  Both
    // are read through ONE address register, offset 0 and 1: four address adds per source row instead of eight (a fifth of
    // the row's vector instructions).  Inline asm, because as C++ the compiler merges the two byte reads into one
    // ds_read_u16 at an odd address, and unaligned LDS reads are slow on gfx950 (663 vs 239 us for the seven launches).
    const uint32_t ldsWin = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)win;
    uint32_t sb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) sb[k] = ldsWin + (uint32_t)s0[k];
    auto hrow = [&](int sr, uint32_t *Hout) {
        const uint32_t ro = (uint32_t)__mul24(sr - sy0, srcPitch);
        const uint32_t ad0 = sb[0] + ro, ad1 = sb[1] + ro, ad2 = sb[2] + ro, ad3 = sb[3] + ro;
        uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
        asm volatile("ds_read_u8 %0, %8\n\tds_read_u8 %1, %8 offset:1\n\t"
                     "ds_read_u8 %2, %9\n\tds_read_u8 %3, %9 offset:1\n\t"
                     "ds_read_u8 %4, %10\n\tds_read_u8 %5, %10 offset:1\n\t"
                     "ds_read_u8 %6, %11\n\tds_read_u8 %7, %11 offset:1\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                     : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3)
                     : "memory");
        Hout[0] = (__umul24(x0, (uint32_t)a0[0]) + __umul24(x1, (uint32_t)a1[0])) >> 4;
        Hout[1] = (__umul24(x2, (uint32_t)a0[1]) + __umul24(x3, (uint32_t)a1[1])) >> 4;
        Hout[2] = (__umul24(x4, (uint32_t)a0[2]) + __umul24(x5, (uint32_t)a1[2])) >> 4;
        Hout[3] = (__umul24(x6, (uint32_t)a0[3]) + __umul24(x7, (uint32_t)a1[3])) >> 4;
    };
#pragma unroll
    for (int rr = 0; rr < kResizeRows / 4; rr++) {
        const int dy = wy0 + rr;
        if (dy >= D.h) break;
        const uint32_t ys = (uint32_t)__builtin_amdgcn_readlane((int)ytap.x, rr), yc = (uint32_t)__builtin_amdgcn_readlane((int)ytap.y, rr);
        const int r0 = (int)(ys & 0xffff), r1 = (int)(ys >> 16);
        // invariant after this block: HA holds source row r0, HB holds source row r1 (register moves, no selects)
        if (r0 != rowA) {
            if (r0 == rowB) {
#pragma unroll
                for (int k = 0; k < 4; k++) HA[k] = HB[k];
            } else {
                hrow(r0, HA);
            }
            rowA = r0;
        }
        if (r1 != rowB) {
            if (r1 == rowA) {
#pragma unroll
                for (int k = 0; k < 4; k++) HB[k] = HA[k];
            } else {
                hrow(r1, HB);
            }
            rowB = r1;
        }
        const uint32_t b0 = yc & 0xffff, b1 = yc >> 16;   // row coefficients: 0 .. 2048
        // cv: (((b0 * H0) >> 16) + ((b1 * H1) >> 16) + 2) >> 2.  The +2 rides in the first product's upper half
        // (no carry can reach it from below), the two >> 16 are the upper halves of the products.
        // T = upper half of p0 + upper half of p1 (<= 1023): the two >> 16 are the WORD_1 selectors of one SDWA add, which also
        // drops every second result into the upper half of the previous pixel's register; >> 2 is then one packed shift per
        // two pixels and the four bytes meet in one v_perm (15 instructions per four pixels; the compiler's own sequence of
        // shifts, masks and ors took 30)
        uint32_t p0[4], p1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            p0[k] = __umul24(b0, HA[k]) + 0x20000u;
            p1[k] = __umul24(b1, HB[k]);
        }
        uint32_t t01v, t23v;
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t01v) : "v"(p0[0]), "v"(p1[0]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t01v) : "v"(p0[1]), "v"(p1[1]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t23v) : "v"(p0[2]), "v"(p1[2]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t23v) : "v"(p0[3]), "v"(p1[3]));
        const u16x2 s01 = __builtin_bit_cast(u16x2, t01v) >> (unsigned short)2, s23 = __builtin_bit_cast(u16x2, t23v) >> (unsigned short)2;
        const uint32_t out = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, s23), __builtin_bit_cast(uint32_t, s01), 0x06040200u);
        *reinterpret_cast<uint32_t *>(base + D.off + (size_t)dy * D.pitch + dx4) = out & mask;
    }
}

// ---------------------------------------------------------------------------
// FAST-9/16 corner score + cell-local 3x3 non-max suppression per 35-px cell
// (ComputeKeyPointsOctTree detection loop, ORBextractor.cpp:804-871, calling
// cv::FAST on each cell ROI).
//
// For a pixel v with ring r[0..15]:
//   A = max( v - min over the 16 contiguous 9-arcs of max(r in arc),      (dark)
//            max over the 16 contiguous 9-arcs of min(r in arc) - v )     (bright)
// p is a FAST corner at threshold t  <=>  A > t, and cornerScore = A - 1
// independently of t.  Everything runs at t = iniThFAST first and, only for the
// cells that came out empty, once more at minThFAST (the reference's second
// cv::FAST call on the same ROI).
//
// One WAVE per workgroup, NC cells per wave (their survivor lists are pooled so that the 64-lane
// trips of passes 2 and 3 are filled); every phase is wave-synchronous: no s_barrier anywhere.
//   stage   ROI rows -> LDS as one u16 per pixel (row = lane, 16-byte global loads, all issued before
//           the first LDS store), so that every later LDS read IS a packed-i16 operand (no unpacking)
//   pass 1  necessary condition on every interior pixel, 4 px per lane (five ds_read_b64): a 9-arc of
//           the 16-ring holds two NEIGHBOURING compass points; a compass point of the pair (up, down)
//           is a ring neighbour of either point of (left, right), hence
//             bright:  min(max(U,D), max(L,R)) > v + t      dark:  max(min(U,D), min(L,R)) < v - t
//           survivors are appended to the work list in raster order (ballots + mbcnt ranks)
//   pass 2  full arc score of the survivors, 64 at a time, on ring values packed as (r[k], r[k+8]):
//           suffix / prefix minima and maxima of the two 8-blocks give all 16 nine-windows
//           (window k = suffix_k of one block + prefix_k of the other), 60 packed min/max in all;
//           the score map receives A where A > t and 0 elsewhere
//   pass 3  3x3 NMS over the work list (already in raster order = cv::FAST's output order),
//           keypoints emitted straight to the cell's slot.
// ---------------------------------------------------------------------------

struct FastCell {      // wave-uniform description of one cell
    int on;            // evaluated (not skipped by the reference's border rule, not degenerate)
    int cell;          // cell index inside the image
    int ph;            // byte phase of the ROI's first column in HBM (iniX & 3) = its column in the LDS tile
    int wi, hi;        // evaluated columns / rows (ROI minus FAST's 3-px margin)
    int rows;          // ROI rows
    int g0, ng;        // first 4-px group holding an evaluated column, number of such groups
    int R;             // rows one pass-1 trip covers: 64 / ng
    float rcp_ng;      // 1 / ng
    uint32_t headMask, tailMask;   // pass-bit masks (FastLane::vmask's positions) of the evaluated columns of a row's first / last group
    uint32_t org;      // pack_cand(cj*wCell - ph, ci*hCell, 0) mod 2^32: tile (col, row) -> candidate record
    const uint8_t *src;   // first staged byte (ROI row 0, column iniX - ph)
    int pitch;         // HBM row pitch of the level
};

// The cell's record comes from a host-built table (fast_cell_table(), 8 dwords per cell of an image): one scalar load
// instead of a seven-step search for the cell's level, a division and ~40 scalar instructions of ROI arithmetic at the
// head of each of the 254 k waves of a launch.
constexpr int kFastCellRecDw = 8;
__device__ __forceinline__ void fast_cell_setup(const uint8_t *pyr, const Geom &g, int img, int cell, const uint32_t *__restrict__ rec, FastCell &C)
{
    const uint4 r = *reinterpret_cast<const uint4 *>(rec + (size_t)cell * kFastCellRecDw);
    const uint2 r4 = *reinterpret_cast<const uint2 *>(rec + (size_t)cell * kFastCellRecDw + 4);
    C.cell = cell;
    C.src = pyr + (size_t)img * g.imgBytes + r.x;
    C.org = r.y;
    C.pitch = (int)r.z;
    C.rows = (int)(r.w & 0xff);
    C.wi = (int)((r.w >> 8) & 0xff);
    C.hi = (int)((r.w >> 16) & 0xff);
    C.on = (int)((r.w >> 24) & 1);
    C.ph = (int)((r.w >> 25) & 3);
    C.g0 = (int)(r4.x & 0xff);
    C.ng = (int)((r4.x >> 8) & 0xff);
    C.R = (int)(r4.x >> 16);
    C.rcp_ng = __uint_as_float(r4.y);
    const uint2 r6 = *reinterpret_cast<const uint2 *>(rec + (size_t)cell * kFastCellRecDw + 6);
    C.headMask = r6.x;
    C.tailMask = r6.y;
}

// host: the records, exactly the reference's ROI arithmetic (ORBextractor.cpp:804-833; float there, exact in int)
int fast_cell_table(const Geom &g, std::vector<uint32_t> &tab)
{
    const int off = (int)tab.size();
    tab.resize(tab.size() + (size_t)g.cells * kFastCellRecDw, 0u);
    for (int level = 0; level < g.nlevels; level++) {
        const LevelGeom &L = g.lv[level];
        for (int cl = 0; cl < L.nCols * L.nRows; cl++) {
            const int ci = cl / L.nCols, cj = cl - ci * L.nCols;
            const int iniY = kMinBorder + ci * L.hCell;
            int maxY = iniY + L.hCell + 6;
            const int iniX = kMinBorder + cj * L.wCell;
            int maxX = iniX + L.wCell + 6;
            const bool skip = (iniY >= L.maxBorderY - 3) || (iniX >= L.maxBorderX - 6);
            if (maxY > L.maxBorderY) maxY = L.maxBorderY;
            if (maxX > L.maxBorderX) maxX = L.maxBorderX;
            const int cols = maxX - iniX, rows = maxY - iniY;
            const int wi = cols - 6, hi = rows - 6;   // FAST evaluates ROI columns 3..cols-4, rows 3..rows-4
            const int ph = iniX & 3, g0 = (ph + 3) >> 2;
            const bool on = !(skip || wi <= 0 || hi <= 0);
            const int ng = on ? ((ph + 2 + wi) >> 2) - g0 + 1 : 1;
            uint32_t *r = tab.data() + off + (size_t)(L.cell0 + cl) * kFastCellRecDw;
            r[0] = on ? (uint32_t)(L.off + (size_t)iniY * L.pitch + (iniX - ph)) : 0u;
            r[1] = ((uint32_t)(ci * L.hCell) << 20) + ((uint32_t)(cj * L.wCell - ph) << 8);
            r[2] = (uint32_t)L.pitch;
            r[3] = on ? ((uint32_t)rows | ((uint32_t)wi << 8) | ((uint32_t)hi << 16) | (1u << 24) | ((uint32_t)ph << 25)) : 0u;
            r[4] = (uint32_t)g0 | ((uint32_t)ng << 8) | ((uint32_t)(64 / ng) << 16);
            const float rcp = 1.0f / (float)ng;
            uint32_t bits;
            memcpy(&bits, &rcp, 4);
            r[5] = bits;
            // evaluated tile columns: [ph + 3, ph + 3 + wi); only a row's first and last group can hold columns outside
            auto pxmask = [&](int G) {
                static const uint32_t bit[4] = {1u << 15, 1u, 1u << 31, 1u << 16};   // px0, px1, px2, px3
                uint32_t m = 0;
                for (int j = 0; j < 4; j++) {
                    const int x = 4 * G + j - (ph + 3);
                    if (x >= 0 && x < wi) m |= bit[j];
                }
                return m;
            };
            r[6] = on ? pxmask(g0) : 0u;
            r[7] = on ? pxmask(g0 + ng - 1) : 0u;
        }
    }
    return off;
}

// Work-list capacity (pixels).  Pass 1 appends one entry per surviving 4-px GROUP (at most 64 per trip) and runs only while
// 64 more still fit cap / 4 groups, so the pixel list the groups expand into never overflows; a cell with more survivors
// than that (dense texture, noise) is processed in several chunks (see the kernel).  The capacity also sets the wave's
// LDS footprint and with it the occupancy; measured at 720p (128 images) with the round-2 kernel: 512 entries = 32 waves
// per CU: 410 us; 640: 387; 768 = 29 waves: 380; 832: 377; 896: 403; 1024 = 26 waves: 402.  The benchmark's cells hold 80
// survivors on average, 250 at most.
constexpr int kFastListCap = 768;

#define EMAX __builtin_elementwise_max
#define EMIN __builtin_elementwise_min
// Per-lane constants of pass 1 for one cell.  A trip covers R = 64 / ng consecutive rows of the cell: lane = r * ng + gi
// tests the 4-px group gi of row (trip's first row + r), which is the raster order of the groups.  Nothing about a lane's
// place changes from trip to trip except the row base, so the tile offset, the column-validity mask and the entry code
// are computed once per cell (round 2 looked them up per trip in a host-built item table: a global load, an LDS read
// and six vector instructions per trip).
struct FastLane {
    int off0;          // tile byte offset of the lane's group in the cell's first row (inactive lanes: a valid one)
    uint32_t vmask;    // pass bits of the group's evaluated columns: px0 -> bit 15, px1 -> bit 0, px2 -> bit 31, px3 -> bit 16
};
template <int TP>
__device__ __forceinline__ FastLane fast_lane_setup(const FastCell &C, int lane)
{
    FastLane L;
    const int r = (int)(((float)lane + 0.5f) * C.rcp_ng);   // lane / ng (the product is at least 1 / (2 ng) away from an integer)
    const int gi = lane - r * C.ng;
    const bool act = r < C.R;
    L.off0 = act ? (3 + r) * TP + 4 * (C.g0 + gi) : 3 * TP + 4 * C.g0;
    uint32_t m = 0x80018001u;
    m = gi == 0 ? m & C.headMask : m;
    m = gi == C.ng - 1 ? m & C.tailMask : m;
    L.vmask = act ? m : 0u;
    return L;
}

// pass 1 over one cell, from row `row0` (a multiple of R) until the cell is done or the group list is full; returns the
// number of group entries and advances row0.  Entry = pass bits (FastLane::vmask's positions) | tile byte offset >> 1.
//
// The necessary condition is evaluated WITHOUT unpacking bytes to 16 bits for half of the pixels: the packed u16
// min / max of two raw dwords orders the lanes by their HIGH byte first, so its high byte is the min / max of the high
// bytes whatever the low bytes hold.  The odd pixels of a group (bytes 1, 3) therefore run on the raw dwords, on values
// scaled by 256 with garbage below, masked once before the two subtractions (v_and: a 2-cycle instruction; the ten
// v_perm of the round-2 version were 4-cycle ones); the even pixels (bytes 0, 2) on the dwords masked with 0x00ff00ff.
//   odd:   w = max(sat(hi - V), sat(V - lo)) in units of 256;  pass <=> w >= 256 (T + 1) <=> sat(w - (256 T + 255)) is odd
//   even:  w = max(hi - V, V - lo);                            pass <=> T - w < 0 (sign bit)
template <int TP>
__device__ __forceinline__ int fast_pass1(const uint8_t *tile, uint32_t *glist, const FastCell &C, int T, int lane, int &row0, int gcap,
                                          const FastLane &L)
{
    const u16x2 Tev = u16x2{(unsigned short)T, (unsigned short)T};
    const u16x2 Tod = u16x2{(unsigned short)(256 * T + 255), (unsigned short)(256 * T + 255)};
    int nG = 0;
    auto trip = [&](int rb, bool tail) {
        int off = L.off0 + rb * TP;
        bool in = true;
        if (tail) {   // last trip: rows past the cell's end read a valid row and are masked out of the ballot
            in = (int)(((float)lane + 0.5f) * C.rcp_ng) + rb < C.hi;
            off = in ? off : 3 * TP + 4 * C.g0;
        }
        const uint32_t *p = reinterpret_cast<const uint32_t *>(tile + off);
        const uint32_t Cc = p[0], Cl = p[-1], Cr = p[1], Up = p[-3 * (TP / 4)], Dn = p[3 * (TP / 4)];
        const uint32_t Lf = __builtin_amdgcn_alignbyte(Cc, Cl, 1);   // px 4G-3 .. 4G
        const uint32_t Rt = __builtin_amdgcn_alignbyte(Cr, Cc, 3);   // px 4G+3 .. 4G+6
#define U16(X) __builtin_bit_cast(u16x2, (uint32_t)(X))
#define S16(X) __builtin_bit_cast(s16x2, (uint32_t)(X))
        // odd pixels: high bytes, raw
        const u16x2 hO = EMIN(EMAX(U16(Up), U16(Dn)), EMAX(U16(Lf), U16(Rt)));
        const u16x2 lO = EMAX(EMIN(U16(Up), U16(Dn)), EMIN(U16(Lf), U16(Rt)));
        const u16x2 hOm = U16(__builtin_bit_cast(uint32_t, hO) & 0xff00ff00u), lOm = U16(__builtin_bit_cast(uint32_t, lO) & 0xff00ff00u);
        const u16x2 vO = U16(Cc & 0xff00ff00u);
        const u16x2 wO = EMAX(__builtin_elementwise_sub_sat(hOm, vO), __builtin_elementwise_sub_sat(vO, lOm));
        const uint32_t zO = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(wO, Tod));   // bit 0 / 16 set <=> px1 / px3 pass
        // even pixels: low bytes
        const s16x2 vE = S16(Cc & 0x00ff00ffu), uE = S16(Up & 0x00ff00ffu), dE = S16(Dn & 0x00ff00ffu);
        const s16x2 lE = S16(Lf & 0x00ff00ffu), rE = S16(Rt & 0x00ff00ffu);
        const s16x2 wE = EMAX(EMIN(EMAX(uE, dE), EMAX(lE, rE)) - vE, vE - EMAX(EMIN(uE, dE), EMIN(lE, rE)));
        const uint32_t sE = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, Tev) - wE);           // bit 15 / 31 set <=> px0 / px2 pass
#undef U16
#undef S16
        const uint32_t M = ((sE & 0x80008000u) | (zO & 0x00010001u)) & L.vmask;
        const bool c = M != 0 && in;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(c);
        if (c) glist[lane_rank(b, nG)] = M | ((uint32_t)off >> 1);
        nG += __popcll(b);
    };
    int rb = row0;
    for (; rb + C.R <= C.hi && nG + 64 <= gcap; rb += C.R) trip(rb, false);   // full trips: every active lane holds a row of the cell
    if (rb < C.hi && rb + C.R > C.hi && nG + 64 <= gcap) {
        trip(rb, true);
        rb += C.R;
    }
    row0 = rb;
    return nG;
}

// group entries -> pixel work list (tile byte offsets, raster order: groups are in raster order, pixels ascend inside a
// group).  64 groups per trip; a group holds 1 .. 4 survivors: lane's first slot = groups before it + their extra
// survivors (two ballots over the bits of n - 1).  The group list sits in the tail of the pixel list's buffer (see the
// kernel): all of a trip's entries are in registers before its first pixel is written, and 4 (g + 64) <= cap pixels
// written after g + 64 groups never reach entry g + 64.
__device__ __forceinline__ int fast_expand(const uint32_t *glist, uint16_t *work, int nG, int lane)
{
    int nA = 0;
    for (int i0 = 0; i0 < nG; i0 += 64) {
        const bool in = i0 + lane < nG;
        const uint32_t E = in ? glist[i0 + lane] : 0u;
        const uint32_t m = E & 0x80018001u;
        const int n1 = in ? __popc(m) - 1 : 0;   // survivors of the group beyond the first
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64((n1 & 1) != 0), b1 = __builtin_amdgcn_ballot_w64(n1 >= 2);
        int o = nA + lane + lane_rank(b0) + 2 * lane_rank(b1);
        const uint32_t off = (E & 0x3ffeu) << 1;
        // the list entries are read before the pixel slots that overlay them are written: same wave, LDS executes a wave's
        // accesses in program order -- the compiler must keep that order too (the two pointers differ in type)
        asm volatile("" ::: "memory");
        if (m & 0x8000u) work[o++] = (uint16_t)off;
        if (m & 0x1u) work[o++] = (uint16_t)(off + 1);
        if (m & 0x80000000u) work[o++] = (uint16_t)(off + 2);
        if (m & 0x10000u) work[o++] = (uint16_t)(off + 3);
        const int cnt = min(nG - i0, 64);
        nA += cnt + __popcll(b0) + 2 * __popcll(b1);
        asm volatile("" ::: "memory");
    }
    return nA;
}

// arc score of the pixel at byte c of the u8 LDS tile (pitch TP): A as defined above, 0 <= A <= 255.
// (Fetching the ring with seven unaligned ds_read_b32/b64 per pixel instead of 17 byte reads is functionally fine on
// gfx950 but was measured 2x slower for the whole kernel: 900 vs 435 us per 128 images.)
template <int TP>
__device__ __forceinline__ int fast_arc_score(const uint8_t *tile, int pos)   // pos = tile byte offset of the pixel
{
    // (the base offset &pixel[-3 rows][-3 columns] is pinned in its register: left alone the compiler rebases on the centre
    // pixel and spends seven vector adds per survivor on the ring bytes whose offsets are then negative -- ds_read
    // offsets are unsigned)
    int lo = pos - (3 * TP + 3);
    asm volatile("" : "+v"(lo));
    const uint8_t *c = tile + lo + (3 * TP + 3);
    // ring offsets (dx,dy), k = 0..15 (cv::FAST makeOffsets, patternSize 16): (0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),
    // then the same negated for k+8.  X[k] = (r[k], r[k+8]) as two u16
    u16x2 X[8];
    X[0] = u16x2{c[3 * TP], c[-3 * TP]};          X[1] = u16x2{c[3 * TP + 1], c[-3 * TP - 1]};
    X[2] = u16x2{c[2 * TP + 2], c[-2 * TP - 2]};  X[3] = u16x2{c[TP + 3], c[-TP - 3]};
    X[4] = u16x2{c[3], c[-3]};                    X[5] = u16x2{c[-TP + 3], c[TP - 3]};
    X[6] = u16x2{c[-2 * TP + 2], c[2 * TP - 2]};  X[7] = u16x2{c[-3 * TP + 1], c[3 * TP - 1]};
    const uint32_t v = c[0];
    // suffix / prefix scans of both 8-blocks at once (block 0 in the low halves, block 1 in the high halves)
    u16x2 Sn[8], Sx[8], Pn[8], Px[8];
    Sn[7] = Sx[7] = X[7];
    Pn[0] = Px[0] = X[0];
#pragma unroll
    for (int i = 6; i >= 1; i--) { Sn[i] = EMIN(X[i], Sn[i + 1]); Sx[i] = EMAX(X[i], Sx[i + 1]); }
#pragma unroll
    for (int i = 1; i < 8; i++) { Pn[i] = EMIN(X[i], Pn[i - 1]); Px[i] = EMAX(X[i], Px[i - 1]); }
    Sn[0] = Pn[7]; Sx[0] = Px[7];   // whole-block minimum / maximum
    // nine-window starting at k (low half) and at k+8 (high half) = suffix k of its own block + prefix k of the other block
    u16x2 bn, bx;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u16x2 wn = EMIN(Sn[k], __builtin_shufflevector(Pn[k], Pn[k], 1, 0));   // min over the window: bright arcs
        const u16x2 wx = EMAX(Sx[k], __builtin_shufflevector(Px[k], Px[k], 1, 0));   // max over the window: dark arcs
        bn = k ? EMAX(bn, wn) : wn;
        bx = k ? EMIN(bx, wx) : wx;
    }
    const int Mx = max((int)bn.x, (int)bn.y), Mn = min((int)bx.x, (int)bx.y);
    const int a = max(Mx - (int)v, (int)v - Mn);
    return a < 0 ? 0 : a;
}
#undef EMAX
#undef EMIN

// TP: LDS row pitch in bytes of every cell of the launch (48 fits all 35-px cells of the usual resolutions), a
// compile-time constant so that the ring / neighbour offsets fold into the ds_read offset fields.
// One wave per cell.  (Several cells per wave, one after the other with the next ROI prefetched into registers, or two
// cells with pooled survivor lists, were measured slower: the extra LDS / SGPRs cost more occupancy than they save --
// 443-456 us and 787 us against 431 us per 128 images.  Round 3, a plain loop over 2 / 3 / 4 / 8 consecutive cells in the same
// wave, nothing prefetched, no extra registers: 394 / 367 / 373 / 419 against 348 us.)
// LDS of the wave: [16 B][tile, tileB][16 B][score map of ROI rows 2 .. rows-3, scB][16 B][work list, cap x 2 B; its second
// half doubles as the group list of pass 1] = 5.4 KiB at TP = 48; <= 80 SGPRs and <= 64 VGPRs leave 8 waves per SIMD.
template <int TP>
__global__ __launch_bounds__(64) void k_fast_cells(const uint8_t *__restrict__ pyr, Geom g, int iniTh, int minTh,
                                                   int tileB, int scB, int cap, const uint32_t *__restrict__ cellRec,
                                                   uint32_t *__restrict__ cell_kp, int *__restrict__ cell_cnt)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int NQ = TP / 16;
    uint8_t *tile = lds + 16;
    uint8_t *sc = tile + tileB + 16;          // score of tile byte `pos` lives at sc2[pos]: NMS touches rows 2 .. rows-3 only
    uint8_t *sc2 = sc - 2 * TP;
    uint16_t *work = reinterpret_cast<uint16_t *>(sc + scB + 16);
    uint32_t *glist = reinterpret_cast<uint32_t *>(work + cap / 2);   // cap / 4 group entries in the list's second half
    const int gcap = cap >> 2;
    struct __attribute__((packed, aligned(4))) Chunk { uint32_t a, b, c, d; };

    const int lane = threadIdx.x;
    const int img = blockIdx.y;
    FastCell C;
    fast_cell_setup(pyr, g, img, __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x)), cellRec, C);
    int *out_cnt = cell_cnt + (size_t)img * g.cells + C.cell;
    if (!C.on) {
        if (lane == 0) *out_cnt = 0;
        return;
    }

    // ---- stage the ROI: row = lane, NQ 16-byte chunks per row (4-byte aligned in HBM, 16-byte aligned in LDS), all
    //      loads issued before the first LDS store; clear the score map; the lane's pass-1 constants meanwhile ----
    FastLane FL;
    for (int r = lane; r < C.rows; r += 64) {   // one trip unless the ROI is taller than 64 rows (70-px cells of odd resolutions)
        Chunk ch[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) ch[q] = *reinterpret_cast<const Chunk *>(C.src + (size_t)r * C.pitch + 16 * q);
        uint4 *dst = reinterpret_cast<uint4 *>(tile + r * TP);
#pragma unroll
        for (int q = 0; q < NQ; q++) dst[q] = uint4{ch[q].a, ch[q].b, ch[q].c, ch[q].d};
    }
    {
        uint4 *s128 = reinterpret_cast<uint4 *>(sc);
        for (int i = lane; i < scB / 16; i += 64) s128[i] = uint4{0, 0, 0, 0};
        FL = fast_lane_setup<TP>(C, lane);
    }
    __syncthreads();   // single-wave workgroup: lowers to a wait, not an s_barrier

    uint32_t *dst = cell_kp + ((size_t)img * g.cells + C.cell) * g.cellCap;
    int total = 0;
    // pass 2: full arc score of the survivors; the score map gets A where A > T, 0 elsewhere
    auto score = [&](int nA, int T) {
        for (int i = lane; i < nA; i += 64) {
            const int pos = work[i];
            const int a = fast_arc_score<TP>(tile, pos);
            sc2[pos] = (uint8_t)(a > T ? a : 0);
        }
    };
    // pass 3: NMS over the work list; it is in raster order, so keypoints are emitted in cv::FAST's order.
    // keep <=> score strictly above the 8 neighbours' (0 where not a corner)
    auto nms = [&](int nA) {
        for (int i0 = 0; i0 < nA; i0 += 64) {
            const int i = i0 + lane;
            // lanes past the list's end look at tile byte 3 * TP: column 0 is never evaluated, its score stays 0, and a
            // zero score is never strictly above its neighbours'
            const int pos = i < nA ? (int)work[i] : 3 * TP;
            int lo = pos - (TP + 1);   // &score[-1 row][-1 column], pinned so that all nine offsets are non-negative immediates
            asm volatile("" : "+v"(lo));
            const uint8_t *s = sc2 + lo + (TP + 1);
            const uint32_t a = s[0];
            const uint32_t n = max(max(max((uint32_t)s[1], (uint32_t)s[-1]), max((uint32_t)s[-TP - 1], (uint32_t)s[-TP])),
                                   max(max((uint32_t)s[-TP + 1], (uint32_t)s[TP - 1]), max((uint32_t)s[TP], (uint32_t)s[TP + 1])));
            const bool keep = a > n;
            const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
            if (keep) {
                const int o = lane_rank(b, total);
                const int yy = (int)(((float)pos + 0.5f) * (1.0f / (float)TP)), xx = __mul24(yy, -TP) + pos;   // tile coordinates
                // keypoint coordinates relative to minBorder: FAST's ROI coordinate + cell origin (:864-865)
                if (o < g.cellCap) dst[o] = C.org + ((uint32_t)yy << 20) + ((uint32_t)xx << 8) + (a - 1);
            }
            total += __popcll(b);
        }
    };
    // The reference calls cv::FAST at iniTh and, only when the cell came out empty, again at minTh
    // (ORBextractor.cpp:847-856).  Same here: the wave-uniform retry is paid by empty cells only.
    int T = iniTh;
#pragma unroll 1
    for (int attempt = 0; attempt < 2; attempt++) {
        // sweep 0 scores the survivors, chunk by chunk when the list cannot hold them all; the usual cell fits in
        // one chunk and is finished right away.  Otherwise sweep 1 goes over the cell again for the NMS (the
        // score map is complete by then).
        int row0 = 0, sweep = 0;
#pragma unroll 1
        for (;;) {
            const bool whole = row0 == 0;
            const int nG = fast_pass1<TP>(tile, glist, C, T, lane, row0, gcap, FL);
            __syncthreads();
            const int nA = fast_expand(glist, work, nG, lane);
            __syncthreads();
            if (sweep == 0) {
                score(nA, T);
                __syncthreads();
            }
            if (sweep == 1 || (whole && row0 >= C.hi)) nms(nA);
            if (row0 >= C.hi) {
                if (sweep == 1 || whole) break;
                sweep = 1;
                row0 = 0;
            }
            __syncthreads();
        }
        if (total || T == minTh) break;
        T = minTh;
        __syncthreads();
    }
    if (lane == 0) *out_cnt = total;
}

// ---------------------------------------------------------------------------
// Candidate compaction + quad-tree bucketing, one workgroup per (level, image).
//
// The per-cell lists of k_fast_cells are in (cell row, cell col, raster) order = the reference's
// vToDistributeKeys order.  The kernel counting-sorts each level's candidates by quad-tree path code
// (path_code(), depth g.lv[level].depth) in LDS and hands the host the bucket start offsets: the
// host-side DistributeOctTree logic then gets every node's key count in O(1) and never has to
// partition the candidate list itself for the first `depth` splits.  Order inside a bucket is
// arbitrary; the selection only depends on the key SETS.  The kernel also finds each bucket's winner
// of the reference's final pick (largest response, first one in vToDistributeKeys order on ties,
// ORBextractor.cpp:757-775): key = response << 23 | (2^23-1 - order), order = cell * cellCap + index
// in the cell (monotone in the vToDistributeKeys position), LDS atomicMax per bucket, and ships
// (key, the winning candidate) per bucket, so the host never scans a node's keys unless the tree
// went deeper than the bucketing.  The bucket tables go to a per-image table block in device memory
// (one DMA per batch takes them to the host); the level's list itself stays on the device unless
// the selection can need it.
//
// Waves walk the cells (wave w takes cells 8w .. 8w+7, then 8(w+W) ..., lane = index inside the cell) with the eight
// cells' loads in flight together: no search for the owning cell, short dependency chains.  The path code of a
// candidate is two LDS table reads, lutx[x] | luty[y] (path_code_tables(), mcorb_common.h).
// (Round 1's version binary-searched the cell of every list position and ran path_code() -- a float division and
// `depth` split iterations -- twice per candidate: 64 M vector instructions in 8192 long waves, 445 us per 128 images.
// It was believed to be PCIe-bound; the same kernel writing to device memory took 408 us.)
// ---------------------------------------------------------------------------
// Private copies of every bucket's counter and winner key in LDS; lane & (copies - 1) picks one.  A cell's candidates fall into
// one to four buckets, so most lanes of an LDS atomic hit the same address and are served one after the other: with four
// copies a quarter as many (46 -> 39.6 us per 128 images; two: 40.3, eight: 39.3).  One copy when four would not fit 64 KiB
// of LDS (levels with thousands of buckets: very large feature budgets on wide images).
constexpr int kCompactCopies = 4;
template <int kCompactWaves>
__device__ __forceinline__ int block_exclusive_scan(int v, int *wsum, int *total)   // 64 * kCompactWaves threads
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(s, o);
        if (lane >= o) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    if (wave == 0) {
        const int w = lane < kCompactWaves ? wsum[lane] : 0;
        int t = w;
#pragma unroll
        for (int o = 1; o < kCompactWaves; o <<= 1) {
            const int u = __shfl_up(t, o);
            if (lane >= o) t += u;
        }
        if (lane < kCompactWaves) wsum[lane] = t - w;
        if (lane == kCompactWaves - 1) *total = t;
    }
    __syncthreads();
    return wsum[wave] + s - v;
}

template <int kCompactWG, int kCompactCells, int NC>
__global__ __launch_bounds__(kCompactWG, 8) void k_compact(const uint32_t *__restrict__ cell_kp, const int *__restrict__ cell_cnt,
                                                        Geom g, const uint16_t *__restrict__ lut, uint32_t *__restrict__ sorted_dev,
                                                        uint32_t *__restrict__ cand, int *__restrict__ tbl,
                                                        int *__restrict__ overflow, int bktCap, int cellsCap)
{
    extern __shared__ __attribute__((aligned(16))) int sh[];     // hist[bktCap] | bkey[bktCap] | cnts[cellsCap] (u16) | tx[W0], ty[H0] (u16)
    int *hist = sh;
    uint32_t *bkey = reinterpret_cast<uint32_t *>(hist + NC * bktCap);
    uint16_t *cnts = reinterpret_cast<uint16_t *>(hist + 2 * NC * bktCap);    // min(count, cellCap) of this level's cells
    constexpr int kCompactWaves = kCompactWG / 64;
    __shared__ int wsum[kCompactWaves];
    __shared__ int s_tot, s_base, s_nz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int level = blockIdx.y, img = blockIdx.x;   // x runs fastest: the large levels of every image are dispatched first
    const LevelGeom &L = g.lv[level];
    const int nc = L.nCols * L.nRows;
    const int *cnt = cell_cnt + (size_t)img * g.cells;
    const int cap = g.cellCap;
    const int B = L.nBuckets;
    const int W0 = L.maxBorderX - kMinBorder, H0 = L.maxBorderY - kMinBorder;
    uint16_t *tx = cnts + cellsCap, *ty = tx + W0;

    // candidates of all previous levels of this image, and of this level
    int s = 0, t = 0, over = 0;
    for (int c = tid; c < L.cell0; c += kCompactWG) { const int v = cnt[c]; s += v < cap ? v : cap; }
    for (int c = tid; c < nc; c += kCompactWG) {
        int v = cnt[L.cell0 + c];
        if (v > cap) { v = cap; over = 1; }
        cnts[c] = (uint16_t)v;
        t += v;
    }
    for (int c = nc + tid; c < min(nc + 2 * kCompactCells, cellsCap); c += kCompactWG) cnts[c] = 0;   // the walk reads whole groups of cells, one past the end too
    for (int i = tid; i < W0 + H0; i += kCompactWG) tx[i] = lut[L.lutx + i];        // (ty follows tx in both places)
    (void)block_exclusive_scan<kCompactWaves>(s, wsum, &s_base);
    (void)block_exclusive_scan<kCompactWaves>(t, wsum, &s_tot);
    const int base = s_base, T = s_tot;
    for (int b = tid; b < NC * (B + 1); b += kCompactWG) { hist[b] = 0; bkey[b] = 0; }
    int *tb = tbl + (size_t)img * tbl_ints(g.bucketTotal);   // this image's table block (device memory; DMA'd to the host afterwards)
    if (tid == 0) {
        s_nz = 0;
        tb[kTblLvlOff + level] = base;
        if (level == g.nlevels - 1) tb[kTblLvlOff + g.nlevels] = base + T;
        if (base + T > g.candCap) over = 1;
    }
    if (over) atomicOr(overflow, 1);
    __syncthreads();

    const uint32_t *src = cell_kp + ((size_t)img * g.cells + L.cell0) * cap;
    // Visit every candidate of the level: wave = kCompactCells consecutive cells at a time, lane = index inside the cell; cells
    // with more than 64 candidates take extra trips afterwards (wave-uniform test).  Nothing stands in the way of the memory
    // pipeline: the group's counts are ONE LDS read, the list loads are unconditional (lanes past a cell's count re-read its
    // last entry: same cache line) and the next group's are in flight while this one is processed.  (As `lane < n ? load : 0`
    // every load sat under its own branch behind its own LDS read of the count, and a wait that follows loads under branches
    // is a wait for all of them; with eight cells per group the kernel also needed 83 registers, which halved the resident
    // workgroups: levels 4 - 7 started when levels 0 - 3 were done.  67 -> 50 us per 128 images.)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto walk = [&](auto &&fn) {
        static_assert(kCompactCells == 8 || kCompactCells == 4 || kCompactCells == 2, "the counts are read as one 4-, 8- or 16-byte word");
        constexpr int step = kCompactWaves * kCompactCells;
        const int ncUp = (nc + kCompactCells - 1) & ~(kCompactCells - 1);
        auto load = [&](int c0, int *n, uint32_t *p) {
            if constexpr (kCompactCells == 8) {
                const uint4 q = *reinterpret_cast<const uint4 *>(cnts + c0);
                n[0] = q.x & 0xffff; n[1] = q.x >> 16; n[2] = q.y & 0xffff; n[3] = q.y >> 16;
                n[4] = q.z & 0xffff; n[5] = q.z >> 16; n[6] = q.w & 0xffff; n[7] = q.w >> 16;
            } else if constexpr (kCompactCells == 4) {
                const uint2 q = *reinterpret_cast<const uint2 *>(cnts + c0);
                n[0] = q.x & 0xffff; n[1] = q.x >> 16; n[2] = q.y & 0xffff; n[3] = q.y >> 16;
            } else {
                const uint32_t q = *reinterpret_cast<const uint32_t *>(cnts + c0);
                n[0] = q & 0xffff; n[1] = q >> 16;
            }
#pragma unroll
            for (int u = 0; u < kCompactCells; u++) {
                const uint32_t *cb = src + (size_t)min(c0 + u, nc - 1) * cap;   // wave-uniform (cells past the end have count 0)
                p[u] = cb[min(lane, max(n[u], 1) - 1)];
            }
        };
        auto run = [&](int c0, const int *n, const uint32_t *p) {
            int big = 0;
#pragma unroll
            for (int u = 0; u < kCompactCells; u++) {
                big |= n[u];
                if (lane < n[u]) fn(p[u], (c0 + u) * cap + lane);
            }
            if (big > 64) {
#pragma unroll
                for (int u = 0; u < kCompactCells; u++)
                    for (int k = lane + 64; k < n[u]; k += 64) fn(src[(size_t)(c0 + u) * cap + k], (c0 + u) * cap + k);
            }
        };
        int c0 = wave_u * kCompactCells;
        if (c0 >= nc) return;
        int nA[kCompactCells], nB[kCompactCells];
        uint32_t pA[kCompactCells], pB[kCompactCells];
        load(c0, nA, pA);
        while (true) {
            load(min(c0 + step, ncUp), nB, pB);   // cnts[nc ..] are zero: past the end nothing new is loaded
            run(c0, nA, pA);
            c0 += step;
            if (c0 >= nc) break;
            load(min(c0 + step, ncUp), nA, pA);
            run(c0, nB, pB);
            c0 += step;
            if (c0 >= nc) break;
        }
    };
    // pass 1: histogram of path codes + per-bucket winner key
    walk([&](uint32_t p, int order) {
        const uint32_t code = (uint32_t)tx[cand_x(p)] | (uint32_t)ty[cand_y(p)];
        const uint32_t key = ((uint32_t)cand_resp(p) << 23) | (uint32_t)(kPickOrderMask - order);
        const int slot = (int)code * NC + (lane & (NC - 1));
        atomicAdd(&hist[slot], 1);
        atomicMax(&bkey[slot], key);
    });
    __syncthreads();
    // bucket starts: exclusive scan of the histogram (each thread owns a contiguous run of buckets)
    {
        const int bper = (B + kCompactWG - 1) / kCompactWG;
        const int b0 = tid * bper;
        int bs = 0, nzl = 0;
        for (int k = 0; k < bper; k++)
            if (b0 + k < B) {
                int v = 0;
#pragma unroll
                for (int c = 0; c < NC; c++) v += hist[(b0 + k) * NC + c];
                bs += v;
                nzl += v > 0;
            }
        if (nzl) atomicAdd(&s_nz, nzl);
        int brun = block_exclusive_scan<kCompactWaves>(bs, wsum, &s_tot);
        for (int k = 0; k < bper; k++) {
            if (b0 + k < B) {
#pragma unroll
                for (int c = 0; c < NC; c++) { const int v = hist[(b0 + k) * NC + c]; hist[(b0 + k) * NC + c] = brun; brun += v; }
            }
        }
        if (tid == 0) hist[B * NC] = T;
    }
    __syncthreads();
    int *bs_out = tb + kTblHead + L.bucket0;
    for (int b = tid; b <= B; b += kCompactWG) bs_out[b] = hist[b * NC];
    __syncthreads();
    // pass 2: scatter into bucket order (device memory), hist[] now counts up from each bucket's start
    uint32_t *sd = sorted_dev + (size_t)img * g.candCap + base;
    const int room = g.candCap - base;   // (the device list is sized for the worst case; stay in bounds regardless)
    walk([&](uint32_t p, int) {
        const uint32_t code = (uint32_t)tx[cand_x(p)] | (uint32_t)ty[cand_y(p)];
        const int slot = atomicAdd(&hist[(int)code * NC + (lane & (NC - 1))], 1);
        if (slot < room) sd[slot] = p;
    });
    __syncthreads();   // the workgroup's own global stores are visible to it after the barrier
    uint2 *bb_out = reinterpret_cast<uint2 *>(tb + tbl_win_off(g.bucketTotal)) + L.bucket0;   // BucketWin {key, val}
    // the winner itself: its key holds its position in the cell lists (order = cell * cap + index)
    for (int b = tid; b < B; b += kCompactWG) {
        uint32_t k = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) k = max(k, bkey[b * NC + c]);
        bb_out[b] = uint2{k, k ? src[kPickOrderMask - (k & (uint32_t)kPickOrderMask)] : 0u};
    }
    // The list itself goes over PCIe only when the host can need it: DistributeOctTree divides a depth-D node (one
    // bucket) only after every node reached depth D with fewer than N nodes in total, and at that point the node count
    // equals the number of non-empty buckets.  With nz >= N the host works from the bucket tables alone.
    bool ship = s_nz < L.quota;
    if (ship && base + T > g.hostCandCap) {   // the host copy is smaller than the device list: tell the host instead of truncating
        ship = false;
        if (tid == 0) atomicOr(overflow, 1);
    }
    if (tid == 0) tb[kTblShipped + level] = ship ? 1 : 0;
    if (ship) {
        uint32_t *dst = cand + (size_t)img * g.hostCandCap + base;
        for (int i = tid; i < T; i += kCompactWG) dst[i] = sd[i];   // coalesced copy-out into host-mapped memory
    }
}

// ---------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, OpenCV >= 4 fixed-point path: taps {18,34,48,56,48,34,18}/256,
// horizontal pass in 8.8 (u16), vertical pass in 16.16, (sum + 32768) >> 16,
// BORDER_REFLECT_101 on the un-bordered level (ORBextractor.cpp:1132-1133).
// Workgroup = 128x32 output tile: aligned dword loads -> LDS, horizontal pass in
// packed u16 (every partial sum <= 65280, so v_pk_* arithmetic is exact), vertical
// pass in 32 bits, dword stores.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, Geom g)
{
    constexpr int TW = kBlurTW, TH = kBlurTH;
    // LDS row: [3 pad dwords][left apron dword][TW/4 dwords][right apron dword], so that the tile's own columns start
    // at a 16-byte boundary and are filled with 16-byte loads / ds_write_b128; A0 = dword index of the left apron
    constexpr int ID = TW / 4 + 8, A0 = 3;  // pitch 160 B, a multiple of 16
    constexpr int NP = (TH + 6) / 2;        // row pairs of the horizontal result
    __shared__ __attribute__((aligned(16))) uint32_t in32[(TH + 6) * ID];
    // horizontal result, two vertically adjacent rows per dword: hp[pair][x] = h[2*pair][x] | h[2*pair+1][x] << 16,
    // so that the vertical pass is four v_dot2_u32_u16 per output pixel
    __shared__ __attribute__((aligned(16))) uint32_t hp[NP * TW];
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int level = 0;
#pragma unroll 1
    for (int l = 1; l < g.nlevels; l++)
        if (tile >= g.lv[l].tile0) level = l;
    const LevelGeom &L = g.lv[level];
    const int t = tile - L.tile0;
    const int ty = t / L.tilesX, tx = t - ty * L.tilesX;
    const int x0 = tx * TW, y0 = ty * TH;
    const uint8_t *plane = pyr + (size_t)img * g.imgBytes + L.off;

    if (x0 >= 4 && x0 + TW + 4 <= L.w && y0 >= 3 && y0 + TH + 3 <= L.h) {
        // interior tile (the common case): no border handling; per row 8 x 16 bytes (x0 is a multiple of 128 and rows are
        // 64-byte aligned) + the two apron dwords; all loads are issued before the first LDS store
        constexpr int QR = TW / 16;                          // 16-byte chunks per row
        constexpr int NQ = ((TH + 6) * QR + 255) / 256;      // chunk loads per thread
        const uint8_t *src = plane + (size_t)(y0 - 3) * L.pitch + x0;
        uint4 v[NQ];
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const int i = tid + 256 * k;
            const int r = i / QR, q = i - r * QR;
            v[k] = i < (TH + 6) * QR ? *reinterpret_cast<const uint4 *>(src + (size_t)r * L.pitch + 16 * q) : uint4{0, 0, 0, 0};
        }
        uint32_t ap = 0;
        const int ar = tid >> 1, aside = tid & 1;            // threads 0 .. 2*(TH+6)-1 fetch one apron dword each
        if (tid < 2 * (TH + 6)) ap = *reinterpret_cast<const uint32_t *>(src + (size_t)ar * L.pitch + (aside ? TW : -4));
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const int i = tid + 256 * k;
            const int r = i / QR, q = i - r * QR;
            if (i < (TH + 6) * QR) *reinterpret_cast<uint4 *>(in32 + r * ID + A0 + 1 + 4 * q) = v[k];
        }
        if (tid < 2 * (TH + 6)) in32[ar * ID + A0 + (aside ? TW / 4 + 1 : 0)] = ap;
    } else {
        // edge tile: the same 16-byte fills with REFLECT_101 row indices; columns outside the image come in as whatever
        // lies there (row padding, the neighbouring row -- always inside the image block, which ends with 256 spare bytes)
        // and are then overwritten, per row, with the reflected pixels: only tile columns -4..-1 (left edge) and
        // w .. w+2 (right edge) can ever be read by an output inside the image.
        constexpr int QR = TW / 16;
        constexpr int NQ = ((TH + 6) * QR + 255) / 256;
        auto src_row = [&](int r) {
            int sy = reflect101(y0 + r - 3, L.h);
            sy = sy < 0 ? 0 : (sy >= L.h ? L.h - 1 : sy);   // rows far below the image (tiles over-cover): never used
            return plane + (size_t)sy * L.pitch + x0;
        };
        uint4 v[NQ];
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const int i = tid + 256 * k;
            const int r = i / QR, q = i - r * QR;
            v[k] = i < (TH + 6) * QR ? *reinterpret_cast<const uint4 *>(src_row(r) + 16 * q) : uint4{0, 0, 0, 0};
        }
        uint32_t ap = 0;
        const int ar = tid >> 1, aside = tid & 1;
        if (tid < 2 * (TH + 6) && (aside || x0 > 0)) ap = *reinterpret_cast<const uint32_t *>(src_row(ar) + (aside ? TW : -4));
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const int i = tid + 256 * k;
            const int r = i / QR, q = i - r * QR;
            if (i < (TH + 6) * QR) *reinterpret_cast<uint4 *>(in32 + r * ID + A0 + 1 + 4 * q) = v[k];
        }
        if (tid < 2 * (TH + 6)) in32[ar * ID + A0 + (aside ? TW / 4 + 1 : 0)] = ap;
        const int e = L.w - 1 - x0;   // tile column of the image's last pixel
        if (x0 == 0 || e <= TW + 2) {
            __syncthreads();
            if (tid < TH + 6) {
                uint8_t *row = reinterpret_cast<uint8_t *>(in32 + tid * ID + A0 + 1);   // row[c] = tile column c, c in [-4, TW + 4)
                if (x0 == 0) { row[-1] = row[1]; row[-2] = row[2]; row[-3] = row[3]; row[-4] = row[4]; }
                if (e <= TW + 2) {
#pragma unroll
                    for (int k = 1; k <= 3; k++)
                        if (e + k <= TW + 3) row[e + k] = row[e - k];
                }
            }
        }
    }
    __syncthreads();
    // horizontal: work item = (row pair, group of 4 output pixels); bytes b0..b11 = tile columns 4g .. 4g+11,
    // output pixel j sits at tile column 4g+4+j and needs b[j+1 .. j+7]: two v_dot4_u32_u8 on the byte windows
    // b[j+1..j+4] (taps 18,34,48,56) and b[j+5..j+8] (taps 48,34,18,0), the second accumulating into the first;
    // the sum is <= 65280, so it is exactly the 8.8 fixed-point value OpenCV's horizontal pass produces
    for (int i = tid; i < NP * (TW / 4); i += 256) {
        const int p = i / (TW / 4), gx = i - p * (TW / 4);
        uint32_t h[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t *row = in32 + (2 * p + rr) * ID + A0 + gx;
            const uint32_t d0 = row[0], d1 = row[1], d2 = row[2];
            constexpr uint32_t T1 = 0x38302212u, T2 = 0x00122230u;
#define WIN(hi, lo, sh) ((sh) == 4 ? (hi) : __builtin_amdgcn_alignbyte(hi, lo, sh))
#define HTAP(j) __builtin_amdgcn_udot4(WIN(d2, d1, (j) + 1), T2, __builtin_amdgcn_udot4(WIN(d1, d0, (j) + 1), T1, 0u, false), false)
            h[rr][0] = HTAP(0); h[rr][1] = HTAP(1); h[rr][2] = HTAP(2); h[rr][3] = HTAP(3);
#undef HTAP
#undef WIN
        }
        uint4 v;   // (h[r][x], h[r+1][x]) per dword
        v.x = __builtin_amdgcn_perm(h[1][0], h[0][0], 0x05040100u);
        v.y = __builtin_amdgcn_perm(h[1][1], h[0][1], 0x05040100u);
        v.z = __builtin_amdgcn_perm(h[1][2], h[0][2], 0x05040100u);
        v.w = __builtin_amdgcn_perm(h[1][3], h[0][3], 0x05040100u);
        *reinterpret_cast<uint4 *>(hp + p * TW + 4 * gx) = v;
    }
    __syncthreads();
    // vertical: thread = 4 columns x 4 rows from five row pairs; taps {18,34,48,56,48,34,18} paired with the rows
    const int gx = tid & (TW / 4 - 1), rg = tid / (TW / 4);
    uint32_t P[5][4];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(hp + (2 * rg + q) * TW + 4 * gx);
        P[q][0] = v.x; P[q][1] = v.y; P[q][2] = v.z; P[q][3] = v.w;
    }
#define KK(lo, hi) (u16x2{(unsigned short)(lo), (unsigned short)(hi)})
#define DOT(a, k, c) __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), k, c, false)
    // The output goes through LDS (in32 is dead since the horizontal pass) so that it reaches HBM as whole tiles of the
    // 16 x 8 blurred-plane layout (mcorb_common.h): eight lanes write one 128-byte line.
    static_assert(ID >= TW / 4 && (TH + 6) * ID >= TH * ID, "output staging reuses the input rows");
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t acc[4];
        const int q = k >> 1;   // first row pair of this output row
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t a = 32768u;
            if ((k & 1) == 0) {   // rows 2q .. 2q+6
                a = DOT(P[q][j], KK(18, 34), a);
                a = DOT(P[q + 1][j], KK(48, 56), a);
                a = DOT(P[q + 2][j], KK(48, 34), a);
                a = DOT(P[q + 3][j], KK(18, 0), a);
            } else {              // rows 2q+1 .. 2q+7
                a = DOT(P[q][j], KK(0, 18), a);
                a = DOT(P[q + 1][j], KK(34, 48), a);
                a = DOT(P[q + 2][j], KK(56, 48), a);
                a = DOT(P[q + 3][j], KK(34, 18), a);
            }
            acc[j] = a;
        }
        // (acc + 32768) >> 16 fits a byte: pick byte 2 of each accumulator
        in32[(rg * 4 + k) * ID + gx] = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
    }
    __syncthreads();
    {
        static_assert(TW == 128 && TH == 32, "one 16-byte tile row per thread: 4 x 8 tiles of 16 x 8");
        const int tc = (tid >> 3) & 7, tr = tid >> 6, sr = tid & 7;
        const int x = x0 + 16 * tc, y = y0 + 8 * tr + sr;
        const uint4 o = *reinterpret_cast<const uint4 *>(in32 + (8 * tr + sr) * ID + 4 * tc);
        if (x < L.w && y < L.h) *reinterpret_cast<uint4 *>(blur + (size_t)img * g.imgBytes + L.off + blur_tiled_offset(L.pitch, x, y)) = o;
    }
#undef KK
#undef DOT
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

constexpr int kDescPerWave = 4;   // keypoints per wave
constexpr int kPatchRows = 27, kPatchDw = 12;  // unrotated taps lie within +-13 px: 27 rows x 3 tile columns of 16 bytes

// Reference behaviour (angle = 0): the 27x27 neighbourhood of each keypoint is staged in LDS, then the 512 taps are
// LDS byte reads.  (Gathering the taps straight from global memory is bound by the texture addresser at ~1 lane/clk
// for divergent byte loads.)  The blurred plane is tiled 16 x 8 (mcorb_common.h): the window is the three tile
// columns from (kx - 13) >> 4 on, 81 aligned 16-byte loads touching 12-15 lines of 128 bytes (27 rows of a row-major
// plane were 27 lines, which made this kernel move 5.7x its algorithmic bytes).
__global__ __launch_bounds__(256) void k_describe(const uint8_t *__restrict__ blur, Geom g,
                                                  const uint32_t *__restrict__ sel, const int *__restrict__ nsel,
                                                  uint8_t *__restrict__ desc)
{
    __shared__ __attribute__((aligned(16))) uint32_t patch[4][kDescPerWave][kPatchRows * kPatchDw];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int img = blockIdx.y;
    const int k0 = (blockIdx.x * 4 + wave) * kDescPerWave;
    const int n = nsel[img];
    if (k0 >= n) return;

    static_assert(kPatchDw == 12 && kPatchRows * 3 <= 128, "patch staging assumes 48-byte rows, two trips of 64 lanes");
    // staging task t = 64 * trip + lane = row * 3 + tile column
    int trow[2], tcol[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int t = 64 * q + lane;
        trow[q] = t / 3;
        tcol[q] = t - 3 * trow[q];
    }
    int shift[kDescPerWave];
    uint4 v[kDescPerWave][2];
#pragma unroll
    for (int u = 0; u < kDescPerWave; u++) {
        const int k = k0 + u < n ? k0 + u : n - 1;   // tail keypoints are recomputed, never stored
        const uint32_t s = sel[(size_t)img * g.kcap + k];
        const int level = (int)(s >> 28), ky = (int)((s >> 14) & 0x3fffu), kx = (int)(s & 0x3fffu);
        const LevelGeom &L = g.lv[level];
        const int tc0 = (kx - 13) >> 4;              // keypoints sit >= 19 px from every edge: tile columns tc0 .. tc0+2 exist
        shift[u] = kx - 16 * tc0;                     // 13..28
        const uint8_t *plane = blur + (size_t)img * g.imgBytes + L.off;
        const int tpr = L.pitch >> 4;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int yy = ky - 13 + trow[q];
            v[u][q] = (q == 0 || lane < kPatchRows * 3 - 64)
                          ? *reinterpret_cast<const uint4 *>(plane + ((size_t)((yy >> 3) * tpr + tc0 + tcol[q]) << 7) + ((yy & 7) << 4))
                          : uint4{0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int u = 0; u < kDescPerWave; u++) {
        *reinterpret_cast<uint4 *>(&patch[wave][u][lane * 4]) = v[u][0];
        if (lane < kPatchRows * 3 - 64) *reinterpret_cast<uint4 *>(&patch[wave][u][(64 + lane) * 4]) = v[u][1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // each wave only reads the patches it wrote itself

    // this lane's four test pairs (pairs lane, 64+lane, 128+lane, 192+lane) as LDS byte offsets
    int o0[4], o1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = (j * 64 + lane) * 4;
        o0[j] = (c_pattern[p + 1] + 13) * (kPatchDw * 4) + c_pattern[p];
        o1[j] = (c_pattern[p + 3] + 13) * (kPatchDw * 4) + c_pattern[p + 2];
    }
#pragma unroll
    for (int u = 0; u < kDescPerWave; u++) {
        const uint8_t *c = reinterpret_cast<const uint8_t *>(patch[wave][u]) + shift[u];
        unsigned long long bits[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bits[j] = __ballot(c[o0[j]] < c[o1[j]]);
        if (lane < 4 && k0 + u < n) {
            const unsigned long long b = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
            reinterpret_cast<unsigned long long *>(desc + ((size_t)img * g.kcap + k0 + u) * 32)[lane] = b;
        }
    }
}

// ---------------------------------------------------------------------------
// Blur + BRIEF in one kernel (reference mode, angle = 0).  The reference blurs every level in full
// (ORBextractor.cpp:1132-1133) and then reads 512 taps within +-13 px of each kept keypoint (:105-145); the blurred
// value of a pixel depends on nothing but its own 7 x 7 neighbourhood, so blurring only the 27 x 27 window around each
// of the ~1000 keypoints per image gives the same bytes while touching a quarter of the pixels -- and the blurred
// planes never travel to HBM and back (k_blur + k_describe moved 2 x pyramid + 0.35 GB per 128 images and were bound
// by vector instructions at 17 per blurred pixel).  Keypoints sit >= 19 px from every edge (EDGE_THRESHOLD), so the
// 33 x 33 source window is always inside the image and BORDER_REFLECT_101 never comes into play.
//
// One wave per keypoint at a time (kDescPerWave in sequence, the next window's loads in flight while this one is
// computed); everything is wave-synchronous, no s_barrier.  Per keypoint, in the wave's own LDS:
//   src   34 rows x 48 bytes: image rows ky-16 .. ky+16 from the 16-byte aligned column at or below kx-16
//   hp    horizontal 8.8 sums, two rows per dword (v_dot2 operands), 17 row pairs x 28 columns (kx-13 .. kx+14)
//   out   blurred window, 28 rows x 32 bytes, row y = image row ky-13+y, column o = image column kx-13+o
// The fixed-point arithmetic is k_blur's: v_dot4_u32_u8 on byte windows, v_dot2_u32_u16 down the rows, + 32768 >> 16.
// ---------------------------------------------------------------------------
constexpr int kFdSrcDw = 34 * 12 + 4;    // + one chunk of slack: the (unused) 28th column of the last row reads past it
constexpr int kFdPairs = 17, kFdHpPitch = 28, kFdHpDw = kFdPairs * kFdHpPitch;
constexpr int kFdOutPitchDw = 7, kFdOutDw = 28 * kFdOutPitchDw;   // 28-byte rows: with 32 the tap reads hit the same bank more often (tools/gen_brief_groups.py)
// which descriptor byte (8 consecutive test pairs) the octet g of tap round j compares: slot 8 j + g (brief_groups_31.inc)
__constant__ uint8_t c_brief_groups[64] = {   // [0, 32): slot -> byte; [32, 64): byte -> slot
#include "brief_groups_31.inc"
};
constexpr int kFdWaveDw = kFdSrcDw + kFdHpDw + kFdOutDw;

template <int B>
__device__ __forceinline__ uint32_t fd_window(const uint32_t (&D)[4])
{
    constexpr int q = B >> 2, sh = B & 3;
    if constexpr (sh == 0) return D[q];
    else return __builtin_amdgcn_alignbyte(D[q + 1], D[q], sh);
}

// horizontal pass for a window whose first source byte sits SH bytes into dword `a` of every row
template <int SH>
__device__ __forceinline__ void fd_hpass(const uint32_t *__restrict__ src, uint32_t *__restrict__ hp, int a, int lane)
{
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int i = lane + 64 * t;
        if (i < kFdPairs * 7) {
            const int p = (int)(__umul24((uint32_t)i, 147u) >> 10), j = i - 7 * p;   // i / 7 for i < 128
            uint32_t h[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const uint32_t *row = src + (2 * p + rr) * 12 + a + j;
                uint32_t D[4] = {row[0], row[1], row[2], 0u};
                D[3] = SH == 3 ? row[3] : D[2];   // byte 12 is only ever needed at SH = 3
                constexpr uint32_t T1 = 0x38302212u, T2 = 0x00122230u;   // taps 18,34,48,56 | 48,34,18,0
#define FD_HTAP(o) __builtin_amdgcn_udot4(fd_window<SH + (o) + 4>(D), T2, __builtin_amdgcn_udot4(fd_window<SH + (o)>(D), T1, 0u, false), false)
                h[rr][0] = FD_HTAP(0); h[rr][1] = FD_HTAP(1); h[rr][2] = FD_HTAP(2); h[rr][3] = FD_HTAP(3);
#undef FD_HTAP
            }
            uint4 v;
            v.x = __builtin_amdgcn_perm(h[1][0], h[0][0], 0x05040100u);
            v.y = __builtin_amdgcn_perm(h[1][1], h[0][1], 0x05040100u);
            v.z = __builtin_amdgcn_perm(h[1][2], h[0][2], 0x05040100u);
            v.w = __builtin_amdgcn_perm(h[1][3], h[0][3], 0x05040100u);
            *reinterpret_cast<uint4 *>(hp + p * kFdHpPitch + 4 * j) = v;
        }
    }
}

__device__ __forceinline__ void fd_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // a wave only ever reads what it wrote itself
}

// (65 registers: seven waves per SIMD where the LDS would allow eight; capped at 64 the compiler spills three and the kernel
// takes 131 instead of 110 us; with the tap offsets packed two to a register it needs 61 and runs eight waves: 109 us, no
// gain -- the kernel is held by vector issue and the LDS pipe together, not by waves to switch to)
__global__ __launch_bounds__(256) void k_describe_fused(const uint8_t *__restrict__ pyr, Geom g,
                                                        const uint32_t *__restrict__ sel, const int *__restrict__ nsel,
                                                        uint8_t *__restrict__ desc, int groups, uint8_t *__restrict__ desc_host)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][kFdWaveDw];
    static_assert(kFdSrcDw % 4 == 0 && kFdHpDw % 4 == 0, "16-byte aligned regions");
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    // 1-D grid, remapped so that each XCD works through whole images one after the other: the windows of neighbouring
    // keypoints overlap, and with an image's workgroups dealt round-robin over the eight L2s none of that overlap hit
    // (0.92 GB fetched per 128 images, 7 KB per keypoint).
    const int b = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
    const int img = b / groups;
    const int k0 = ((b - img * groups) * 4 + wave) * kDescPerWave;
    const int n = nsel[img];
    if (k0 >= n) return;
    uint32_t *src = lds[wave], *hp = src + kFdSrcDw, *out = hp + kFdHpDw;

    // staging task t = 64 * q + lane = row * 3 + chunk (99 chunks of 16 bytes)
    int toff[2], tlds[2];
    bool tact[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int t = 64 * q + lane;
        tact[q] = t < 99;
        const int tt = tact[q] ? t : 98;
        const int row = (int)(__umul24((uint32_t)tt, 171u) >> 9), ch = tt - 3 * row;   // tt / 3 for tt < 128
        toff[q] = row | (ch << 8);
        tlds[q] = row * 12 + 4 * ch;
    }
    // this lane's four test pairs as byte offsets into `out`.  Round j, octet g = lane / 8 compares descriptor byte
    // c_brief_groups[8 j + g] (pairs 8 b + lane % 8): the bytes are dealt to the rounds and half-waves so that the 32 lanes of a
    // half hit few LDS banks twice -- 32 instead of 54 LDS cycles for a keypoint's eight ds_read_u8 (the pattern is constant:
    // tools/gen_brief_groups.py).  A ballot word still holds eight whole descriptor bytes, only their order differs.
    int o0[4], o1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = (8 * (int)c_brief_groups[8 * j + (lane >> 3)] + (lane & 7)) * 4;
        o0[j] = (c_pattern[p + 1] + 13) * (kFdOutPitchDw * 4) + c_pattern[p] + 13;
        o1[j] = (c_pattern[p + 3] + 13) * (kFdOutPitchDw * 4) + c_pattern[p + 2] + 13;
    }
    // where descriptor byte (lane % 32) ends up: ballot word bslot >> 3, byte bslot & 7
    const int bslot = c_brief_groups[32 + (lane & 31)];

    // window fetch of keypoint u into (va, vb); kept in plain scalars (an array captured by a lambda ended up in scratch)
    uint4 va, vb;
    int kxn;
#define FD_FETCH(u)                                                                                                         \
    {                                                                                                                       \
        const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane((int)selv, (u));                                            \
        const int level_ = (int)(s_ >> 28), ky_ = (int)((s_ >> 14) & 0x3fffu), kx_ = (int)(s_ & 0x3fffu);                   \
        const LevelGeom &L_ = g.lv[level_];                                                                                 \
        const uint8_t *p0_ = pyr + (size_t)img * g.imgBytes + L_.off + (size_t)(ky_ - 16) * L_.pitch + ((kx_ - 16) & ~15);  \
        va = *reinterpret_cast<const uint4 *>(p0_ + (size_t)__mul24(toff0 & 0xff, L_.pitch) + 16 * (toff0 >> 8));           \
        vb = *reinterpret_cast<const uint4 *>(p0_ + (size_t)__mul24(toff1 & 0xff, L_.pitch) + 16 * (toff1 >> 8));           \
        kxn = kx_;                                                                                                          \
    }
    // the wave's keypoint records, one per lane, read back with v_readlane (tail keypoints are recomputed, never stored)
    const uint32_t selv = sel[(size_t)img * g.kcap + min(k0 + (lane < kDescPerWave ? lane : 0), n - 1)];
    const int toff0 = toff[0], toff1 = toff[1], tlds0 = tlds[0], tlds1 = tlds[1];
    const bool tact1 = tact[1];
    FD_FETCH(0)
#pragma unroll 1
    for (int u = 0; u < kDescPerWave; u++) {
        const int A = (kxn - 16) & 15;   // first source byte of the window inside the staged rows
        *reinterpret_cast<uint4 *>(src + tlds0) = va;
        if (tact1) *reinterpret_cast<uint4 *>(src + tlds1) = vb;
        if (u + 1 < kDescPerWave) FD_FETCH(u + 1)
        fd_wave_sync();
        switch (A & 3) {   // wave-uniform: four copies of the pass with compile-time byte shifts
        case 0: fd_hpass<0>(src, hp, A >> 2, lane); break;
        case 1: fd_hpass<1>(src, hp, A >> 2, lane); break;
        case 2: fd_hpass<2>(src, hp, A >> 2, lane); break;
        default: fd_hpass<3>(src, hp, A >> 2, lane); break;
        }
        fd_wave_sync();
        if (lane < 49) {   // vertical: lane = 4 rows x 4 columns from five row pairs
            const int rg = (int)(__umul24((uint32_t)lane, 147u) >> 10), j = lane - 7 * rg;
            uint32_t P[5][4];
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const uint4 w = *reinterpret_cast<const uint4 *>(hp + (2 * rg + q) * kFdHpPitch + 4 * j);
                P[q][0] = w.x; P[q][1] = w.y; P[q][2] = w.z; P[q][3] = w.w;
            }
#define KK(lo, hi) (u16x2{(unsigned short)(lo), (unsigned short)(hi)})
#define DOT(a, k, c) __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), k, c, false)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t acc[4];
                const int q = k >> 1;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    uint32_t a = 32768u;
                    if ((k & 1) == 0) {
                        a = DOT(P[q][c], KK(18, 34), a);
                        a = DOT(P[q + 1][c], KK(48, 56), a);
                        a = DOT(P[q + 2][c], KK(48, 34), a);
                        a = DOT(P[q + 3][c], KK(18, 0), a);
                    } else {
                        a = DOT(P[q][c], KK(0, 18), a);
                        a = DOT(P[q + 1][c], KK(34, 48), a);
                        a = DOT(P[q + 2][c], KK(56, 48), a);
                        a = DOT(P[q + 3][c], KK(34, 18), a);
                    }
                    acc[c] = a;
                }
                out[(4 * rg + k) * kFdOutPitchDw + j] =
                    __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
            }
#undef KK
#undef DOT
        }
        fd_wave_sync();
        const uint8_t *c = reinterpret_cast<const uint8_t *>(out);
        unsigned long long bits[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bits[j] = __ballot(c[o0[j]] < c[o1[j]]);
        // the four ballot words hold the 32 descriptor bytes in slot order: through 32 bytes of LDS (the row-pair region is idle
        // by now), lane b picks up descriptor byte b from its slot
        if (lane < 4) {
            const unsigned long long b = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
            reinterpret_cast<unsigned long long *>(hp)[lane] = b;
        }
        fd_wave_sync();
        if (lane < 32 && k0 + u < n) {   // lane b stores descriptor byte b
            const uint8_t byte = reinterpret_cast<const uint8_t *>(hp)[bslot];
            desc[((size_t)img * g.kcap + k0 + u) * 32 + lane] = byte;
            // small batches: the host's copy is written here too (host-mapped memory) instead of by a copy behind the kernel
            if (desc_host) desc_host[((size_t)img * g.kcap + k0 + u) * 32 + lane] = byte;
        }
    }
#undef FD_FETCH
}

// IC_Angle enabled (non-reference mode): orientation from the un-blurred level, rotated taps
// (up to +-19 px) gathered from global memory.
__global__ __launch_bounds__(256) void k_describe_oriented(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                           Geom g, const uint32_t *__restrict__ sel,
                                                           const int *__restrict__ nsel, uint8_t *__restrict__ desc,
                                                           float *__restrict__ angles)
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

    // IC_Angle (ORBextractor.cpp:75-102) on the un-blurred level; lanes share the 31 rows
    const uint8_t *c = pyr + coff;
    int m01 = 0, m10 = 0;
    if (lane < 31) {
        const int uu = lane - 15;
        m10 = uu * c[uu];
    }
    for (int v = 1; v <= 15; v++) {
        const int d = c_umax[v];
        if (lane <= 2 * d) {
            const int uu = lane - d;
            const int vp = c[uu + v * pitch], vm = c[uu - v * pitch];
            m01 += v * (vp - vm);
            m10 += uu * (vp + vm);
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
    // the host libm's cosf / sinf are within 0.56 ulp, i.e. almost always the correctly rounded value: round the double
    // result rather than use the device's 1-2 ulp float routines (a last-ulp difference can flip a rounded tap offset)
    const float ca = (float)cos((double)rad), sa = (float)sin((double)rad);

    const uint8_t *cb = blur + (size_t)img * g.imgBytes + L.off;   // tiled 16 x 8 (mcorb_common.h)
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = (j * 64 + lane) * 4;
        const float x0 = (float)c_pattern[p], y0 = (float)c_pattern[p + 1], x1 = (float)c_pattern[p + 2], y1 = (float)c_pattern[p + 3];
        // cvRound(x*b + y*a), cvRound(x*a - y*b) with a = cos, b = sin (:117-118)
        const int oy0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, sa), __fmul_rn(y0, ca)));
        const int ox0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sa)));
        const int oy1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, sa), __fmul_rn(y1, ca)));
        const int ox1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sa)));
        const int t0 = cb[blur_tiled_offset(pitch, kx + ox0, ky + oy0)], t1 = cb[blur_tiled_offset(pitch, kx + ox1, ky + oy1)];
        bits[j] = __ballot(t0 < t1);
    }
    if (lane < 4) {
        const unsigned long long b = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long *>(desc + ((size_t)img * g.kcap + k) * 32)[lane] = b;
    }
}

// ---------------------------------------------------------------------------
// All-pairs Hamming k-NN, k = 2 (BFMatcher knnMatch, MultiCameraFrame.cpp:1053-1055) on the matrix cores, FP4 form.
//
// Hamming(a, b) over 256 bits is a dense contraction: with every bit expanded to +-1,
//   dot(Ea, Eb) = 256 - 2 * hamming.
// +-1 is exact in e2m1 (nibbles 0x2 / 0xA), gfx950's block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 multiplies 64 of them
// per row and step in the time v_mfma_i32_32x32x32_i8 takes for 32 bytes (32.6 cycles either way, tools/fp4_probe.hip), and
// its f32 accumulator holds every value that occurs here exactly (integers below 2^21; the probe checks all 1024 elements
// of random, near-duplicate and extreme tiles against the CPU): 4 K-steps per 32 x 32 tile instead of round 3's 8 + 1, and
// 128 expanded bytes per descriptor instead of 256.  Both block scales are 2^6, so a product is +-4096 and
//   acc = 4096 * (256 - 2 * hamming) + C.
// The tie-break rides in C, the accumulator input of a tile's first step: C = 31 - (train row inside the tile), a
// per-lane constant (16 registers), so a tile's accumulator is its sort key RELATIVE to the tile,
//   key' = 8192 * (128 - hamming) + (31 - row),      global key = key' + 8160 - 32 * tile  (= ... + 8191 - trainIndex),
// and instead of adding the tile term to 16 elements the two running maxima are raised by 32 per tile (they are kept
// relative to the tile being folded).  Larger key <=> smaller (distance, index) = knnMatch's order with its lowest-index
// tie-break.  Every lane owns one query column (16 train rows of it per tile); the running top-2 of two new keys x, y is
//   k1 = max(k1, med3(k0, x, y)), k0 = max3(k0, x, y)
// (the second largest of {k0, k1, x, y} when k1 <= k0), which the compiler pairs up further: 5 vector instructions per 4
// keys (v_med3 x2, v_max3 x3) where the integer form needed 8.  Results are converted back to the
// (distance << 16 | trainIndex) partials k_knn2_finalize merges.
//
// k_expand writes the nibbles in MFMA FRAGMENT ORDER, [tile of 32 descriptors][K-step 4][lane = half * 32 + row][16 B]:
// lane (half, row) of step s holds bits [64 s + 32 half, + 32) of descriptor `row`.  A wave's A or B operand of one K-step
// is 1 KiB of consecutive memory: query fragments are loaded straight into registers (coalesced), train tiles are copied
// linearly into LDS and read back conflict-free with ds_read_b128.  The lane -> k mapping inside a fragment is the same
// for A and B, whatever the hardware's order of k: the sum over k pairs the same bits.
// ---------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kKnnQT = 2;                  // query tiles (32 queries each) per wave
constexpr int kKnnWaves = 4;               // waves per workgroup: 256 queries
constexpr int kTileU4 = 4 * 64;            // uint4 elements of one expanded tile (4 K-steps x 64 lanes = 4 KiB)
static_assert(32 * kKnnQT * kKnnWaves == kKnnQueriesPerBlock, "the host reads the accepted-pair lists in blocks of kKnnQueriesPerBlock queries");
static_assert(kTileU4 * 16 == 32 * kKnnExpandBytes, "k_expand's tile and the scratch size the engine allocates");

// bit descriptors -> e2m1 +-1 nibbles in fragment order; also gathers the sets a match refers to (setmap) into local order
// and writes their clamped counts.  One thread per descriptor (two coalesced 16-byte loads, eight 16-byte stores that
// form 512 contiguous bytes across the 32 lanes of a tile); grid (kcap / 256, local sets)
__global__ __launch_bounds__(256) void k_expand(const uint8_t *__restrict__ desc, const int *__restrict__ counts,
                                                const int *__restrict__ setmap, int kcap, uint4 *__restrict__ E,
                                                int *__restrict__ lcounts)
{
    const int i = blockIdx.y;
    const int src = setmap ? setmap[i] : i;
    const int n = min(max(counts[src], 0), kcap);
    if (blockIdx.x == 0 && threadIdx.x == 0) lcounts[i] = n;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= n) return;   // rows >= n of the last tile are never written: the k-NN kernel masks them
    const uint4 *dp = reinterpret_cast<const uint4 *>(desc + ((size_t)src * kcap + d) * 32);
    const uint4 lo = dp[0], hi = dp[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint4 *out = E + ((size_t)i * (kcap / 32) + (d >> 5)) * kTileU4 + (d & 31);
#pragma unroll
    for (int sh = 0; sh < 8; sh++) {                  // K-step * 2 + half: bits [32 sh, 32 sh + 32)
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // byte -> eight nibbles: bit j lands on bit 4j, then 1 / 0 -> 0xA (-1.0) / 0x2 (+1.0)
            uint32_t x = (w[sh] >> (8 * q)) & 0xffu;
            x = (x | (x << 12)) & 0x000f000fu;
            x = (x | (x << 6)) & 0x03030303u;
            x = (x | (x << 3)) & 0x11111111u;
            o[q] = (x << 3) | 0x22222222u;
        }
        out[(sh >> 1) * 64 + (sh & 1) * 32] = uint4{o[0], o[1], o[2], o[3]};
    }
}

// kKnnStageTiles: train tiles per LDS stage (4 KiB each), double buffered.  The stages are filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers).
// 1-D grid, XCD-aware: workgroups b and b + 8 share an XCD, so the query blocks of one (pair, chunk) unit are given ids
// that are congruent mod 8 -- they stream the same train set and find it in their XCD's L2 after the first has fetched it.
// FOLD (launches whose train sets fit one chunk, i.e. every full batch): the kernel finishes the job itself -- k-NN rows, the
// BruteForceMatch accept test and the accepted pairs compacted per 256-query block (mlist[pair][block * 256 + k], count in
// mcount[pair * qblocks + block]; blocks are in query order, so the host concatenates them) -- instead of writing partials
// for k_knn2_finalize: one launch, a 23-us kernel and a partial-table round trip less per batch.
template <int kKnnStageTiles, int kWavesPerSimd, bool FOLD, int DBG = 0>
__global__ __launch_bounds__(64 * kKnnWaves, kWavesPerSimd) void k_knn2(const uint4 *__restrict__ E, const int *__restrict__ lcounts,
                                                           const int2 *__restrict__ pairs, int kcap, int nchunks, int npairs,
                                                           int qblocks, uint2 *__restrict__ part, int chunkLen,
                                                           float dist_thresh, float ratio, KnnRow *__restrict__ rows,
                                                           uint32_t *__restrict__ mlist, int *__restrict__ mcount)
{
    // (+ two fragments of slack: the tile loop reads two steps ahead, at a stage's last tile past its end)
    __shared__ __attribute__((aligned(16))) uint4 stage[2 * kKnnStageTiles * kTileU4 + 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int slot = blockIdx.x >> 3, grp = slot / qblocks;
    const int unit = grp * 8 + (blockIdx.x & 7);        // (pair, chunk) unit
    if (unit >= npairs * nchunks) return;
    const int pair = unit / nchunks, chunk = unit - pair * nchunks;
    const int2 qt = pairs[pair];
    const int nq = lcounts[qt.x], nt = lcounts[qt.y];
    const int qbi = slot - grp * qblocks;
    const int qb = qbi * kKnnQueriesPerBlock;   // first query of the workgroup
    if (qb >= nq) {
        if (FOLD && tid == 0) mcount[pair * qblocks + qbi] = 0;
        return;
    }
    const int t0 = chunk * chunkLen;
    const int tn = min(nt - t0, chunkLen);              // trains of this chunk
    const int tilesPerSet = kcap / 32;
    const int ntiles = tn > 0 ? (tn + 31) >> 5 : 0;
    const float kNone = -__builtin_inff();              // "no key": below every key, unchanged by the per-tile raise
    float k0[kKnnQT], k1[kKnnQT];                       // running two largest keys, relative to the tile being folded
#pragma unroll
    for (int u = 0; u < kKnnQT; u++) k0[u] = k1[u] = kNone;

    if (tn > 0) {
        // query fragments: kKnnQT tiles x 4 K-steps, 1 KiB coalesced loads (tiles past the set's end stay inside the buffer:
        // kcap is a multiple of 64; their columns are never stored)
        const int qtile0 = (qb >> 5) + wave * kKnnQT;
        v4i Bf[kKnnQT][4];
#pragma unroll
        for (int u = 0; u < kKnnQT; u++) {
            const int tq = min(qtile0 + u, tilesPerSet - 1);
            const uint4 *src = E + ((size_t)qt.x * tilesPerSet + tq) * kTileU4 + lane;
#pragma unroll
            for (int s = 0; s < 4; s++) Bf[u][s] = __builtin_bit_cast(v4i, src[s * 64]);
        }
        // accumulator input of a tile's first step: 31 - (train row inside the tile); C/D layout: row = (e&3) + 8 (e>>2) + 4 (lane>>5)
        v16f Crow;
#pragma unroll
        for (int e = 0; e < 16; e++) Crow[e] = (float)(31 - ((e & 3) + 8 * (e >> 2) + 4 * half));
        const uint4 *Et = E + ((size_t)qt.y * tilesPerSet + (t0 >> 5)) * kTileU4;
        const int nstage = (ntiles + kKnnStageTiles - 1) / kKnnStageTiles;
        static_assert(kTileU4 == 64 * kKnnWaves, "one LDS-DMA round of the workgroup = one tile (wave w moves K-step w)");
        auto fill = [&](int st, int buf) {   // linear copy of the stage, one 1-KiB LDS-DMA per wave and tile; tiles wholly past the end are skipped
#pragma unroll
            for (int k = 0; k < kKnnStageTiles; k++) {
                const int e0 = k * kTileU4 + wave * 64;            // first element of this wave's transfer
                if (st * kKnnStageTiles + k < ntiles)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Et + (size_t)st * kKnnStageTiles * kTileU4 + e0 + lane),
                                                     (__attribute__((address_space(3))) void *)(&stage[buf * kKnnStageTiles * kTileU4 + e0]), 16, 0, 0);
            }
        };
        fill(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // The tile loop is software-pipelined: while the matrix pipe multiplies tile t+1 (into a second accumulator set),
        // the vector ALU folds tile t's keys into the running top-2.  A wave issues in order, and a multiply that
        // accumulates into the register of the one two instructions back stalls the wave until that one is done -- so the
        // top-2 instructions are placed BETWEEN the multiplies in program order (two multiplies, then the ten vector
        // instructions of eight keys; sched_group_barrier keeps the compiler from regrouping them).
        auto fold2 = [&](int u, float x, float y) {
            const float t = __builtin_amdgcn_fmed3f(k0[u], x, y);
            k0[u] = __builtin_fmaxf(__builtin_fmaxf(k0[u], x), y);
            k1[u] = __builtin_fmaxf(k1[u], t);
        };
        auto raise = [&]() {   // the running keys move on to the next tile's frame: its rows are 32 indices further
#pragma unroll
            for (int u = 0; u < kKnnQT; u++) { k0[u] += 32.0f; k1[u] += 32.0f; }
        };
        auto top2 = [&](const v16f (&acc)[kKnnQT]) {
            raise();
#pragma unroll
            for (int u = 0; u < kKnnQT; u++)
#pragma unroll
                for (int e = 0; e < 16; e += 2) fold2(u, acc[u][e], acc[u][e + 1]);
        };
        constexpr int kSc = 0x85858585;   // E8M0 2^6 in every byte: the block scale of both operands
        // Train fragments travel LDS -> registers two K-steps ahead of the multiplies that use them, through a ring of four that
        // runs on across the tiles of a stage (a pair of tiles is eight steps: the ring index of a step is a constant): a wave
        // waits for LDS once per stage, not once per tile.
        v4i Af[4];
        auto tile = [&](const int base, const uint4 *S, int tb, v16f (&acc)[kKnnQT], const v16f (&old)[kKnnQT]) {   // S: the PAIR's first tile; base = 0 / 4
            raise();
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int g = base + s;
                Af[(g + 2) & 3] = __builtin_bit_cast(v4i, S[(g + 2) * 64 + lane]);   // (steps 8, 9: the next pair's first two)
                const v4i a = Af[g & 3];
                const v8i A = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < kKnnQT; u++) {
                    const v8i B = {Bf[u][s][0], Bf[u][s][1], Bf[u][s][2], Bf[u][s][3], 0, 0, 0, 0};
                    acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, s == 0 ? Crow : acc[u], 4, 4, 0, kSc, 0, kSc);
                }
#pragma unroll
                for (int u = 0; u < kKnnQT; u++) {
                    if (DBG & 4) { if (s == 0) fold2(u, old[u][0], old[u][15]); continue; }
                    fold2(u, old[u][4 * s], old[u][4 * s + 1]);
                    fold2(u, old[u][4 * s + 2], old[u][4 * s + 3]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // the ds_read for step g + 2
#pragma unroll
                for (int u = 0; u < kKnnQT; u++) {                              // a multiply, then five folds (four keys) in its shadow
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            }
            // (pins the folds in front of the branch below: nothing else uses their results in this block, and the compiler's
            // sinking pass moved all of them behind it, out of the multiplies' shadow)
#pragma unroll
            for (int u = 0; u < kKnnQT; u++) asm volatile("" : "+v"(k0[u]), "+v"(k1[u]));
            if (tb + 32 > tn) {   // last tile of the set: rows past the end hold stale nibbles
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const bool ok = tb + (e & 3) + 8 * (e >> 2) + 4 * half < tn;
#pragma unroll
                    for (int u = 0; u < kKnnQT; u++) acc[u][e] = ok ? acc[u][e] : kNone;
                }
            }
        };
        static_assert(kKnnStageTiles % 2 == 0, "the tile loop walks a stage as pairs of tiles");
        v16f accA[kKnnQT], accB[kKnnQT];   // even tiles multiply into accB while accA (the odd tile before) is folded, and vice versa
#pragma unroll
        for (int u = 0; u < kKnnQT; u++)
#pragma unroll
            for (int e = 0; e < 16; e++) accA[u][e] = kNone;   // "nothing pending": folding it changes nothing
        for (int st = 0; st < nstage; st++) {
            if (!(DBG & 1) && st + 1 < nstage) fill(st + 1, (st + 1) & 1);   // travels while this stage is multiplied (that buffer was last read a barrier ago)
            const uint4 *Sst = stage + (st & 1) * kKnnStageTiles * kTileU4;
            Af[0] = __builtin_bit_cast(v4i, Sst[lane]);
            Af[1] = __builtin_bit_cast(v4i, Sst[64 + lane]);
            const int tcount = min(ntiles - st * kKnnStageTiles, kKnnStageTiles);   // tiles of this stage
#pragma unroll 1
            for (int pr = 0; 2 * pr < tcount; pr++) {   // two tiles per trip
                const uint4 *S = Sst + pr * 2 * kTileU4;
                const int tb0 = (st * kKnnStageTiles + 2 * pr) * 32;   // first train of the pair, relative to t0
                tile(0, S, tb0, accB, accA);
                if (2 * pr + 1 < tcount) tile(4, S, tb0 + 32, accA, accB);
            }
            if (!(DBG & 1)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next stage has landed
                if (!(DBG & 16)) __syncthreads();
            }
        }
        if (ntiles & 1) top2(accB); else top2(accA);           // the last tile's keys are still pending
    }
    // the two lane halves hold different train rows of the same query columns: merge, convert, store
    const int koff = 8160 - 32 * (ntiles - 1) + (1 << 20);     // tile frame of the last tile -> global key, + 2^20: non-negative
    auto conv = [&](float key) -> uint32_t {
        if (key == kNone) return 0xffffffffu;
        const uint32_t kp = (uint32_t)((int)key + koff);            // 8192 * (256 - hamming) + (8191 - index)
        return ((256u - (kp >> 13)) << 16) | (uint32_t)(t0 + 8191 - (int)(kp & 8191u));
    };
    if (!FOLD) {
        uint2 *out = part + ((size_t)pair * nchunks + chunk) * kcap;
#pragma unroll
        for (int u = 0; u < kKnnQT; u++) {
            const float o0 = __shfl_xor(k0[u], 32), o1 = __shfl_xor(k1[u], 32);
            const float m0 = fmaxf(k0[u], o0), m1 = fmaxf(fminf(k0[u], o0), fmaxf(k1[u], o1));
            const int q = qb + (wave * kKnnQT + u) * 32 + (lane & 31);
            if (half == 0 && q < nq) out[q] = uint2{conv(m0), conv(m1)};
        }
        return;
    }
    // folded finalize (k_knn2_finalize's statements on this workgroup's 256 queries): rows, accept flag, compaction in query order
    __syncthreads();                                     // every wave has read its last tile: the stages are idle
    int *wcnt = reinterpret_cast<int *>(&stage[0]);
    uint32_t packed[kKnnQT];
    unsigned long long bal[kKnnQT];
#pragma unroll
    for (int u = 0; u < kKnnQT; u++) {
        const float o0 = __shfl_xor(k0[u], 32), o1 = __shfl_xor(k1[u], 32);
        const float m0 = fmaxf(k0[u], o0), m1 = fmaxf(fminf(k0[u], o0), fmaxf(k1[u], o1));
        const int q = qb + (wave * kKnnQT + u) * 32 + (lane & 31);
        const uint32_t c0 = conv(m0), c1 = conv(m1);
        const bool v0 = c0 != 0xffffffffu, v1 = c1 != 0xffffffffu;
        const uint32_t d0 = v0 ? c0 >> 16 : 0u, d1 = v1 ? c1 >> 16 : 0u, ti0 = c0 & 0xffffu;
        uint32_t acc = 0;
        if (v0 && v1) {
            const float f0 = (float)d0, f1 = (float)d1;   // DMatch::distance is float
            acc = (f0 < __fmul_rn(ratio, f1)) && !(f0 > dist_thresh);
        }
        if (DBG) acc = 0;   // (timing variants compute garbage: nothing is accepted, the host sees empty lists)
        const bool mine = half == 0 && q < nq;
        if (mine) {
            KnnRow r;
            r.idx = (v0 ? ti0 : 0xffffu) | ((v1 ? (c1 & 0xffffu) : 0xffffu) << 16);
            r.d = d0 | (d1 << 9) | (acc << 18);
            rows[(size_t)pair * kcap + q] = r;
        }
        bal[u] = __ballot(mine && acc != 0);
        packed[u] = ((uint32_t)q << 16) | ti0;
    }
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < kKnnQT; u++) wcnt[wave * kKnnQT + u] = __popcll(bal[u]);
    }
    __syncthreads();
    int before = 0;
    for (int i = 0; i < wave * kKnnQT; i++) before += wcnt[i];
    uint32_t *ml = mlist + (size_t)pair * (qblocks * kKnnQueriesPerBlock) + qb;
#pragma unroll
    for (int u = 0; u < kKnnQT; u++) {
        if ((bal[u] >> lane) & 1) ml[lane_rank(bal[u], before)] = packed[u];
        before += __popcll(bal[u]);
    }
    if (wave == kKnnWaves - 1 && lane == 0) mcount[pair * qblocks + qbi] = before;
}

__device__ __forceinline__ void knn_insert(uint32_t key, uint32_t &k0, uint32_t &k1)
{
    const uint32_t hi = key > k0 ? key : k0;   // k0 <= k1 always: v_max, v_min, v_min
    k1 = hi < k1 ? hi : k1;
    k0 = key < k0 ? key : k0;
}

// Merge chunk partials, emit the knnMatch table and BruteForceMatch's accept
// flag: m0.distance < ratio * m1.distance && !(m0.distance > dist_thresh)
// (MultiCameraFrame.cpp:1061-1063), float arithmetic as in the reference.
__global__ __launch_bounds__(1024) void k_knn2_finalize(const uint2 *__restrict__ part, const int *__restrict__ counts,
                                                        const int2 *__restrict__ pairs, int kcap, int nchunks, int chunkLen,
                                                        float dist_thresh, float ratio, KnnRow *__restrict__ out,
                                                        uint32_t *__restrict__ mlist, int *__restrict__ mcount, int qblocks)
{
    // One workgroup per camera pair: besides the k-NN rows it emits BruteForceMatch's accepted (query, train) pairs
    // compacted in query order (mlist[pair][k] = query << 16 | train, mcount[pair]), so the host does not scan the rows.
    __shared__ int wsum[16];
    __shared__ int s_run;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int2 qt = pairs[pair];
    const int nq = min(max(counts[qt.x], 0), kcap), nt = min(max(counts[qt.y], 0), kcap);
    const int used = (nt + chunkLen - 1) / chunkLen;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nq; base += 1024) {
        const int q = base + tid;
        uint32_t acc = 0, t0 = 0;
        if (q < nq) {
            uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu;
            for (int c = 0; c < used; c += 4) {   // four chunks' partials in flight at a time (the empty key for chunks past the end)
                uint2 p[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    p[u] = c + u < used ? part[((size_t)pair * nchunks + c + u) * kcap + q] : uint2{0xffffffffu, 0xffffffffu};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    knn_insert(p[u].x, k0, k1);
                    knn_insert(p[u].y, k0, k1);
                }
            }
            const bool v0 = k0 != 0xffffffffu, v1 = k1 != 0xffffffffu;
            const uint32_t d0 = v0 ? k0 >> 16 : 0u, d1 = v1 ? k1 >> 16 : 0u;
            if (v0 && v1) {
                const float f0 = (float)d0, f1 = (float)d1;   // DMatch::distance is float
                acc = (f0 < __fmul_rn(ratio, f1)) && !(f0 > dist_thresh);
            }
            t0 = k0 & 0xffffu;
            KnnRow r;
            r.idx = (v0 ? t0 : 0xffffu) | ((v1 ? (k1 & 0xffffu) : 0xffffu) << 16);
            r.d = d0 | (d1 << 9) | (acc << 18);
            out[(size_t)pair * kcap + q] = r;
        }
        const unsigned long long b = __ballot(acc != 0);
        if (lane == 0) wsum[wave] = __popcll(b);
        __syncthreads();
        int before = s_run;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (acc) mlist[(size_t)pair * (qblocks * kKnnQueriesPerBlock) + lane_rank(b, before)] = ((uint32_t)q << 16) | t0;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += wsum[w];
            s_run += tot;
        }
        __syncthreads();
    }
    // (same layout as the folded k_knn2 writes -- one count per 256-query block -- with the whole list in block 0)
    if (tid < qblocks) mcount[pair * qblocks + tid] = tid == 0 ? s_run : 0;
}

// popcount(x) + acc in one instruction; chaining the eight words of a 256-bit XOR through the
// accumulator operand saves the separate adds the compiler otherwise emits
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t hamming256(const ulonglong4 &a, const ulonglong4 &b)
{
    const unsigned long long x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
    uint32_t d = __builtin_popcount((uint32_t)x0);
    d = bcnt_acc((uint32_t)(x0 >> 32), d);
    d = bcnt_acc((uint32_t)x1, d);
    d = bcnt_acc((uint32_t)(x1 >> 32), d);
    d = bcnt_acc((uint32_t)x2, d);
    d = bcnt_acc((uint32_t)(x2 >> 32), d);
    d = bcnt_acc((uint32_t)x3, d);
    d = bcnt_acc((uint32_t)(x3 >> 32), d);
    return d;
}

// ---------------------------------------------------------------------------
// DBoW2 vocabulary-tree descent (TemplatedVocabulary::transform's per-feature part, used by
// MultiCameraFrame::extractFeatureSingle, MultiCameraFrame.cpp:257): from the root, move to the child
// with the smallest Hamming distance (strict '<': the first child wins ties) until a leaf; remember the
// node reached at depth `nid_level`.  One descriptor per lane; the children of a node are stored
// contiguously (descriptor + node id), k*32 bytes per step.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bow_descend(const uint8_t *__restrict__ desc, int n, const int *__restrict__ child_start,
                                                     const int *__restrict__ child_count, const ulonglong4 *__restrict__ child_desc,
                                                     const int *__restrict__ child_id, const int *__restrict__ word_id,
                                                     const double *__restrict__ weight, int nid_level, BowRes *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const ulonglong4 q = *reinterpret_cast<const ulonglong4 *>(desc + (size_t)i * 32);
    int node = 0, nid = 0, level = 0;
    int cc = child_count[0];
    while (cc > 0) {
        ++level;
        const int cs = child_start[node];
        uint32_t best = 0xffffffffu;
        int bj = 0;
        for (int j = 0; j < cc; j++) {
            const uint32_t d = hamming256(q, child_desc[cs + j]);
            if (d < best) { best = d; bj = j; }
        }
        node = child_id[cs + bj];
        if (level == nid_level) nid = node;
        cc = child_count[node];
    }
    out[i] = BowRes{word_id[node], nid, weight[node]};
}

// ---------------------------------------------------------------------------
// BoW-guided intra-rig matching, the data-parallel half (MultiCameraFrame::computeIntraMatches(matches,
// words_), MultiCameraFrame.cpp:708-745): for feature a of camera c1 and every camera c2 > c1, the best
// and second-best Hamming distance among c2's features that fell into the same vocabulary node, skipping
// candidates whose row differs by 50 px or more; strict '<' so the first minimum wins.  The serial
// track bookkeeping that consumes this table stays on the host.
// All frames of a batch in one launch: blockIdx.z = frame, blockIdx.y = camera pair (c1 < c2).  Per frame f the
// index tables are laid out for image index m = f * ncams + c: slot_of / node_feats / yv at m * kcap,
// node_range at (rg_base[f] + slot) * ncams + c; out at ((f * npairs + pair) * kcap + a).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bow_best2(const uint8_t *__restrict__ desc, int img0, int kcap, int ncams,
                                                   const float *__restrict__ yv, const int *__restrict__ slot_of,
                                                   const int2 *__restrict__ node_range, const int *__restrict__ rg_base,
                                                   const int *__restrict__ node_feats, const int *__restrict__ nfeat,
                                                   int4 *__restrict__ out)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.z, npairs = ncams * (ncams - 1) / 2;
    int c1 = 0, rem = blockIdx.y;   // pair index -> (c1, c2), pairs in (0,1), (0,2), .., (1,2), .. order
    while (rem >= ncams - 1 - c1) { rem -= ncams - 1 - c1; c1++; }
    const int c2 = c1 + 1 + rem;
    const int m1 = f * ncams + c1, m2 = f * ncams + c2;
    if (a >= nfeat[m1]) return;
    int4 r = int4{-1, 0x7fffffff, 0x7fffffff, 0};
    const int slot = slot_of[(size_t)m1 * kcap + a];
    if (slot >= 0) {
        const int2 rg = node_range[(size_t)(rg_base[f] + slot) * ncams + c2];
        const ulonglong4 q = *reinterpret_cast<const ulonglong4 *>(desc + ((size_t)(img0 + m1) * kcap + a) * 32);
        const float y1 = yv[(size_t)m1 * kcap + a];
        for (int j = 0; j < rg.y; j++) {
            const int b = node_feats[(size_t)m2 * kcap + rg.x + j];
            if (fabsf(__fsub_rn(y1, yv[(size_t)m2 * kcap + b])) >= 50.f) continue;
            const int d = (int)hamming256(q, *reinterpret_cast<const ulonglong4 *>(desc + ((size_t)(img0 + m2) * kcap + b) * 32));
            if (d < r.y) { r.x = j; r.z = r.y; r.y = d; }
            else if (d < r.z) r.z = d;
        }
    }
    out[((size_t)f * npairs + blockIdx.y) * kcap + a] = r;
}

// ---------------------------------------------------------------------------
// Device -> pinned host copy as a SMALL kernel.  hipMemcpyAsync to pinned memory runs as a blit kernel on this stack
// (__amd_rocclr_copyBuffer; HSA_ENABLE_SDMA changes nothing), and the kernel of the job that runs beside it is stretched
// to the copy's length (k_expand: 12 us alone, 150 us beside the descriptor read-back).  The link moves ~50 GB/s whatever
// feeds it; kCopyWG workgroups looping over the buffer with 16-byte accesses, four loads in flight per lane, feed it as
// well and leave the neighbour alone (k_expand 23 us).  Used for single-slot rigs only: with six slots in flight the
// runtime's copy still gives 4-6 % more frames/s (profiles/r03_copy_kernel_ab.txt; the smaller the grid the closer: 1024
// workgroups 30.7 k, 128: 30.9 k, 24: 32.3 k, 4: 33.5 k, runtime 34.0 k frames/s).
// ---------------------------------------------------------------------------
constexpr int kCopyWG = 24;
__global__ __launch_bounds__(256) void k_copy_to_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const v4i *s = reinterpret_cast<const v4i *>(src);
    v4i *d = reinterpret_cast<v4i *>(dst);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {   // four 16-byte loads in flight per lane
        const v4i a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + stride);
        const v4i c = __builtin_nontemporal_load(s + i + 2 * stride), e = __builtin_nontemporal_load(s + i + 3 * stride);
        __builtin_nontemporal_store(a, d + i);
        __builtin_nontemporal_store(b, d + i + stride);
        __builtin_nontemporal_store(c, d + i + 2 * stride);
        __builtin_nontemporal_store(e, d + i + 3 * stride);
    }
    for (; i < n16; i += stride) {
        const v4i v = __builtin_nontemporal_load(s + i);
        __builtin_nontemporal_store(v, d + i);
    }
}

void launch_copy_to_host(hipStream_t st, const void *src_dev, void *dst_host_mapped, size_t bytes)
{
    const size_t n16 = (bytes + 15) / 16;   // both buffers are allocated in multiples of 16 bytes
    if (!n16) return;
    static const int wg_env = getenv("MCORB_COPY_WG") ? atoi(getenv("MCORB_COPY_WG")) : 0;
    const size_t cap = wg_env > 0 ? (size_t)wg_env : (size_t)kCopyWG;
    const int wgs = (int)((n16 + 255) / 256 < cap ? (n16 + 255) / 256 : cap);
    hipLaunchKernelGGL(k_copy_to_host, dim3(wgs), dim3(256), 0, st, reinterpret_cast<const uint4 *>(src_dev),
                       reinterpret_cast<uint4 *>(dst_host_mapped), n16);
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

void launch_pyramid(hipStream_t st, uint8_t *pyr, const Geom &g, const ResizeTap *tabs, const int *win, int nimg)
{
    for (int l = 1; l < g.nlevels; l++) {
        dim3 grid((g.lv[l].w + 255) / 256, (g.lv[l].h + kResizeRows - 1) / kResizeRows, nimg);
        const int srcPitch = win[2 * l], srcRows = win[2 * l + 1];   // LDS window, sized on the host from the tables
        const int chunks = (srcPitch >> 4) * srcRows, nf = (chunks + 255) / 256;   // 16-byte loads per thread to fill the window
        const size_t lds = (size_t)srcPitch * srcRows + 16;   // + the spare chunk idle fill lanes write
        if (nf <= 4) hipLaunchKernelGGL(k_resize<4>, grid, dim3(256), lds, st, pyr, g, l, tabs, srcPitch, srcRows);
        else if (nf <= 8) hipLaunchKernelGGL(k_resize<8>, grid, dim3(256), lds, st, pyr, g, l, tabs, srcPitch, srcRows);
        else hipLaunchKernelGGL(k_resize<0>, grid, dim3(256), lds, st, pyr, g, l, tabs, srcPitch, srcRows);
    }
}

void fast_layout(const Geom &g, int &tp, int &rows)
{
    // LDS tile of a cell: ROI rows x pitch (3 phase bytes + wCell + 6, rounded up to whole 16-byte chunks)
    int pitch = 0;
    rows = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int p = ((3 + g.lv[l].wCell + 6 + 15) >> 4) << 4;
        pitch = pitch > p ? pitch : p;
        rows = rows > g.lv[l].hCell + 6 ? rows : g.lv[l].hCell + 6;
    }
    tp = pitch <= 48 ? 48 : (pitch <= 64 ? 64 : 80);   // wCell < 70 by construction (build_geometry): pitch <= 80
}

void launch_fast(hipStream_t st, const uint8_t *pyr, const Geom &g, int iniTh, int minTh, const uint32_t *cellRec,
                 uint32_t *cell_kp, int *cell_cnt, int nimg)
{
    // LDS per wave: the tile, the score map (four rows fewer), the bounded survivor list
    int tp, rows;
    fast_layout(g, tp, rows);
    static const int cap_env = getenv("MCORB_FAST_LISTCAP") ? atoi(getenv("MCORB_FAST_LISTCAP")) : 0;   // tuning knob; any multiple of 8 >= 512 is safe
    const int cap = cap_env >= 512 ? (cap_env & ~7) : kFastListCap;
    const int tileB = (rows * tp + 15) & ~15, scB = ((rows - 4) * tp + 15) & ~15;
    // MCORB_FAST_LDS_PAD (experiment): extra bytes per one-wave workgroup -- 29 waves of 5.4 KiB fill a CU's
    // 160 KiB, so no other kernel's workgroup fits beside them; padded to 6.6 KiB there are 24 and 32 KiB stay free
    static const int pad_env = getenv("MCORB_FAST_LDS_PAD") ? atoi(getenv("MCORB_FAST_LDS_PAD")) : 0;
    const size_t lds = 16 + (size_t)tileB + 16 + (size_t)scB + 16 + (size_t)cap * 2 + (size_t)(pad_env > 0 ? (pad_env & ~15) : 0);
    dim3 grid(g.cells, nimg);
    if (iniTh < 0) iniTh = 0;   // (pass 1's sign tests rely on thresholds in 0 .. 255; FAST thresholds are)
    if (minTh < 0) minTh = 0;
    if (iniTh > 255) iniTh = 255;   // no pixel passes at 255 or above either way
    if (minTh > 255) minTh = 255;
    if (tp == 48) hipLaunchKernelGGL(k_fast_cells<48>, grid, dim3(64), lds, st, pyr, g, iniTh, minTh, tileB, scB, cap, cellRec, cell_kp, cell_cnt);
    else if (tp == 64) hipLaunchKernelGGL(k_fast_cells<64>, grid, dim3(64), lds, st, pyr, g, iniTh, minTh, tileB, scB, cap, cellRec, cell_kp, cell_cnt);
    else hipLaunchKernelGGL(k_fast_cells<80>, grid, dim3(64), lds, st, pyr, g, iniTh, minTh, tileB, scB, cap, cellRec, cell_kp, cell_cnt);
}

void launch_compact(hipStream_t st, const uint32_t *cell_kp, const int *cell_cnt, const Geom &g, const uint16_t *lut,
                    uint32_t *sorted_dev, uint32_t *cand, int *tbl, int *overflow, int nimg)
{
    int maxb = 1, maxc = 1, maxwh = 2;
    for (int l = 0; l < g.nlevels; l++) {
        maxb = maxb > g.lv[l].nBuckets ? maxb : g.lv[l].nBuckets;
        maxc = maxc > g.lv[l].nCols * g.lv[l].nRows ? maxc : g.lv[l].nCols * g.lv[l].nRows;
        const int wh = (g.lv[l].maxBorderX - kMinBorder) + (g.lv[l].maxBorderY - kMinBorder);
        maxwh = maxwh > wh ? maxwh : wh;
    }
    const int bktCap = (maxb + 1 + 3) & ~3;
    static const int wg_env = getenv("MCORB_COMPACT_WG") ? atoi(getenv("MCORB_COMPACT_WG")) : 0;
    static const int cells_env = getenv("MCORB_COMPACT_CELLS") ? atoi(getenv("MCORB_COMPACT_CELLS")) : 0;
    // 512 threads per (level, image) workgroup and four cells per group for a full batch (1024 threads: 42 against 39.5 us per
    // 128 images; two cells: 45.6; eight cells need more than 64 registers, which halves the resident workgroups); a small
    // batch is the latency of its level-0 workgroup, shorter with 1024 threads
    const int wg = wg_env == 1024 || wg_env == 512 ? wg_env : (nimg <= 8 ? 1024 : 512), cellsInFlight = cells_env == 2 ? 2 : 4;
    const int cellsCap = (maxc + 8 + 7) & ~7;   // u16 entries, a multiple of 8
    const size_t ldsTail = (size_t)(cellsCap + maxwh + 8) * sizeof(uint16_t);
    const bool fits = (size_t)(2 * kCompactCopies * bktCap) * sizeof(int) + ldsTail <= 64 * 1024;
    static const bool one_env = getenv("MCORB_COMPACT_ONE_COPY") != nullptr;   // (test / comparison knob)
    const int copies = fits && !one_env ? kCompactCopies : 1;
    const size_t lds = (size_t)(2 * copies * bktCap) * sizeof(int) + ldsTail;
    const dim3 grid(nimg, g.nlevels);
#define MCORB_COMPACT_LAUNCH(WG_, C_) \
    do { if (copies == 1) hipLaunchKernelGGL((k_compact<WG_, C_, 1>), grid, dim3(WG_), lds, st, cell_kp, cell_cnt, g, lut, sorted_dev, cand, tbl, overflow, bktCap, cellsCap); \
         else hipLaunchKernelGGL((k_compact<WG_, C_, kCompactCopies>), grid, dim3(WG_), lds, st, cell_kp, cell_cnt, g, lut, sorted_dev, cand, tbl, overflow, bktCap, cellsCap); } while (0)
    if (wg == 1024) { if (cellsInFlight == 2) MCORB_COMPACT_LAUNCH(1024, 2); else MCORB_COMPACT_LAUNCH(1024, 4); }
    else { if (cellsInFlight == 2) MCORB_COMPACT_LAUNCH(512, 2); else MCORB_COMPACT_LAUNCH(512, 4); }
#undef MCORB_COMPACT_LAUNCH
}

void launch_blur(hipStream_t st, const uint8_t *pyr, uint8_t *blur, const Geom &g, int nimg)
{
    dim3 grid(g.tiles, nimg);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, st, pyr, blur, g);
}

void launch_describe(hipStream_t st, const uint8_t *pyr, const uint8_t *blur, const Geom &g, const uint32_t *sel,
                     const int *nsel, int orientation, uint8_t *desc, float *angles, int nimg, uint8_t *desc_host)
{
    if (orientation) {
        dim3 grid((g.kcap + 3) / 4, nimg);
        hipLaunchKernelGGL(k_describe_oriented, grid, dim3(256), 0, st, pyr, blur, g, sel, nsel, desc, angles);
    } else {
        dim3 grid((g.kcap + 4 * kDescPerWave - 1) / (4 * kDescPerWave), nimg);
        if (blur) hipLaunchKernelGGL(k_describe, grid, dim3(256), 0, st, blur, g, sel, nsel, desc);   // from blurred planes (k_blur ran)
        else hipLaunchKernelGGL(k_describe_fused, dim3(grid.x * nimg), dim3(256), 0, st, pyr, g, sel, nsel, desc, (int)grid.x, desc_host);
    }
}

void launch_knn2(hipStream_t st, const uint8_t *desc, const int *counts, const int *setmap, int nsets, const int2 *pairs, int npairs,
                 int kcap, void *expanded, int *lcounts, uint2 *part, float dist_thresh, float ratio, KnnRow *out, uint32_t *mlist,
                 int *mcount, hipEvent_t ev_exp, hipEvent_t ev_mid)
{
    uint4 *E = reinterpret_cast<uint4 *>(expanded);
    hipLaunchKernelGGL(k_expand, dim3((kcap + 255) / 256, nsets), dim3(256), 0, st, desc, counts, setmap, kcap, E, lcounts);
    if (ev_exp) (void)hipEventRecord(ev_exp, st);
    const int chunkLen = knn_chunk_len(npairs), nchunks = (kcap + chunkLen - 1) / chunkLen;
    const int qblocks = knn_qblocks(kcap), units = npairs * nchunks;
    dim3 grid(8 * qblocks * ((units + 7) / 8));
    // 130 VGPRs: 3 waves per SIMD.  (Capping at 128 for 4 waves spills one query fragment into scratch: 177 vs 162 us.)
    static const int dbg = getenv("MCORB_KNN_DBG") ? atoi(getenv("MCORB_KNN_DBG")) : 0;   // timing experiments only (scripts/knn_dbg.sh): results are wrong
    // bit 0: no LDS fill, no barrier; bit 2: no top-2 folds; bit 4: fill, but no barrier
#define MCORB_KNN_DBG_LAUNCH(D_) if (dbg == D_) { hipLaunchKernelGGL((k_knn2<4, 3, true, D_>), grid, dim3(64 * kKnnWaves), 0, st, E, lcounts, pairs, kcap, nchunks, npairs, qblocks, part, chunkLen, dist_thresh, ratio, out, mlist, mcount); if (ev_mid) (void)hipEventRecord(ev_mid, st); return; }
    if (nchunks == 1 && dbg) { MCORB_KNN_DBG_LAUNCH(1) MCORB_KNN_DBG_LAUNCH(4) MCORB_KNN_DBG_LAUNCH(5) MCORB_KNN_DBG_LAUNCH(16) }
    if (nchunks == 1) {
        hipLaunchKernelGGL((k_knn2<4, 3, true>), grid, dim3(64 * kKnnWaves), 0, st, E, lcounts, pairs, kcap, nchunks, npairs, qblocks, part, chunkLen,
                           dist_thresh, ratio, out, mlist, mcount);
        if (ev_mid) (void)hipEventRecord(ev_mid, st);
        return;
    }
    hipLaunchKernelGGL((k_knn2<4, 3, false>), grid, dim3(64 * kKnnWaves), 0, st, E, lcounts, pairs, kcap, nchunks, npairs, qblocks, part, chunkLen,
                       dist_thresh, ratio, out, mlist, mcount);
    if (ev_mid) (void)hipEventRecord(ev_mid, st);
    hipLaunchKernelGGL(k_knn2_finalize, dim3(npairs), dim3(1024), 0, st, part, lcounts, pairs, kcap, nchunks, chunkLen, dist_thresh, ratio, out, mlist,
                       mcount, qblocks);
}

void launch_bow_best2(hipStream_t st, const uint8_t *desc, int img0, int kcap, int ncams, int nframes, const float *yv,
                      const int *slot_of, const int2 *node_range, const int *rg_base, const int *node_feats, const int *nfeat, int4 *out)
{
    if (ncams < 2 || nframes < 1) return;
    dim3 grid((kcap + 255) / 256, ncams * (ncams - 1) / 2, nframes);
    hipLaunchKernelGGL(k_bow_best2, grid, dim3(256), 0, st, desc, img0, kcap, ncams, yv, slot_of, node_range, rg_base, node_feats, nfeat, out);
}

void launch_bow_descend(hipStream_t st, const uint8_t *desc, int n, const int *child_start, const int *child_count,
                        const void *child_desc, const int *child_id, const int *word_id, const double *weight, int nid_level,
                        BowRes *out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bow_descend, dim3((n + 255) / 256), dim3(256), 0, st, desc, n, child_start, child_count,
                       reinterpret_cast<const ulonglong4 *>(child_desc), child_id, word_id, weight, nid_level, out);
}

}  // namespace mcorb
