// mcorb_kernels.h -- launch wrappers of the gfx950 kernels (mcorb_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "mcorb_common.h"
#include "mcorb_signal.h"

namespace mcorb {

constexpr int kKnnChunk = 4096;  // train descriptors per k-NN partial (one workgroup column); the in-chunk index must stay below 8192
// Chunk length a launch of `npairs` camera pairs uses.  A workgroup walks one chunk of the train set for 256 queries: with the
// 192 pairs of a 32-frame batch that is 1 536 workgroups of 63 tiles, plenty; the 6 pairs of ONE rig frame (how MC-SLAM calls
// the front-end) are 48 workgroups on 256 CUs, each walking all 63 tiles: 44 us of a 0.6 ms frame.  Few pairs -> short chunks
// -> more, shorter workgroups (the partials are merged by k_knn2_finalize either way).  Multiples of 64 (one LDS stage).
__host__ __device__ inline int knn_chunk_len(int npairs) { return npairs <= 12 ? 256 : npairs <= 24 ? 512 : npairs <= 48 ? 1024 : kKnnChunk; }
// The accepted (query << 16 | train) pairs of a camera pair come back in blocks of 256 queries: mlist[pair][block * 256 + k],
// k < mcount[pair * knn_qblocks(kcap) + block]; blocks are in query order (concatenate them).
constexpr int kKnnQueriesPerBlock = 256;   // = the k-NN workgroup's queries (64 per wave x 4 waves; static_assert in mcorb_kernels.hip)
inline int knn_qblocks(int kcap) { return (kcap + kKnnQueriesPerBlock - 1) / kKnnQueriesPerBlock; }
inline size_t knn_mlist_stride(int kcap) { return (size_t)knn_qblocks(kcap) * kKnnQueriesPerBlock; }
// partials (uint2) a slot needs for jobs of up to max_pairs pairs at capacity kcap
inline size_t knn_part_entries(int max_pairs, int kcap)
{
    size_t need = 0;
    for (int np : {12, 24, 48, max_pairs}) {
        const int n = np < max_pairs ? np : max_pairs, cl = knn_chunk_len(n);
        const size_t e = (size_t)n * ((kcap + cl - 1) / cl) * kcap;
        need = need > e ? need : e;
    }
    return need;
}
constexpr int kKnnExpandBytes = 128;   // bytes of one descriptor expanded to 256 e2m1 nibbles of +-1 (k_expand)

// one row of the knnMatch(k=2) table, 8 bytes so the PCIe write-back stays small:
// idx = trainIdx0 | trainIdx1 << 16 (0xffff = absent), d = dist0 | dist1 << 9 | accept << 18
// (accept = BruteForceMatch's ratio + threshold test)
struct KnnRow { uint32_t idx, d; };
__host__ __device__ inline int knn_idx0(const KnnRow &r) { return (r.idx & 0xffffu) == 0xffffu ? -1 : (int)(r.idx & 0xffffu); }
__host__ __device__ inline int knn_idx1(const KnnRow &r) { return (r.idx >> 16) == 0xffffu ? -1 : (int)(r.idx >> 16); }
__host__ __device__ inline int knn_d0(const KnnRow &r) { return knn_idx0(r) < 0 ? -1 : (int)(r.d & 0x1ffu); }
__host__ __device__ inline int knn_d1(const KnnRow &r) { return knn_idx1(r) < 0 ? -1 : (int)((r.d >> 9) & 0x1ffu); }
__host__ __device__ inline bool knn_accept(const KnnRow &r) { return (r.d >> 18) & 1u; }

hipError_t upload_umax(const int umax[16]);
void launch_stage_f32(hipStream_t st, const float *src, int w, int h, int pitch_f, int channels, size_t img_stride_f,
                      uint8_t *pyr, const Geom &g, int nimg);
// win[2*l], win[2*l+1]: LDS source-window pitch (bytes, multiple of 4) and rows of level l's resize workgroups
void launch_pyramid(hipStream_t st, uint8_t *pyr, const Geom &g, const ResizeTap *tabs, const int *win, int nimg);
// k_fast_cells' per-cell records for this geometry (built once per rig, uploaded by the caller): ROI origin, size,
// pass-1 lane layout; appended to `tab`, returns their dword offset
int fast_cell_table(const Geom &g, std::vector<uint32_t> &tab);
void launch_fast(hipStream_t st, const uint8_t *pyr, const Geom &g, int iniTh, int minTh, const uint32_t *cellRec,
                 uint32_t *cell_kp, int *cell_cnt, int nimg);
// tbl: device table blocks (tbl_ints(g.bucketTotal) ints per image, layout in mcorb_common.h); cand / overflow: host-mapped
// lut: path-code tables (LevelGeom::lutx / luty)
void launch_compact(hipStream_t st, const uint32_t *cell_kp, const int *cell_cnt, const Geom &g, const uint16_t *lut,
                    uint32_t *sorted_dev, uint32_t *cand, int *tbl, int *overflow, int nimg);
void launch_blur(hipStream_t st, const uint8_t *pyr, uint8_t *blur, const Geom &g, int nimg);
void launch_describe(hipStream_t st, const uint8_t *pyr, const uint8_t *blur, const Geom &g, const uint32_t *sel,
                     const int *nsel, int orientation, uint8_t *desc, float *angles, int nimg,
                     uint8_t *desc_host = nullptr);   // desc_host (k_describe_fused only): host-mapped second destination
// desc / counts: `kcap`-strided bit descriptors and their counts; setmap[i] (null = identity) = the set that becomes local
// set i, i < nsets; pairs: (query, train) LOCAL set indices; expanded: nsets * kcap * kKnnExpandBytes bytes of scratch;
// lcounts: nsets ints (receives the clamped counts); part: knn_part_entries(max pairs per launch, kcap) partials
void launch_knn2(hipStream_t st, const uint8_t *desc, const int *counts, const int *setmap, int nsets, const int2 *pairs, int npairs,
                 int kcap, void *expanded, int *lcounts, uint2 *part, float dist_thresh, float ratio, KnnRow *out, uint32_t *mlist,
                 int *mcount, hipEvent_t ev_exp, hipEvent_t ev_mid);   // events (optional): after k_expand, after k_knn2

// DistributeOctTree's list discipline + operator()'s assembly on the GPU (mcorb_select_gpu.hip).  tbl: k_compact's device table
// blocks, sorted: its bucket-sorted candidate lists (read only where a tree goes below the bucketing depth); sel_val / sel_cnt: select_cap(g) retained candidates and their count per (image, level), -1 = the level needs the host
// stage (fallback |= 1); sel / nsel: what the descriptor kernel and the matcher read; resp / mono: FAST responses and monoIndex
// for the host's keypoint records; fallback bit 1 = more than kcap keypoints.
int select_cap(const Geom &g);
bool select_fits(const Geom &g);   // false: the level trees of this geometry do not fit a wave's LDS (very large feature budgets)
// deep_cap: a bucket with more candidates than this is not scanned node by node when a tree goes below the bucketing depth
hipError_t launch_select(hipStream_t st, const int *tbl, const uint32_t *sorted, const Geom &g, uint32_t *sel_val, int *sel_cnt, int *fallback, int nimg, int deep_cap = 4096);
void launch_assemble(hipStream_t st, const uint32_t *sel_val, const int *sel_cnt, const Geom &g, const float *scale, int lap0, int lap1,
                     uint32_t *sel, uint8_t *resp, int *nsel, int *mono, int *fallback, int nimg,
                     // optional (all or none): host-mapped copies of sel / resp and a per-image signal word set behind them
                     uint32_t *sel_h = nullptr, uint8_t *resp_h = nullptr, unsigned long long *sig_h = nullptr);
// (the signal word of an image, k_assemble -> host: mcorb_signal.h)
// test hook: std::sort's permutation of n 64-bit entries (upper halves compared) by one wave (wave_std_sort)
hipError_t sort_selftest(const uint64_t *in_dev, int n, uint64_t *out_dev);

// device -> host-mapped pinned memory with a small grid (instead of the runtime's blit kernel); sizes rounded up to 16 bytes
void launch_copy_to_host(hipStream_t st, const void *src_dev, void *dst_host_mapped, size_t bytes);

// all frames [0, nframes) of a batch whose first image is img0 of `desc` (kcap-strided sets); tables indexed by the
// batch-local image index f * ncams + c (see k_bow_best2)
void launch_bow_best2(hipStream_t st, const uint8_t *desc, int img0, int kcap, int ncams, int nframes, const float *yv,
                      const int *slot_of, const int2 *node_range, const int *rg_base, const int *node_feats, const int *nfeat, int4 *out);
void launch_bow_descend(hipStream_t st, const uint8_t *desc, int n, const int *child_start, const int *child_count,
                        const void *child_desc, const int *child_id, const int *word_id, const double *weight, int nid_level,
                        BowRes *out);

}  // namespace mcorb
