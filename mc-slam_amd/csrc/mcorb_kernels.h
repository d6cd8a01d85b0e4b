// mcorb_kernels.h -- launch wrappers of the gfx950 kernels (mcorb_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "mcorb_common.h"

namespace mcorb {

constexpr int kKnnChunk = 256;   // train descriptors staged in LDS per workgroup (8 KiB)

// one row of the knnMatch(k=2) table; bit 30 of d1 carries BruteForceMatch's accept flag
struct KnnRow { int32_t idx0, idx1, d0, d1; };

hipError_t upload_umax(const int umax[16]);
void launch_stage_f32(hipStream_t st, const float *src, int w, int h, int pitch_f, int channels, size_t img_stride_f,
                      uint8_t *pyr, const Geom &g, int nimg);
void launch_pyramid(hipStream_t st, uint8_t *pyr, const Geom &g, const ResizeTap *tabs, int nimg);
// ev_mid (optional) is recorded between the two kernels so each can be timed on its own
void launch_fast(hipStream_t st, const uint8_t *pyr, const Geom &g, int iniTh, int minTh, uint32_t *cell_kp,
                 int *cell_cnt, uint32_t *cand, int *lvl_off, int *overflow, int nimg, hipEvent_t ev_mid);
void launch_blur(hipStream_t st, const uint8_t *pyr, uint8_t *blur, const Geom &g, int nimg);
void launch_describe(hipStream_t st, const uint8_t *pyr, const uint8_t *blur, const Geom &g, const uint32_t *sel,
                     const int *nsel, int orientation, uint8_t *desc, float *angles, int nimg);
void launch_knn2(hipStream_t st, const uint8_t *desc, const int *counts, const int2 *pairs, int npairs, int kcap,
                 uint2 *part, float dist_thresh, float ratio, KnnRow *out, hipEvent_t ev_mid);

}  // namespace mcorb
