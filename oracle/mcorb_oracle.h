/*
 * mcorb_oracle.h -- CPU restatement of MC-SLAM's multi-camera ORB front-end.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and bench.py's cpu_baseline leg may load it, and only
 * as the checker / timed CPU baseline.  libmcorb (the HIP product) never links
 * or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests
 * or fixtures for this path, and its arithmetic lives in
 * un-vendored OpenCV (4.x generic C++ paths are what is restated here), which
 * is not installed in this container.  The restatement is pinned only by
 * hand-derivable known-answer tests of each primitive (tests/test_oracle_*.py)
 * and by structural invariants taken from the reference source.
 *
 * Every function cites the reference file:line (relative to the MC-SLAM
 * checkout) or the OpenCV routine whose published algorithm it follows.
 */
#ifndef MCORB_ORACLE_H
#define MCORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16
#define ORC_MAX_CAMS 16

/* bit-compatible with cv::KeyPoint's field order */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct orc_extractor orc_extractor;

/* ORBextractor::ORBextractor, MCSlam/src/ORBextractor.cpp:408-468.
 * orientation: 0 = reference behaviour (angle = 0, ORBextractor.cpp:475),
 *              1 = IC_Angle enabled (the commented-out call, :75-102). */
orc_extractor *orc_create(int nfeatures, float scale_factor, int nlevels,
                          int ini_th_fast, int min_th_fast, int orientation);
void orc_destroy(orc_extractor *e);

/* scale tables as the reference's getters expose them (ORBextractor.h:61-81)
 * plus per-level quotas mnFeaturesPerLevel (:433-444) and umax (:450-467). */
void orc_get_tables(const orc_extractor *e, float *scale, float *inv_scale,
                    float *sigma2, float *inv_sigma2, int *quota, int *umax16);

/* level size for a w x h input, ORBextractor.cpp:1177-1178 */
void orc_level_size(const orc_extractor *e, int level, int w, int h, int *lw, int *lh);

/* ORBextractor::operator(), ORBextractor.cpp:1085-1171.
 * Returns monoIndex (>=0), -1 for an empty image (reference :1090-1091),
 * -2 if the image is too small / too tall for the reference's cell and
 * root-node arithmetic to be defined (the reference would divide by zero),
 * -3 if cap is too small.  n_out receives the keypoint count. */
int orc_extract(orc_extractor *e, const uint8_t *gray, int w, int h, int stride,
                int lap_x0, int lap_x1, orc_keypoint *kps, uint8_t *desc, int cap,
                int *n_out);

/* intermediates of the last orc_extract call (for stage-by-stage parity) */
/* un-bordered level plane (interior ROI of mvImagePyramid[level]) */
const uint8_t *orc_last_level(const orc_extractor *e, int level, int *w, int *h, int *stride);
/* bordered plane (w+38)x(h+38), ORBextractor.cpp:1179-1194 */
const uint8_t *orc_last_level_bordered(const orc_extractor *e, int level, int *w, int *h, int *stride);
/* blurred level (workingMat after GaussianBlur, :1132-1133); NULL if level had no keypoints */
const uint8_t *orc_last_blurred(const orc_extractor *e, int level, int *w, int *h, int *stride);
/* vToDistributeKeys of a level (:793-871): x,y relative to minBorder, response */
int orc_last_candidates(const orc_extractor *e, int level, float *x, float *y, float *resp, int cap);
/* allKeypoints[level] after DistributeOctTree + fix-up (:876-889), level coords */
int orc_last_level_keypoints(const orc_extractor *e, int level, orc_keypoint *kps, int cap);

/* ---- third-party primitives (OpenCV 4.x generic paths) ---- */
int orc_cv_round_f(float v);          /* cvRound(float): round-half-even */
int orc_cv_round_d(double v);         /* cvRound(double) */
int orc_cv_floor_f(float v);
int orc_cv_ceil_f(float v);
/* cv::resize(..., INTER_LINEAR) for CV_8UC1 (A.3) */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride);
/* the coefficient tables resize() builds: xofs/ialpha (2 per dx), yofs/ibeta */
void orc_resize_tables(int ssize, int dsize, int *ofs, int16_t *coef2);
/* cv::copyMakeBorder(..., BORDER_REFLECT_101) (A.5) */
void orc_copy_make_border_101(const uint8_t *src, int w, int h, int sstride,
                              uint8_t *dst, int dstride, int border);
/* cv::FAST(roi, kps, threshold, true/false), TYPE_9_16 (A.2).  Returns count;
 * xs/ys are ROI coordinates in raster order, score = response. */
int orc_fast_9_16(const uint8_t *img, int stride, int cols, int rows, int threshold,
                  int nonmax, int *xs, int *ys, int *score, int cap);
/* cornerScore<16> for the pixel at (x,y) */
int orc_fast_corner_score(const uint8_t *img, int stride, int x, int y, int threshold);
/* cv::GaussianBlur(src,dst,Size(7,7),2,2,BORDER_REFLECT_101), 8UC1 fixed point (A.4) */
void orc_gaussian_blur_7x7_s2(const uint8_t *src, int w, int h, int sstride,
                              uint8_t *dst, int dstride);
void orc_gaussian_kernel_q8(int taps7[7]);   /* {18,34,48,56,48,34,18} derived, not hard-coded */
/* cv::fastAtan2 (A.8), degrees */
float orc_fast_atan2(float y, float x);
/* frame hand-off: multiply(f32,255) -> convertTo(CV_8U) -> BGR2GRAY if 3 channels
 * (MCSlam/src/MultiCameraFrame.cpp:108-116, A.6) */
void orc_stage_f32(const float *img, int w, int h, int stride_bytes, int channels,
                   uint8_t *gray, int gstride);

/* ---- selection ---- */
/* ORBextractor::DistributeOctTree, ORBextractor.cpp:554-778 (+ DivideNode :479-535,
 * compareNodes :537-552).  Inputs are the candidate x,y,response in
 * vToDistributeKeys order; out_idx receives the index of each retained key in
 * result (list) order.  Returns the count, or -2 when nIni < 1. */
int orc_distribute_octree(const float *x, const float *y, const float *resp, int n,
                          int minX, int maxX, int minY, int maxY, int N, int *out_idx, int cap);

/* ---- descriptors / matching ---- */
/* ORBextractor::DescriptorDistance, ORBextractor.cpp:1202-1218 (SWAR popcount) */
int orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32]);
/* ORBextractor::getMatches_distRatio, :1228-1290; returns #matches, BookK accumulates */
int orc_get_matches_dist_ratio(const uint8_t *A, const uint32_t *iA, int nA,
                               const uint8_t *B, const uint32_t *iB, int nB,
                               double max_neighbor_ratio,
                               uint32_t *mA, uint32_t *mB, int *bookK);
/* argsorte (MCSlam/include/MCSlam/utils.h:21-30): indices sorted by data with std::sort, ascending or descending */
void orc_argsorte(const float *data, int n, int ascen, int *indices_out);
/* BFMatcher(NORM_HAMMING).knnMatch(q,t,out,2) (A.7): idx/dist are nq x 2, absent = -1 */
void orc_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist);
/* MultiCameraFrame::BruteForceMatch filter, MultiCameraFrame.cpp:1060-1078 */
int orc_bruteforce_match(const uint8_t *q, int nq, const uint8_t *t, int nt,
                         float dist_thresh, float neigh_ratio,
                         uint32_t *idx1, uint32_t *idx2, int cap);
/* MultiCameraFrame::computeIntraMatches(matches, old=false), :1100-1288 track merge.
 * desc[c] / n[c]: per-camera descriptors.  tracks: ntracks x ncams ints (-1 absent).
 * Returns number of tracks; mergeable receives cnt_mergable_matches. */
int orc_intra_matches(const uint8_t *const *desc, const int *n, int ncams,
                      float dist_thresh, float neigh_ratio,
                      int32_t *tracks, int cap_tracks, int *mergeable);
/* the same with old=true: epipolar check of every pair (:1178-1207) against F (one row-major 3x3 per
 * camera pair, the matrix of :1142), image_kps_undist and GetScaleSigmaSquares() */
int orc_intra_matches_epi(const uint8_t *const *desc, const int *n, int ncams,
                          float dist_thresh, float neigh_ratio, const double *F,
                          const orc_keypoint *const *kps, const float *sigma2,
                          int32_t *tracks, int cap_tracks, int *mergeable);
/* MultiCameraFrame::computeRepresentativeDesc, :530-567 (least median distance) */
int orc_representative_desc(const uint8_t *descs, int n);

/* DBoW2 TemplatedVocabulary<FORB>::transform(features, BowVector, FeatureVector, levelsup) (A.9).
 * Vocabulary given as loadFromTextFile reads it: nodes 1..nnodes in file order (parent, leaf flag,
 * 32-byte descriptor, weight).  Outputs as std::map iteration order (ascending ids). */
int orc_bow_transform(int k, int L, int scoring, int weighting, const int32_t *parent, const uint8_t *is_leaf,
                      const uint8_t *ndesc, const double *nweight, int nnodes, const uint8_t *feats, int nf,
                      int levelsup, uint32_t *bow_ids, double *bow_vals, int *nbow, uint32_t *fv_nodes,
                      int32_t *fv_offsets, int *nfv, int32_t *fv_feats);

/* MultiCameraFrame::computeIntraMatches(matches, words_), the BoW-guided live variant
 * (MultiCameraFrame.cpp:586-943).  Per camera: descriptors, keypoint y (image_kps_undist[].pt.y) and the
 * FeatureVector as arrays.  tracks: ntracks x ncams; n_rays per track; words_: node id per accepted feature. */
int orc_intra_matches_bow(const uint8_t *const *desc, const float *const *kp_y, int ncams,
                          const uint32_t *const *fv_nodes, const int32_t *const *fv_offsets,
                          const int32_t *const *fv_feats, const int *nfv, double max_neighbor_ratio,
                          int32_t *tracks, int32_t *n_rays_out, int cap_tracks, uint32_t *words_out, int cap_words,
                          int *nwords_out);

#ifdef __cplusplus
}
#endif
#endif
