/*
 * mcorb.h -- C ABI of libmcorb, the MI355X (gfx950) multi-camera ORB front-end.
 *
 * Drop-in boundary for MC-SLAM's hot path.  Every entry point
 * names the reference interface it replaces (paths relative to the MC-SLAM
 * checkout).  Plain pointers and sizes only; no C++ types, no exceptions; every
 * call returns a status code (0 ok, <0 error) unless stated otherwise.
 *
 * All compute runs in hand-written HIP kernels on the selected device; the
 * library has no CPU fallback and fails with MCORB_E_NODEVICE / MCORB_E_HIP
 * when no gfx950 device can be used.  The one host-side stage is the quad-tree
 * keypoint selection (DistributeOctTree), which the reference's own design
 * makes serial and order-defining; it runs on a host worker pool between two
 * GPU phases (DESIGN.md "Selection").
 */
#ifndef MCORB_H
#define MCORB_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCORB_OK 0
#define MCORB_E_EMPTY (-1)     /* empty image: the reference's `return -1` (ORBextractor.cpp:1090-1091) */
#define MCORB_E_SIZE (-2)      /* image too small / too tall: the reference's cell or root-node
                                  arithmetic would divide by zero (ORBextractor.cpp:558,799-802) */
#define MCORB_E_CAP (-3)       /* caller buffer too small */
#define MCORB_E_ARG (-4)       /* bad argument */
#define MCORB_E_HIP (-5)       /* HIP runtime error, see mcorb_last_error() */
#define MCORB_E_NODEVICE (-6)  /* no usable gfx950 device */
#define MCORB_E_STATE (-7)     /* call out of order (e.g. match before extract) */
#define MCORB_E_OVERFLOW (-8)  /* a sparse level's candidate list does not fit the host buffer (raise cand_cap) */

#define MCORB_MAX_LEVELS 16
#define MCORB_MAX_CAMS 16      /* IntraMatch::matchIndex is array<int,5> in the reference
                                  (MultiCameraFrame.h:44); widened here for the 8-camera rig */
/* DistributeOctTree (ORBextractor.cpp:554-778) after the GPU bucketing: HOST = worker threads between two GPU phases (needs ~10 cores
 * per GPU at full rate); GPU = one wave per (image, level), the whole job is one submission and the host only reads results
 * (levels whose tree goes below the bucketing depth -- clustered corners -- fall back to the host stage for that batch).
 * AUTO = GPU, unless the environment says MCORB_SELECT=host|gpu. */
#define MCORB_SELECT_AUTO 0
#define MCORB_SELECT_HOST 1
#define MCORB_SELECT_GPU 2
#define MCORB_ORIENT_NONE 0    /* reference behaviour: angle = 0 (ORBextractor.cpp:475) */
#define MCORB_ORIENT_IC_ANGLE 1 /* the reference's dormant IC_Angle (ORBextractor.cpp:75-102) */

/* Field order is bit-compatible with cv::KeyPoint (pt.x, pt.y, size, angle,
 * response, octave, class_id), the element type of the reference's outputs. */
typedef struct mcorb_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} mcorb_keypoint;

/* Constructor arguments of ORBextractor (ORBextractor.h:49-50) plus placement. */
typedef struct mcorb_params {
    int nfeatures;        /* ORBextractor.nFeatures */
    float scale_factor;   /* ORBextractor.scaleFactor */
    int nlevels;          /* ORBextractor.nLevels, 1..MCORB_MAX_LEVELS */
    int ini_th_fast;      /* ORBextractor.iniThFAST */
    int min_th_fast;      /* ORBextractor.minThFAST */
    int orientation;      /* MCORB_ORIENT_* */
    int device_id;        /* HIP device ordinal */
    int host_threads;     /* selection workers; 0 = one per camera image, capped at hw concurrency */
    int cand_cap;         /* host-side candidate slots per image (the device list is sized for the worst case);
                             only sparse levels are copied to the host; 0 = default (max(65536, w*h/4)) */
    int selection;        /* MCORB_SELECT_*: where DistributeOctTree's list discipline runs (0 = default) */
    int gpu_jobs;         /* at most this many slots' jobs on the GPU at once; further slots wait for a turn while the finished
                             ones are post-processed on the host (0 = no limit: every slot's job goes straight to the GPU) */
    int reserved[5];
} mcorb_params;

void mcorb_default_params(mcorb_params *p);   /* 2000, 1.2, 8, 20, 7, none, dev 0 */
const char *mcorb_last_error(void);           /* thread-local message of the last failure */
const char *mcorb_version(void);
/* number of visible HIP devices whose arch is gfx950 (does not create a context) */
int mcorb_device_count(void);

/* ------------------------------------------------------------------------- */
/* Rig engine: one object per GPU, handles a batch of equally sized images    */
/* (cameras x frames).  Replaces MultiCameraFrame's extract + intra-rig match */
/* members (MultiCameraFrame.h:73-90) and the frame hand-off of               */
/* MultiCameraFrame::setData (MultiCameraFrame.cpp:95-152).                   */
/* ------------------------------------------------------------------------- */
typedef struct mcorb_rig mcorb_rig;

/* ncams cameras of width x height; up to max_frames rig frames per batch
 * (images are indexed m = frame*ncams + cam); nslots >= 1 independent buffer
 * sets so that batches can be in flight concurrently (submit/wait below). */
int mcorb_rig_create(const mcorb_params *p, int ncams, int width, int height, int max_frames,
                     int nslots, mcorb_rig **out);
void mcorb_rig_destroy(mcorb_rig *r);

/* Stage `nimg` 8-bit gray images (host, stride bytes per row) into slot's
 * level-0 planes through pinned buffers + hipMemcpyAsync.  Replaces the u8 end
 * of the hand-off (MultiCameraFrame.cpp:108-140). */
int mcorb_rig_upload_u8(mcorb_rig *r, int slot, const uint8_t *const *images, int nimg, int stride);
/* Zero-copy variant of the hand-off: the reader decodes straight into the slot's pinned staging buffer
 * (image m: *ptr, W bytes per row, H rows) and mcorb_rig_upload_staged starts the DMA of images 0..nimg-1.
 * This is the "pinned hipMemcpyAsync" staging of DatasetReader::loadNext's cv::imread target
 * (MCDataUtils/src/DatasetReader.cpp:688-719) without the intermediate clone. */
int mcorb_rig_staging(mcorb_rig *r, int slot, int m, uint8_t **ptr, int *stride);
int mcorb_rig_upload_staged(mcorb_rig *r, int slot, int nimg);
/* Same for the reference's staging format: CV_32F in [0,1], 1 or 3 (BGR)
 * channels (DatasetReader.cpp:699-712); x255, round-half-even, saturate and
 * BGR2GRAY run on the device. */
int mcorb_rig_upload_f32(mcorb_rig *r, int slot, const float *const *images, int nimg,
                         int stride_bytes, int channels);

/* extractFeaturesParallel (MultiCameraFrame.cpp:203-262) for the `nimg` images
 * already resident in `slot`: pyramid, FAST, selection, blur, descriptors.
 * lap_x0/lap_x1 = vLappingArea (ORBextractor.cpp:1153), {0,0} in the reference.
 * submit returns once the first GPU phase is enqueued; wait blocks until
 * keypoints and descriptors of the slot are complete on host and device. */
int mcorb_rig_extract_submit(mcorb_rig *r, int slot, int nimg, int lap_x0, int lap_x1);
int mcorb_rig_extract_wait(mcorb_rig *r, int slot);
/* submit + wait */
int mcorb_rig_extract(mcorb_rig *r, int slot, int nimg, int lap_x0, int lap_x1);

/* One pass of the whole hot path for `nframes` rig frames (nframes*ncams images
 * resident in the slot): extraction as above, then the intra-rig match below,
 * with a single device synchronisation at the end. */
int mcorb_rig_process_submit(mcorb_rig *r, int slot, int nframes, int lap_x0, int lap_x1,
                             float dist_thresh, float ratio);
int mcorb_rig_process_wait(mcorb_rig *r, int slot);
/* the same, synchronously on the calling thread (lowest latency for one frame at a time) */
int mcorb_rig_process(mcorb_rig *r, int slot, int nframes, int lap_x0, int lap_x1, float dist_thresh, float ratio);

/* results of image m of a slot: ORBextractor::operator() outputs
 * (ORBextractor.cpp:1085-1171): keypoints, N x 32 descriptors, monoIndex */
int mcorb_rig_num_keypoints(mcorb_rig *r, int slot, int m);
int mcorb_rig_get_features(mcorb_rig *r, int slot, int m, mcorb_keypoint *kps, uint8_t *desc, int cap,
                           int *n_out, int *mono_index_out);

/* computeIntraMatches(matches, false) (MultiCameraFrame.cpp:1100-1288) for the
 * first `nframes` rig frames of a slot: BruteForceMatch(i, j, dist_thresh,
 * ratio) for all i<j on the GPU (all-pairs Hamming k-NN, k = 2), then the
 * reference's track merge on the host. */
int mcorb_rig_match(mcorb_rig *r, int slot, int nframes, float dist_thresh, float ratio);
int mcorb_rig_match_submit(mcorb_rig *r, int slot, int nframes, float dist_thresh, float ratio);
int mcorb_rig_match_wait(mcorb_rig *r, int slot);
/* BruteForceMatch outputs of pair (cam_i < cam_j) of a frame: indices_1/2
 * (MultiCameraFrame.cpp:1070-1071); kps1/kps2 are kps[idx] of the two images */
int mcorb_rig_get_pair_matches(mcorb_rig *r, int slot, int frame, int cam_i, int cam_j,
                               uint32_t *idx1, uint32_t *idx2, int cap, int *n_out);
/* raw knnMatch(k=2) table of the pair, nq x 2 (trainIdx, distance); -1 = absent */
int mcorb_rig_get_pair_knn2(mcorb_rig *r, int slot, int frame, int cam_i, int cam_j,
                            int32_t *idx, int32_t *dist, int cap_rows, int *nq_out);
/* IntraMatch tracks of a frame: ntracks x ncams matchIndex rows, -1 = absent;
 * mergeable = cnt_mergable_matches (MultiCameraFrame.cpp:1256) */
int mcorb_rig_get_tracks(mcorb_rig *r, int slot, int frame, int32_t *tracks, int cap_tracks,
                         int *ntracks_out, int *mergeable_out);

/* computeIntraMatches(matches, old=true) (MultiCameraFrame.cpp:1123-1143,1178-1207): the same merge as
 * mcorb_rig_get_tracks with the epipolar check applied to every BruteForceMatch pair first.
 * F: one row-major 3x3 per camera pair (i<j in the order (0,1),(0,2)..), x_j^T F x_i = 0 -- the matrix
 * the reference builds from K_mats_/R_mats_/t_mats_ (:1126-1142); that cv::Mat algebra stays with the
 * caller (include/mcorb_adapter.hpp does it with cv:: when OpenCV is there).  kps_undist[c]: the camera's
 * image_kps_undist (pt and octave are read), NULL = the extracted keypoints (no distortion). */
int mcorb_rig_get_tracks_epipolar(mcorb_rig *r, int slot, int frame, const double *F, const mcorb_keypoint *const *kps_undist,
                                  int32_t *tracks, int cap_tracks, int *ntracks_out, int *mergeable_out);

/* intermediates for stage-by-stage parity tests (device -> host copies) */
int mcorb_rig_level_size(mcorb_rig *r, int level, int *w, int *h);
int mcorb_rig_get_level(mcorb_rig *r, int slot, int m, int level, uint8_t *dst, int dst_stride);
int mcorb_rig_get_blurred(mcorb_rig *r, int slot, int m, int level, uint8_t *dst, int dst_stride);
/* vToDistributeKeys of a level (ORBextractor.cpp:793-871): packed (y<<20 | x<<8 | response) */
int mcorb_rig_get_candidates(mcorb_rig *r, int slot, int m, int level, uint32_t *packed, int cap, int *n_out);

/* timing of the last completed job of a slot, microseconds between HIP events
 * recorded on the slot's stream around the launches:
 * [0] pyramid+FAST+compaction, [1] selection: host wall time (MCORB_SELECT_HOST) or k_select + k_assemble (MCORB_SELECT_GPU), [2] blur + describe(+D2H),
 * [3] k-NN + finalize, [4] pyramid launches, [5] FAST kernel alone, [6] compaction kernel (side stream),
 * [7] k-NN kernel alone, [8] blur kernel, [9] describe kernel */
int mcorb_rig_last_timing(mcorb_rig *r, int slot, float us[10]);
/* MCORB_SELECT_HOST or MCORB_SELECT_GPU: what this rig runs; jobs of a slot that fell back to the host stage so far */
int mcorb_rig_select_mode(mcorb_rig *r);
int mcorb_rig_select_fallbacks(mcorb_rig *r, int slot);
/* small batches (results through host-mapped memory): images whose early read the signal word's checksum rejected so far -- their
 * keypoint records were built after the job's end event instead (0 in every run so far) */
int mcorb_rig_early_reads_rejected(mcorb_rig *r, int slot);
/* MCORB_SELECT_GPU only: a job is the same ~20 launches and copies every time, so a slot captures it once into a HIP graph and
 * replays it with one call.  every = 0: never (launch by launch, per-kernel HIP events: mcorb_rig_last_timing is complete),
 * 1: every job (last_timing reports [0] = the whole job, the rest 0), K > 1: all but every K-th job of a slot, which runs
 * launch by launch -- a timed sample of the same pipeline.  Default: the environment's MCORB_GRAPH, else 1 for a rig with one slot
 * (one job at a time: the replay saves ~60 us of a 0.4 ms rig frame) and 0 otherwise (with several jobs in flight it measured slower). */
int mcorb_rig_set_graph(mcorb_rig *r, int every);
/* test hook for the GPU selection's sort: the permutation std::sort (libstdc++) leaves n keys in, computed by one GPU wave
 * (perm_dev) and by std::sort itself on the host (perm_std); entries compare by key only (mcorb_sortmodel.h) */
int mcorb_dev_sort_selftest(int device, const uint32_t *keys, int n, uint32_t *perm_dev, uint32_t *perm_std);

/* multi-GPU plumbing (one process per GPU; the collective itself is the
 * caller's: bench.py's throughput path uses one RCCL all-to-all with uneven splits per round, its
 * --exchange allgather path the all-gather SURVEY 8(e) describes; both over torch.distributed).
 * Descriptor block layout on the device: [sets][kcap][32] bytes. */
int mcorb_rig_kcap(mcorb_rig *r);
/* worker threads of the rig's host stage (selection, track merges): what mcorb_params.host_threads / the core budget resolved to */
int mcorb_rig_host_threads(mcorb_rig *r);
/* sizes of one image's device structures (what the byte counts of the measurements are made of):
 * out = {kcap, FAST cells, blur tiles, candidate slots per cell, candidate list capacity, quad-tree bucket entries,
 *        bytes of one pyramid block, levels} */
int mcorb_rig_info(mcorb_rig *r, int32_t out[8]);
void *mcorb_rig_desc_device_ptr(mcorb_rig *r, int slot);
void *mcorb_rig_stream(mcorb_rig *r, int slot);
/* copy the first nimg descriptor sets of a slot into caller device memory
 * (e.g. a tensor that is then all-gathered); counts_host receives the counts */
int mcorb_rig_export_descriptors(mcorb_rig *r, int slot, void *dst_dev, int32_t *counts_host, int nimg);
/* computeIntraMatches(matches,false) over an external (all-gathered) descriptor
 * block: `ntotal` sets with counts[set] descriptors each; sets[f*ncams + c] names
 * the set that holds camera c of frame f.  Results are read back with
 * mcorb_rig_get_pair_matches / _get_pair_knn2 / _get_tracks as for mcorb_rig_match. */
int mcorb_rig_match_external(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts, int ntotal,
                             const int32_t *sets, int nframes, float dist_thresh, float ratio);
/* asynchronous form: counts/sets must stay valid until mcorb_rig_match_wait(r, slot) returns */
int mcorb_rig_match_external_submit(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts, int ntotal,
                                    const int32_t *sets, int nframes, float dist_thresh, float ratio);

/* Stream-ordered forms of the two calls above, for a pipelined exchange without host synchronisation:
 * _export_descriptors_dev enqueues the copies (descriptor sets + their int32 counts, both into caller DEVICE memory)
 * on the slot's stream and makes `then_stream` (a HIP stream of this process, e.g. the stream the collective is issued
 * on; NULL = block until the copies are done) wait for them;
 * _match_external_dev_submit takes the counts from device memory and lets the slot's stream wait for everything
 * enqueued so far on `after_stream` (the collective; NULL = the caller has synchronised already).
 * The legacy NULL stream cannot be named here (its handle IS NULL): issue the collective on a stream created with
 * hipStreamCreate / torch.cuda.Stream().
 * An external block holds at most max(4096, 64 x images per slot) sets (MCORB_E_ARG beyond). */
int mcorb_rig_export_descriptors_dev(mcorb_rig *r, int slot, void *dst_dev, int32_t *counts_dev, int nimg, void *then_stream);
int mcorb_rig_match_external_dev_submit(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts_dev, int ntotal,
                                        const int32_t *sets, int nframes, float dist_thresh, float ratio, void *after_stream);

/* Pair-partitioned matching, the multi-GPU split SURVEY.md 8(e) describes for ONE rig frame across GPUs: every camera's
 * descriptors are all-gathered, BruteForceMatch of camera pair (i, j) (MultiCameraFrame.cpp:1118 loop, :1024-1086) runs on one
 * rank, and the accepted (query, train) lists return to the rank that runs computeIntraMatches' serial merge (:1167-1268).
 * _match_pairs_external*: pair_sets[2p], pair_sets[2p + 1] = query / train set of pair p inside the external block; at most
 * (images per slot) distinct sets and ncams (ncams - 1) / 2 x (frames per slot) pairs per job; no tracks are built
 * (mcorb_rig_get_tracks fails), the lists are read with mcorb_rig_get_pairlist (pair = index into pair_sets).  The _dev_submit
 * form orders the job behind `after_stream` like mcorb_rig_match_external_dev_submit; wait with mcorb_rig_match_wait.
 * LIFETIME: the submit forms return before the slot's driver thread has read `sets` / `pair_sets` (and the host `counts` of the
 * non-_dev forms): those arrays must stay valid and unchanged until mcorb_rig_match_wait has returned for that slot.
 * mcorb_host_merge_tracks: the merge itself on caller-supplied lists (pairs in (0,1), (0,2), .., (1,2), .. order, npair[p]
 * entries each, counts[c] keypoints per camera) -> tracks [n][ncams], -1 = absent; no device involved. */
int mcorb_rig_match_pairs_external(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts, int ntotal,
                                   const int32_t *pair_sets, int npairs, float dist_thresh, float ratio);
int mcorb_rig_match_pairs_external_dev_submit(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts_dev, int ntotal,
                                              const int32_t *pair_sets, int npairs, float dist_thresh, float ratio, void *after_stream);
int mcorb_rig_get_pairlist(mcorb_rig *r, int slot, int pair, uint32_t *idx1, uint32_t *idx2, int cap, int *n_out);
int mcorb_host_merge_tracks(int ncams, const int32_t *counts, const uint32_t *const *idx1, const uint32_t *const *idx2,
                            const int32_t *npair, int32_t *tracks, int cap_tracks, int *ntracks_out, int *mergeable_out);

/* Device-resident descriptor sets + matching between any two of them (findInterMatches / findMatchesMono call
 * knnMatch(k = 2) on the LF descriptors of consecutive keyframes, <= 3000 x 3000, ratio 0.7, threshold 50, FrontEnd.cpp:3114-3500).
 * A block holds nsets sets of up to kcap descriptors in HBM (kcap = mcorb_rig_kcap of the rig that will match them: create that
 * rig with nfeatures >= the largest set).  Upload a keyframe's descriptors ONCE; the previous keyframe's set stays resident, so an
 * inter-frame match moves one set over PCIe, not two (mcorb_knn2 re-uploads both).  mcorb_rig_match_sets = BFMatcher knnMatch(k=2)
 * + the (dist_thresh, ratio) filter of BruteForceMatch on explicit (query set, train set) pairs; read the accepted pairs with
 * mcorb_rig_get_pairlist and the raw k-NN table (DMatch order: lowest train index first on ties) with mcorb_rig_get_pairknn2. */
typedef struct mcorb_descblock mcorb_descblock;
int mcorb_descblock_create(int device, int nsets, int kcap, mcorb_descblock **out);
void mcorb_descblock_destroy(mcorb_descblock *b);
int mcorb_descblock_upload(mcorb_descblock *b, int set, const uint8_t *desc, int n);
void *mcorb_descblock_desc_ptr(mcorb_descblock *b);
int32_t *mcorb_descblock_counts_dev(mcorb_descblock *b);
int mcorb_rig_match_sets(mcorb_rig *r, int slot, mcorb_descblock *b, const int32_t *pair_sets, int npairs, float dist_thresh, float ratio);
int mcorb_rig_get_pairknn2(mcorb_rig *r, int slot, int pair, int32_t *idx, int32_t *dist, int cap_rows, int *nq_out);

/* ------------------------------------------------------------------------- */
/* Single-camera extractor: ORBextractor (ORBextractor.h:43-116)              */
/* ------------------------------------------------------------------------- */
typedef struct mcorb_extractor mcorb_t;

/* ORBextractor::ORBextractor (ORBextractor.cpp:408-468); buffers are sized for
 * images up to max_width x max_height (geometry is rebuilt when the size changes) */
int mcorb_create(const mcorb_params *p, int max_width, int max_height, mcorb_t **out);
void mcorb_destroy(mcorb_t *e);
/* ORBextractor::operator() (ORBextractor.cpp:1085-1171); the mask argument of
 * the reference is ignored there and absent here.  Returns MCORB_OK and
 * *mono_index_out = the reference's return value, or MCORB_E_EMPTY for the
 * reference's -1. */
int mcorb_extract(mcorb_t *e, const uint8_t *gray, int w, int h, int stride_bytes,
                  int lap_x0, int lap_x1, mcorb_keypoint *kps, uint8_t *desc, int cap,
                  int *n_out, int *mono_index_out);
/* same, fed with the reference's CV_32F [0,1] frame (setData, MultiCameraFrame.cpp:108-116) */
int mcorb_extract_f32(mcorb_t *e, const float *img01, int w, int h, int stride_bytes, int channels,
                      int lap_x0, int lap_x1, mcorb_keypoint *kps, uint8_t *desc, int cap,
                      int *n_out, int *mono_index_out);
/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (ORBextractor.h:61-81) + mnFeaturesPerLevel */
int mcorb_get_tables(const mcorb_params *p, float *scale, float *inv_scale, float *sigma2,
                     float *inv_sigma2, int *features_per_level);
/* mvImagePyramid[level] interior of the last call (ORBextractor.h:89) */
int mcorb_get_pyramid_level(mcorb_t *e, int level, uint8_t *dst, int dst_stride, int *w, int *h);

/* ------------------------------------------------------------------------- */
/* Descriptor distance / matchers                                             */
/* ------------------------------------------------------------------------- */
/* ORBextractor::DescriptorDistance (ORBextractor.cpp:1202-1218); host, 0..256 */
int mcorb_hamming256(const uint8_t a[32], const uint8_t b[32]);
/* MultiCameraFrame::computeRepresentativeDesc (MultiCameraFrame.cpp:530-567): index of the descriptor
 * (n x 32 bytes, n <= 64: one per camera of a track) with the least median distance to the rest. Host. */
int mcorb_representative_desc(const uint8_t *descs, int n);
/* DescriptorMatcher("BruteForce-Hamming")->knnMatch(q, t, out, 2)
 * (MultiCameraFrame.cpp:1053-1055; FrontEnd.cpp findInterMatches): host
 * descriptor arrays in, nq x 2 (trainIdx, distance) out, -1 = absent. */
int mcorb_knn2(mcorb_t *e, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist);
/* BruteForceMatch (MultiCameraFrame.cpp:1024-1086): knn2 + ratio/threshold filter */
int mcorb_match_ratio(mcorb_t *e, const uint8_t *q, int nq, const uint8_t *t, int nt,
                      float dist_thresh, float ratio, uint32_t *idx1, uint32_t *idx2, int cap, int *n_out);

/* ------------------------------------------------------------------------- */
/* DBoW2 vocabulary: transform(features, BowVector, FeatureVector, levelsup)  */
/* (MultiCameraFrame.cpp:257, FrontEnd.cpp:525,929).  The    */
/* tree descent runs on the GPU; the std::map-ordered weight accumulation and */
/* normalisation on the host, in feature order.                               */
/* ------------------------------------------------------------------------- */
typedef struct mcorb_vocab mcorb_vocab;
/* nodes 1..nnodes in file order (node 0 is the root): parent id, leaf flag, 32-byte descriptor, weight.
 * scoring: 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT;
 * weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (DBoW2's enums; ORBvoc.txt is "10 6 0 0"). */
int mcorb_vocab_create(int k, int L, int scoring, int weighting, const int32_t *parent, const uint8_t *is_leaf,
                       const uint8_t *desc, const double *weight, int nnodes, int device, mcorb_vocab **out);
/* TemplatedVocabulary::loadFromTextFile (FrontEnd.h:137-138) */
int mcorb_vocab_load_text(const char *path, int device, mcorb_vocab **out);
void mcorb_vocab_destroy(mcorb_vocab *v);
int mcorb_vocab_info(const mcorb_vocab *v, int *k, int *L, int *nnodes, int *nwords);
/* transform n descriptors (host, n x 32).  BowVector: (word id, value) ascending by id;
 * FeatureVector: node ids ascending, fv_offsets[i]..fv_offsets[i+1] index fv_feats (feature indices). */
int mcorb_vocab_transform(mcorb_vocab *v, const uint8_t *desc, int n, int levelsup, uint32_t *bow_ids, double *bow_vals,
                          int bow_cap, int *nbow, uint32_t *fv_nodes, int32_t *fv_offsets, int fv_cap, int *nfv,
                          int32_t *fv_feats, int feat_cap);
/* same for image m of a rig slot, reading the descriptors where extraction left them in HBM */
int mcorb_rig_transform_image(mcorb_rig *r, int slot, int m, mcorb_vocab *v, int levelsup, uint32_t *bow_ids,
                              double *bow_vals, int bow_cap, int *nbow, uint32_t *fv_nodes, int32_t *fv_offsets,
                              int fv_cap, int *nfv, int32_t *fv_feats, int feat_cap);

/* computeIntraMatches(matches, words_), the BoW-guided live variant (MultiCameraFrame.cpp:586-943,
 * FrontEnd.cpp:1009) for one extracted rig frame of a slot: vocabulary descent and the per-node
 * best / second-best distance table on the GPU, the reference's serial track bookkeeping on the host.
 * tracks: ntracks x ncams matchIndex rows (-1 absent); n_rays per track; words: node id pushed to
 * words_ for every accepted feature.  max_neighbor_ratio = ORBextractor::max_neighbor_ratio (0.85). */
/* transform() of images [img0, img0 + nimg) of a slot at once (one descent launch; the order-defined folds of the
 * images run on the worker pool); read each image's vectors with mcorb_rig_get_transform (layout as
 * mcorb_vocab_transform's outputs) */
int mcorb_rig_transform_images(mcorb_rig *r, int slot, int img0, int nimg, mcorb_vocab *v, int levelsup);
int mcorb_rig_get_transform(mcorb_rig *r, int slot, int m, uint32_t *bow_ids, double *bow_vals, int bow_cap, int *nbow,
                            uint32_t *fv_nodes, int32_t *fv_offsets, int fv_cap, int *nfv, int32_t *fv_feats, int feat_cap);
int mcorb_rig_match_bow(mcorb_rig *r, int slot, int frame, mcorb_vocab *v, int levelsup, double max_neighbor_ratio,
                        int32_t *tracks, int32_t *n_rays, int cap_tracks, int *ntracks_out, uint32_t *words,
                        int cap_words, int *nwords_out);
/* The same for frames [frame0, frame0 + nframes) of a slot at once (one descent launch, one table launch, the serial
 * bookkeeping of the frames on the worker pool); results are kept per frame and read with mcorb_rig_get_bow_tracks.
 * y_undist (may be NULL): y_undist[m], m = frame * ncams + cam, points at the undistorted rows of image m's keypoints --
 * image_kps_undist[cam][k].pt.y, which the reference's |dy| < 50 gate reads (MultiCameraFrame.cpp:708-716); NULL = the raw
 * rows (RECTIFY, or zero distortion: image_kps_undist == image_kps, MultiCameraFrame.cpp:302-307). */
int mcorb_rig_match_bow_frames(mcorb_rig *r, int slot, int frame0, int nframes, mcorb_vocab *v, int levelsup,
                               double max_neighbor_ratio, const float *const *y_undist);
int mcorb_rig_get_bow_tracks(mcorb_rig *r, int slot, int frame, int32_t *tracks, int32_t *n_rays, int cap_tracks,
                             int *ntracks_out, uint32_t *words, int cap_words, int *nwords_out);

/* ------------------------------------------------------------------------- */
/* FrontEnd::obtainLfFeatures (MCSlam/src/FrontEnd.cpp:213-593): the consumer  */
/* of the IntraMatch tracks.  Host code.                     */
/* ------------------------------------------------------------------------- */
/* one camera of camconfig_: K_mats_[i] (row-major 3x3) and build_Rt(R_mats_[i], t_mats_[i]) (row-major 3x4) */
typedef struct mcorb_camera {
    double K[9];
    double Rt[12];
} mcorb_camera;
/* one entry of currentFrame->intraMatches as obtainLfFeatures leaves it: IntraMatch{matchIndex, uv_ref, mono, n_rays,
 * matchDesc, point3D} (MultiCameraFrame.h:42-57); point3d is meaningful when mono == 0 */
typedef struct mcorb_lf_feature {
    int32_t match_index[MCORB_MAX_CAMS];
    float uv_ref[2];
    int32_t mono, n_rays;
    double point3d[3];
    uint8_t desc[32];
} mcorb_lf_feature;
/* tracks: ntracks x ncams ints (matches_map, -1 = absent) of `frame` of the slot, words (may be NULL): words_ per track;
 * cams: ncams entries; seg_masks (may be NULL, or NULL per camera = all zero): per camera a float image with `seg_stride`
 * floats per row, a view is dropped where the mask at the RAW keypoint is >= 0.7 (:262-270); kps_undist (may be NULL):
 * image_kps_undist per camera (uv_ref and the response of mono features are read from it, :397-408, :497-505);
 * total_feats: 3000 in the reference (:430).  out receives the accepted multi-view tracks in track order (triangulated with
 * cv::sfm::triangulatePoints' DLT, kept when 0.5 < z < 40, :309) followed by the mono features in argsorte(responses, false)
 * order (:514-521); *intramatch_size_out / *mono_size_out = lf_frame->intramatch_size / mono_size; words_fil (may be NULL)
 * = the std::set filled at :344.  lIds is all -1 (one per output entry) and lfBoW is mcorb_vocab_transform of the output
 * descriptors (:525).  Parity: every integer / ordering result exact; point3d and uv_ref of triangulated tracks to 1e-9
 * relative (the SVD behind the reference's triangulation is un-vendored: unpinned). */
int mcorb_rig_obtain_lf_features(mcorb_rig *r, int slot, int frame, const int32_t *tracks, int ntracks, const uint32_t *words,
                                 const mcorb_camera *cams, const float *const *seg_masks, int seg_stride,
                                 const mcorb_keypoint *const *kps_undist, int total_feats, mcorb_lf_feature *out, int cap,
                                 int *n_out, int *intramatch_size_out, int *mono_size_out, uint32_t *words_fil, int cap_words,
                                 int *nwords_fil_out);
/* The same for all frames [frame0, frame0 + nframes) of a slot in one call, one worker-pool task per frame (FrontEnd.cpp:1024
 * calls obtainLfFeatures once per frame; a batch holds many).  tracks / words: the frames' arrays back to back -- ntracks[f]
 * tracks (ncams ints each) and, if words != NULL, as many words per frame; seg_masks / kps_undist: nframes * ncams pointers
 * (index f * ncams + cam) or NULL; out: nframes blocks of `cap` entries; words_fil (may be NULL): nframes blocks of cap_words;
 * n_out / intramatch_size_out / mono_size_out / nwords_fil_out: nframes entries each.  Returns the first failing frame's status. */
int mcorb_rig_obtain_lf_features_frames(mcorb_rig *r, int slot, int frame0, int nframes, const int32_t *tracks, const int32_t *ntracks,
                                        const uint32_t *words, const mcorb_camera *cams, const float *const *seg_masks, int seg_stride,
                                        const mcorb_keypoint *const *kps_undist, int total_feats, mcorb_lf_feature *out, int cap,
                                        int *n_out, int *intramatch_size_out, int *mono_size_out, uint32_t *words_fil, int cap_words,
                                        int *nwords_fil_out);

/* ------------------------------------------------------------------------- */
/* Host stages exposed for the CPU test-suite (no device needed)              */
/* ------------------------------------------------------------------------- */
/* The engine's quad-tree selection, DistributeOctTree's equivalent (ORBextractor.cpp:554-778), run
 * end to end on the host: the candidate bucketing that k_compact performs on the device is
 * restated on the CPU, then the host tree logic runs on it.  packed = (y<<20 | x<<8 | response),
 * x/y relative to minBorder, in vToDistributeKeys order (the "first maximum wins" tie of the final
 * pick follows the input order); wCell/hCell = cell grid of the detection loop, from which the same
 * order (cell row, cell col, y, x) is recovered for nodes deeper than the bucketing (pass 0,0 when
 * the candidates are in plain raster order).  out_idx receives indices into `packed` in result
 * order.  Returns the count, MCORB_E_SIZE, or MCORB_E_CAP. */
int mcorb_host_select(const uint32_t *packed, int n, int minX, int maxX, int minY, int maxY,
                      int nfeatures_level, int wCell, int hCell, int32_t *out_idx, int cap);
/* The engine's cv::resize coefficient table for one axis: per destination index
 * (s0, s1, c0, c1) as int32 quadruples (x axis: clamped per HResizeLinear; y axis:
 * row indices clipped, fraction kept). */
int mcorb_host_resize_axis(int ssize, int dsize, int is_x, int32_t *quads);
/* level geometry the engine derives for a w x h image: per level
 * {w, h, nCols, nRows, wCell, hCell}; returns MCORB_OK or MCORB_E_SIZE */
int mcorb_host_geometry(const mcorb_params *p, int w, int h, int32_t *six_per_level);
/* the N-view DLT triangulation of mcorb_rig_obtain_lf_features alone (cv::sfm::triangulatePoints for one point):
 * x = nv normalised image points (x0, y0, x1, y1, ..), P = nv row-major 3x4 [R|t], 2 <= nv <= MCORB_MAX_CAMS */
int mcorb_host_triangulate(const double *x, const double *P, int nv, double X[3]);

/* ------------------------------------------------------------------------- */
/* Synthetic input; host utility, see csrc/mcorb_synth.c       */
/* ------------------------------------------------------------------------- */
int mcorb_synth_rig_frame(uint32_t frame, int ncams, int cam, int w, int h, uint8_t *out, int stride);

#ifdef __cplusplus
}
#endif
#endif
