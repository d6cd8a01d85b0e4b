// mcorb_engine.h -- host orchestration of the gfx950 ORB front-end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcorb.h"
#include "mcorb_common.h"
#include "mcorb_kernels.h"
#include "mcorb_select.h"

namespace mcorb {

void set_error(const std::string &msg);
const char *get_error();

// ORBextractor constructor tables (ORBextractor.cpp:408-468)
struct Tables {
    int nlevels = 0;
    float scale[kMaxLevels], inv_scale[kMaxLevels], sigma2[kMaxLevels], inv_sigma2[kMaxLevels];
    int quota[kMaxLevels];
    int scaled_patch[kMaxLevels];
    int umax[16];
};
int compute_tables(const mcorb_params &p, Tables &t);

// cv::resize's per-axis tables, folded into ResizeTap entries
void build_resize_axis(int ssize, int dsize, bool is_x, std::vector<ResizeTap> &out, int pad_to);

// level / cell / tile geometry for W x H; returns MCORB_OK or MCORB_E_SIZE
// lut (optional): receives the path-code tables of all levels (LevelGeom::lutx / luty index into it)
int build_geometry(const mcorb_params &p, const Tables &t, int W, int H, Geom &g, std::vector<ResizeTap> &taps,
                   std::vector<uint16_t> *lut = nullptr);

class WorkerPool {
public:
    explicit WorkerPool(int nthreads);
    ~WorkerPool();
    // runs fn(task, worker_index) for task in [0, n); returns when all are done
    // caller_widx >= 0: the calling thread takes tasks as well, using that worker index
    void parallel_for(int n, const std::function<void(int, int)> &fn, int caller_widx = -1);
    int size() const { return (int)threads_.size(); }

private:
    struct Batch {
        const std::function<void(int, int)> *fn;
        std::atomic<int> next{0};
        int n = 0;
        std::atomic<int> done{0};
        std::atomic<int> refs{0};
        std::mutex m;
        std::condition_variable cv;
    };
    void run(int widx);
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Batch *> queue_;
    std::atomic<int> pending_{0};   // batches in queue_ (lets idle workers spin without the lock)
    bool stop_ = false;
};

struct Job {
    enum Kind { NONE, EXTRACT, MATCH, PROCESS } kind = NONE;
    int nimg = 0, nframes = 0, lap0 = 0, lap1 = 0;
    float dist_thresh = 75.f, ratio = 0.85f;
    // external descriptor block (RCCL path): ext_total sets of [kcap][32] bytes on the device,
    // ext_counts[set] descriptors each; ext_sets[f*ncams + c] = set holding camera c of frame f
    const void *ext_desc = nullptr;
    const int32_t *ext_counts = nullptr;
    int ext_total = 0;
    const int32_t *ext_sets = nullptr;
    // device-resident form (no host synchronisation between the collective and the match): the counts live in device
    // memory, and the slot's stream first waits for everything enqueued so far on `after_stream` (the collective)
    const int32_t *ext_counts_dev = nullptr;
    hipStream_t after_stream = nullptr;
    // explicit pair list instead of "all camera pairs of nframes frames" (the pair-partitioned multi-GPU path: pair (i, j) of a
    // frame is matched on one rank): ext_npairs pairs of (query set, train set) indices into the external block; no
    // track merge on this rank (nframes = 0), the per-pair lists are read with mcorb_rig_get_pairlist
    const int32_t *ext_pairs = nullptr;
    int ext_npairs = 0;
};

class Rig;

// computeIntraMatches' track merge (MultiCameraFrame.cpp:1167-1268) over the BruteForceMatch lists of one frame's camera pairs
// in (0,1), (0,2), .., (1,2), .. order: counts[c] keypoints per camera, idx1[p] / idx2[p] the accepted (query, train) indices of
// pair p (np[p] of them).  gate != nullptr adds the old=true epipolar check.  Shared by the engine and mcorb_host_merge_tracks.
struct EpipolarGate;
void merge_pair_lists(int ncams, const int *counts, const uint32_t *const *idx1, const uint32_t *const *idx2, const int *np,
                      const EpipolarGate *gate, std::vector<int32_t> &tr, int &mergeable_out);

// computeIntraMatches(matches, words_) of one frame (mcorb_rig_match_bow_frames): tracks (ncams ints each), their n_rays, words_
struct BowFrameOut {
    std::vector<int32_t> tracks, n_rays;
    std::vector<uint32_t> words;
};

// transform() of one image (mcorb_rig_transform_images): BowVector as sorted (word id, value) lists, FeatureVector as
// node ids + offsets into the feature list
struct BowImageOut {
    std::vector<uint32_t> bow_ids, fv_nodes;
    std::vector<double> bow_vals;
    std::vector<int32_t> fv_offsets, fv_feats;
};

struct Slot {
    Rig *rig = nullptr;
    int index = 0;
    bool shared_st = false;
    bool blur_valid = false;   // d_blur holds the blurred planes of the images in d_pyr
    hipStream_t st = nullptr, st_copy = nullptr, st_dma = nullptr;   // compute; PCIe-bound compaction kernel; D2H copies only
    hipEvent_t ev_x = nullptr;   // cross-stream hand-offs with the caller's streams (export / external match)
    hipEvent_t ev_c = nullptr;   // k_compact finished (the table DMA follows it on the side stream)
    hipEvent_t ev_e = nullptr;   // k_expand finished (k_knn2 follows)
    hipEvent_t ev[12] = {};  // 0 start, 1 pyramid done, 2 FAST done, 3 compact done, 4 blur done, 5/6 describe(+D2H), 7 knn2 start, 8 knn2 done, 9 finalize done
    // device
    uint8_t *d_pyr = nullptr, *d_blur = nullptr, *d_desc = nullptr;
    uint32_t *d_cellkp = nullptr, *d_sorted = nullptr;
    int *d_cellcnt = nullptr;
    float *d_angles = nullptr, *d_f32 = nullptr;
    uint2 *d_part = nullptr;
    uint8_t *d_exp = nullptr;        // descriptors of the sets being matched, expanded to +-64 int8 in MFMA fragment order (k_expand)
    int *d_lcounts = nullptr;        // their clamped counts, local set order
    size_t f32_bytes = 0;
    // host, device-mapped (kernels write/read these directly over PCIe)
    uint32_t *h_cand = nullptr;
    int *h_overflow = nullptr;
    // k_compact's per-image table blocks (level offsets, shipped flags, bucket starts, bucket winners; layout in
    // mcorb_common.h): written to d_tbl by the kernel, brought to the pinned h_tbl by one DMA per batch
    int *d_tbl = nullptr, *h_tbl = nullptr;
    int tbl_ints_per_image = 0;
    const int *tbl(int img) const { return h_tbl + (size_t)img * tbl_ints_per_image; }
    volatile uint32_t touch_sink[16] = {};
    KnnRow *d_knn = nullptr;         // k-NN rows per pair (device; read back only by mcorb_rig_get_pair_knn2)
    uint32_t *h_mlist = nullptr;     // per pair: accepted (query << 16 | train), query order (k_knn2_finalize)
    int *h_mcount = nullptr;
    // control block: one pinned host buffer + one device mirror, copied with a single
    // hipMemcpyAsync: [extcounts ext_cap ints][nsel][setmap][pairs][sel]
    uint8_t *h_ctrl = nullptr, *d_ctrl = nullptr;
    size_t ctrl_pairs_end = 0, ctrl_bytes = 0;
    int *h_extcounts = nullptr, *h_nsel = nullptr, *h_setmap = nullptr;
    int2 *h_pairs = nullptr;
    uint32_t *h_sel = nullptr;
    int *d_extcounts = nullptr, *d_nsel = nullptr, *d_setmap = nullptr;
    int2 *d_pairs = nullptr;
    uint32_t *d_sel = nullptr;
    // GPU selection (MCORB_SELECT_GPU): k_select's per-(image, level) lists, and the result block k_assemble fills for the host
    // ([16 ints of flags][mono M ints][responses M x kcap bytes]; sel / nsel are written into the control block)
    uint32_t *d_selval = nullptr;
    int *d_selcnt = nullptr;
    uint8_t *d_res = nullptr, *h_res = nullptr;
    size_t res_bytes = 0, res_mono_off = 0, res_resp_off = 0, ctrl_nsel_off = 0;
    unsigned long long *h_sig = nullptr;      // small GPU-selected jobs: per image, set by k_assemble behind its host-mapped results
    bool capturing = false;    // enqueue_gpu_job is being captured into the slot's graph
    bool gpu_small = false;    // the running GPU-selected job is a small batch: results through host-mapped memory, no copies but the flags
    int stale_reads = 0;       // small batches: images whose early read did not match the signal word's checksum (redone after the end event)
    int fallbacks = 0;         // jobs of this slot the host stage had to redo (a tree below the bucketing depth)
    hipEvent_t ev_s = nullptr; // k_select + k_assemble finished
    hipEvent_t ev_g = nullptr; // in front of a replayed job graph
    struct GraphKey { int nimg, match, nframes, lap0, lap1; float dist_thresh, ratio; };
    hipGraphExec_t graph_exec = nullptr;   // the captured job (run_gpu_selected), valid for graph_key
    GraphKey graph_key = {};
    unsigned job_counter = 0;
    bool graph_timing = false;             // the last job ran as a graph: timing[] holds the whole job only
    // host, pinned
    uint8_t *h_stage = nullptr, *h_desc = nullptr;
    float *h_angles = nullptr;
    // results
    std::vector<std::vector<mcorb_keypoint>> kps;   // per image
    std::vector<int> mono;
    std::vector<std::vector<uint32_t>> sel_val;     // per (image, level): retained candidates (packed), result order
    int npairs_done = 0, nframes_done = 0, nimg_done = 0;
    int nsets_local = 0;   // descriptor sets the last match expanded (frames x cameras, or the distinct sets of an explicit pair list)
    std::vector<std::vector<uint32_t>> m_idx1, m_idx2;   // per pair
    std::vector<std::vector<int32_t>> tracks;            // per frame, ncams ints per track
    std::vector<int> mergeable;
    // cached BoW results of the images the slot holds NOW: a new extraction clears the flags (run_extract_phaseA), the
    // getters refuse frames / images that were not matched / transformed since
    std::vector<BowFrameOut> bow;   // per frame
    std::vector<uint8_t> bow_ok;    // per frame: bow[f] belongs to the current extraction
    std::vector<BowImageOut> bowvec;   // per image
    std::vector<uint8_t> bowvec_ok;
    void invalidate_bow()
    {
        std::fill(bow_ok.begin(), bow_ok.end(), (uint8_t)0);
        std::fill(bowvec_ok.begin(), bowvec_ok.end(), (uint8_t)0);
    }
    float timing[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int> match_sets, match_counts;   // per (frame, cam) of the last match: set index, descriptor count
    bool match_external = false;
    bool small_job = false;   // the running job is a small batch: tables, control block and descriptors go through host-mapped memory, no copies
    // driver thread
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    Job job;
    bool busy = false, quit = false, inline_job = false;
    int status = MCORB_OK;
    std::string err;
};

// inputs of the old=true epipolar check: F per camera pair (i<j, row-major 3x3, x_j^T F x_i = 0), the
// per-camera keypoints the reference reads (image_kps_undist) and the extractor's sigma2 table
struct EpipolarGate {
    const double *F;
    const mcorb_keypoint *const *kps;
    const float *sigma2;
};

class Rig {
public:
    Rig() = default;
    ~Rig();
    int init(const mcorb_params &p, int ncams, int W, int H, int max_frames, int nslots);

    int upload_u8(int slot, const uint8_t *const *images, int nimg, int stride);
    int upload_staged(int slot, int nimg);
    int upload_f32(int slot, const float *const *images, int nimg, int stride_bytes, int channels);
    int submit(int slot, const Job &job);
    int run_sync(int slot, const Job &job);   // submit + wait on the calling thread
    int wait(int slot);

    mcorb_params params;
    Tables tab;
    Geom geom;
    int ncams = 0, W = 0, H = 0, max_frames = 0, max_images = 0, npp = 0 /* pairs per frame */;
    int ext_cap = 0;   // descriptor sets an external block may hold (mcorb_rig_match_external*)
    int device = 0;
    ResizeTap *d_taps = nullptr;
    uint16_t *d_lut = nullptr;             // path-code tables of all levels (k_compact)
    uint32_t *d_fasttab = nullptr;         // k_fast_cells' per-cell records (fast_cell_table)
    int fast_cell_off = 0;                  // dword offset of the records inside d_fasttab
    SelectParams selp[kMaxLevels];         // per-level DistributeOctTree constants + bucketing depth
    int resize_win[2 * kMaxLevels] = {};   // per level: LDS window pitch, rows (see launch_pyramid)
    std::vector<Slot *> slots;
    WorkerPool *pool = nullptr;
    int pool_threads = 0;
    // admission to the GPU (mcorb_params.gpu_jobs): with more slots than jobs the GPU runs well side by side, the extra slots are
    // the ones whose results the host is post-processing -- the GPU does not wait for the host, and is not oversubscribed either
    int gpu_job_limit = 0;     // 0 = none
    int gpu_jobs_running = 0;
    std::mutex gpu_jobs_m;
    std::condition_variable gpu_jobs_cv;
    void gpu_job_begin();
    void gpu_job_end();
    bool upload_pipelined = true;   // upload_u8 of a small batch: staging copy and DMA overlapped image by image
    std::atomic<int> graph_every{0};   // GPU-selected jobs replayed from a captured HIP graph: 0 never, 1 always, K all but every K-th (mcorb_rig_set_graph)
    int select_deep_cap = 4096; // k_select: largest bucket it scans node by node below the bucketing depth (MCORB_SELECT_DEEP_CAP at rig creation)
    bool gpu_select = false;   // DistributeOctTree's list discipline runs in k_select (MCORB_SELECT_GPU); else on the worker pool
    int wait_mode = 0;         // how a thread waits for a HIP event: 0 spin (hipEventSynchronize), 1 interrupt-driven, 2 poll + short sleeps
    hipError_t wait_event(hipEvent_t ev) const;
    std::vector<SelectScratch *> scratch;   // one per worker

    void merge_tracks(Slot &s, int f, const EpipolarGate *gate, std::vector<int32_t> &tr, int &mergeable_out) const;
    int max_pairs() const { return std::max(1, npp * max_frames); }

private:
    bool copy_kernel = false;  // D2H of tables / descriptors by k_copy_to_host instead of hipMemcpyAsync (see Rig::init)
    bool blur_planes = false;  // k_blur runs with every job (orientation mode / MCORB_BLUR_PLANES); otherwise blur is fused into k_describe_fused
    void driver(Slot *s);
    int execute(Slot &s, const Job &j);
    int run_extract_phaseA(Slot &s, const Job &j);
    int run_select_and_describe(Slot &s, const Job &j, bool then_match);
    int run_gpu_selected(Slot &s, const Job &j, bool then_match);   // the whole job as one submission (gpu_select)
    int enqueue_gpu_job(Slot &s, const Job &j, bool then_match);
    int prepare_match(Slot &s, const Job &j);
    int enqueue_match(Slot &s, const Job &j, bool ctrl_on_device);
    int finish_match(Slot &s, const Job &j);
};

}  // namespace mcorb

// the C handle of include/mcorb.h
struct mcorb_rig {
    mcorb::Rig rig;
};
