// mcorb_engine.cpp -- host orchestration: tables, geometry, buffers, slot drivers.
//
// Data flow of one batch (nimg equally sized images, all resident in HBM):
//   phase A (GPU)  pyramid L1..L7 -> FAST score + cell NMS -> candidate compaction
//                  (bucket tables to the host by DMA; whole-level blur only in the IC-angle mode)
//   selection      host worker pool, one task per image                 [mcorb_select.cpp]
//   phase B (GPU)  blur around the selected keypoints + BRIEF descriptors (one kernel) -> D2H
//   match (GPU)    all-pairs Hamming k-NN (k=2) + ratio/threshold flags -> host-mapped
//   merge          per-frame IntraMatch track merge on the host
// Each slot owns a stream, a complete buffer set and a driver thread, so several
// batches can be in flight and the host stage of one overlaps the GPU phases of others.
#include "mcorb_engine.h"

#include <alloca.h>
#include <math.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>
#include <sys/prctl.h>
#include <algorithm>
#include <new>
#include <cmath>
#include <chrono>

namespace mcorb {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_));                 \
            return MCORB_E_HIP;                                                        \
        }                                                                              \
    } while (0)

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------
// ORBextractor::ORBextractor (ORBextractor.cpp:408-468).  scaleFactor is a
// double member initialised from a float argument (ORBextractor.h:103).
// ---------------------------------------------------------------------------
int compute_tables(const mcorb_params &p, Tables &t)
{
    if (p.nlevels < 1 || p.nlevels > kMaxLevels || p.nfeatures < 1 || !(p.scale_factor > 1.0f)) {
        set_error("bad extractor parameters");
        return MCORB_E_ARG;
    }
    const int L = p.nlevels;
    const double sf = (double)p.scale_factor;
    t.nlevels = L;
    t.scale[0] = 1.0f;
    t.sigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) {
        t.scale[i] = (float)((double)t.scale[i - 1] * sf);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < L; i++) {
        t.inv_scale[i] = 1.0f / t.scale[i];
        t.inv_sigma2[i] = 1.0f / t.sigma2[i];
        t.scaled_patch[i] = (int)(31 * t.scale[i]);   // PATCH_SIZE*mvScaleFactor[level] (:879)
    }
    const float factor = (float)(1.0 / sf);
    float desired = (float)p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        t.quota[l] = cv_round_f(desired);
        sum += t.quota[l];
        desired *= factor;
    }
    t.quota[L - 1] = std::max(p.nfeatures - sum, 0);
    // umax (:450-467)
    const int HP = 15;
    int v, v0;
    const int vmax = cv_floor_f(HP * sqrtf(2.f) / 2 + 1);
    const int vmin = cv_ceil_f(HP * sqrtf(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v < 16; v++) t.umax[v] = 0;
    for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    return MCORB_OK;
}

// cv::resize's table loop for one axis: x clamps (sx, fx), y keeps
// the fraction and clips the row indices at use.
void build_resize_axis(int ssize, int dsize, bool is_x, std::vector<ResizeTap> &out, int pad_to)
{
    const double scale = (double)ssize / dsize;
    const size_t first = out.size();
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor_f(f);
        f -= s;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        int c0 = cv_round_f((1.f - f) * 2048.f), c1 = cv_round_f(f * 2048.f);
        c0 = std::min(std::max(c0, -32768), 32767);
        c1 = std::min(std::max(c1, -32768), 32767);
        int s0 = std::min(std::max(s, 0), ssize - 1);
        int s1 = std::min(std::max(s + 1, 0), ssize - 1);
        if (is_x && s + 1 >= ssize) { c0 = 2048; c1 = 0; }   // HResizeLinear tail: S[sx]*ONE
        ResizeTap t;
        t.s0 = (uint16_t)s0; t.s1 = (uint16_t)s1; t.c0 = (int16_t)c0; t.c1 = (int16_t)c1;
        out.push_back(t);
    }
    // padding: copies of the last tap, so that a kernel reading whole groups of `pad_to` never sees an out-of-range column
    const ResizeTap last = out.size() > first ? out.back() : ResizeTap{0, 0, 0, 0};
    while ((out.size() - first) % pad_to) out.push_back(last);
}

int build_geometry(const mcorb_params &p, const Tables &t, int W, int H, Geom &g, std::vector<ResizeTap> &taps,
                   std::vector<uint16_t> *lut)
{
    memset(&g, 0, sizeof(g));
    taps.clear();
    if (lut) lut->clear();
    g.nlevels = t.nlevels;
    size_t off = 0;
    int cells = 0, tiles = 0, cellCap = 4, buckets = 0;
    for (int l = 0; l < t.nlevels; l++) {
        LevelGeom &L = g.lv[l];
        const float sc = t.inv_scale[l];
        L.w = cv_round_f((float)W * sc);   // ORBextractor.cpp:1177-1178
        L.h = cv_round_f((float)H * sc);
        L.maxBorderX = L.w - kEdge + 3;
        L.maxBorderY = L.h - kEdge + 3;
        const float width = (float)(L.maxBorderX - kMinBorder);
        const float height = (float)(L.maxBorderY - kMinBorder);
        L.nCols = (int)(width / (float)kCellW);
        L.nRows = (int)(height / (float)kCellW);
        if (L.nCols < 1 || L.nRows < 1) {
            set_error("image too small for the reference's 35-px cell grid at level " + std::to_string(l));
            return MCORB_E_SIZE;
        }
        L.wCell = (int)ceilf(width / L.nCols);
        L.hCell = (int)ceilf(height / L.nRows);
        const SelectParams sp = make_select_params(kMinBorder, L.maxBorderX, kMinBorder, L.maxBorderY, t.quota[l], 1, 1);
        if (sp.nIni < 1) {
            set_error("image too tall: DistributeOctTree would have no root node");
            return MCORB_E_SIZE;
        }
        if (sp.nIni > 16) { set_error("image too wide (more than 16 root nodes)"); return MCORB_E_SIZE; }
        L.nIni = sp.nIni;
        L.hX = sp.hX;
        L.depth = sp.depth;
        L.nBuckets = sp.nIni << (2 * sp.depth);
        L.bucket0 = buckets;
        L.quota = t.quota[l];
        buckets += L.nBuckets + 1;
        if (lut) {   // path-code tables of this level (k_compact): code(x, y) = lut[L.lutx + x] | lut[L.luty + y]
            const int W0 = L.maxBorderX - kMinBorder, H0 = L.maxBorderY - kMinBorder;
            L.lutx = (uint32_t)lut->size();
            L.luty = L.lutx + (uint32_t)W0;
            lut->resize(lut->size() + (size_t)W0 + H0);
            path_code_tables(W0, H0, L.nIni, L.hX, L.depth, lut->data() + L.lutx, lut->data() + L.luty);
            while (lut->size() & 7) lut->push_back(0);
        }
        if (L.w > 4096 || L.h > 4096) { set_error("image larger than 4096 px"); return MCORB_E_SIZE; }
        L.pitch = (int)align_up((size_t)L.w, 64);
        L.off = (uint32_t)off;
        off += align_up((size_t)L.pitch * align_up((size_t)L.h, kBlurTileRows), 256);   // whole 16x8 tiles (blurred planes)
        L.cell0 = cells;
        cells += L.nCols * L.nRows;
        L.tilesX = (L.w + kBlurTW - 1) / kBlurTW;
        L.tilesY = (L.h + kBlurTH - 1) / kBlurTH;
        L.tile0 = tiles;
        tiles += L.tilesX * L.tilesY;
        cellCap = std::max(cellCap, ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2));
        if (l > 0) {
            L.xtab = (uint32_t)taps.size();
            build_resize_axis(g.lv[l - 1].w, L.w, true, taps, 4);
            L.ytab = (uint32_t)taps.size();
            build_resize_axis(g.lv[l - 1].h, L.h, false, taps, 4);
        }
    }
    g.cells = cells;
    g.tiles = tiles;
    g.bucketTotal = buckets;
    g.cellCap = (int)align_up((size_t)cellCap, 4);
    g.imgBytes = (uint32_t)(off + 256);
    g.kcap = (int)align_up((size_t)p.nfeatures + 4 * t.nlevels + 48, 64);
    if (g.kcap > 65535) { set_error("nfeatures too large (k-NN index is 16 bits)"); return MCORB_E_ARG; }
    // default: one candidate slot per 4 level-0 pixels (measured: ~1 per 28 px on the synthetic rig frames)
    // device list: worst case (every cell full: one corner per 2x2 px survives the 3x3 NMS at most), so FAST itself can
    // never overflow; host copy: one slot per 4 level-0 pixels by default (~1 per 28 px measured on the synthetic rig
    // frames) -- it is only written when the quad-tree may go below the bucketing, i.e. for sparse levels
    g.candCap = (int)align_up((size_t)g.cells * g.cellCap, 4096);
    g.hostCandCap = p.cand_cap > 0 ? p.cand_cap : (int)align_up(std::max((size_t)65536, (size_t)W * H / 4), 4096);
    if (g.hostCandCap > g.candCap) g.hostCandCap = g.candCap;
    if (g.lv[0].nCols * g.lv[0].nRows * g.cellCap > kPickOrderMask) { set_error("image too large (pick order is 23 bits)"); return MCORB_E_SIZE; }
    return MCORB_OK;
}

// ---------------------------------------------------------------------------
// optional host-side profile (MCORB_HOST_PROF=1): wall and thread-CPU time of the selection tasks
// ---------------------------------------------------------------------------
namespace HostProf {
static const bool on = getenv("MCORB_HOST_PROF") != nullptr;
static std::atomic<long long> wall[4], cpu[4], cnt[4];
static inline long long now(clockid_t c) { timespec t; clock_gettime(c, &t); return t.tv_sec * 1000000000LL + t.tv_nsec; }
struct Scope {
    int k; long long w0 = 0, c0 = 0;
    explicit Scope(int k_) : k(k_) { if (on) { w0 = now(CLOCK_MONOTONIC); c0 = now(CLOCK_THREAD_CPUTIME_ID); } }
    ~Scope() { if (on) { wall[k] += now(CLOCK_MONOTONIC) - w0; cpu[k] += now(CLOCK_THREAD_CPUTIME_ID) - c0; cnt[k]++; } }
};
static void report()
{
    if (!on) return;
    const char *names[4] = {"select task (per image)", "  select_octree x levels", "finish_match (per job)", "prepare_match (per job)"};
    for (int k = 0; k < 4; k++)
        if (cnt[k].load())
            fprintf(stderr, "[mcorb host prof] %-26s n=%lld wall %.1f us cpu %.1f us\n", names[k], cnt[k].load(),
                    wall[k].load() / 1e3 / cnt[k].load(), cpu[k].load() / 1e3 / cnt[k].load());
}
}  // namespace HostProf

// ---------------------------------------------------------------------------
// worker pool
// ---------------------------------------------------------------------------
WorkerPool::WorkerPool(int nthreads)
{
    for (int i = 0; i < nthreads; i++) threads_.emplace_back([this, i] { run(i); });
}
WorkerPool::~WorkerPool()
{
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
}
void WorkerPool::run(int widx)
{
    for (;;) {
        Batch *b = nullptr;
        {
            std::unique_lock<std::mutex> lk(m_);
            // spin briefly before sleeping: batches arrive every few hundred microseconds when the
            // pipeline is busy, and a futex wake-up costs more than the selection of one image
            if (queue_.empty() && !stop_) {
                lk.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                while (pending_.load(std::memory_order_acquire) == 0 &&
                       std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(150)) {
                    __builtin_ia32_pause();
                }
                lk.lock();
            }
            cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
            if (stop_ && queue_.empty()) return;
            b = queue_.front();
            if (b->next.load() >= b->n) {   // exhausted: drop it from the queue
                queue_.erase(queue_.begin());
                pending_.fetch_sub(1, std::memory_order_acq_rel);
                continue;
            }
            b->refs.fetch_add(1, std::memory_order_acq_rel);   // the batch lives on its submitter's stack
        }
        for (;;) {
            const int t = b->next.fetch_add(1);
            if (t >= b->n) break;
            (*b->fn)(t, widx);
            if (b->done.fetch_add(1) + 1 == b->n) {
                std::lock_guard<std::mutex> lk(b->m);
                b->cv.notify_all();
            }
        }
        b->refs.fetch_sub(1, std::memory_order_acq_rel);
    }
}
void WorkerPool::parallel_for(int n, const std::function<void(int, int)> &fn, int caller_widx)
{
    if (n <= 0) return;
    Batch b;
    b.fn = &fn;
    b.n = n;
    {
        std::lock_guard<std::mutex> lk(m_);
        queue_.push_back(&b);
        pending_.fetch_add(1, std::memory_order_acq_rel);
    }
    cv_.notify_all();
    // the submitting thread works on its own batch too (with its own scratch index): a one-frame batch does not
    // have to wait for a sleeping worker to wake up
    if (caller_widx >= 0) {
        for (;;) {
            const int t = b.next.fetch_add(1);
            if (t >= b.n) break;
            fn(t, caller_widx);
            b.done.fetch_add(1);
        }
    }
    {
        std::unique_lock<std::mutex> lk(b.m);
        b.cv.wait(lk, [&b] { return b.done.load() >= b.n; });
    }
    {
        std::lock_guard<std::mutex> lk(m_);   // after this no new worker can pick the batch up
        auto it = std::find(queue_.begin(), queue_.end(), &b);
        if (it != queue_.end()) {
            queue_.erase(it);
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    while (b.refs.load(std::memory_order_acquire) != 0) std::this_thread::yield();   // workers still leaving the task loop
}

// ---------------------------------------------------------------------------
// Rig
// ---------------------------------------------------------------------------
// Elapsed milliseconds between two events of a finished job; 0 when either was not recorded on a stream (a job that ran from its
// captured graph holds them as graph nodes).  A failed query must not stay behind as the thread's "last error".
static inline void ev_elapsed(float *ms, hipEvent_t a, hipEvent_t b)
{
    if (hipEventElapsedTime(ms, a, b) != hipSuccess) { *ms = 0.f; (void)hipGetLastError(); }
}

hipError_t Rig::wait_event(hipEvent_t ev) const
{
    if (wait_mode != 2) return hipEventSynchronize(ev);   // spins, or sleeps on the interrupt (event flag)
    for (int i = 0;; i++) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (i < 4) { std::this_thread::yield(); continue; }   // a wait that is almost over
        timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
}

// Cores this process can actually use: hardware threads, cut down to the scheduler affinity mask and to the cgroup CPU
// quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), whichever is smallest.
static int usable_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1) {}
        else quota = -1;
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1LL, quota / period));
    // one process per GPU on a multi-GPU node: the ranks of the node share those cores (torch.distributed.run exports
    // LOCAL_WORLD_SIZE; MCORB_LOCAL_RANKS says the same for other launchers)
    const char *lr = getenv("MCORB_LOCAL_RANKS") ? getenv("MCORB_LOCAL_RANKS") : getenv("LOCAL_WORLD_SIZE");
    const int ranks = lr ? atoi(lr) : 1;
    if (ranks > 1) n = std::max(2, n / ranks);
    return n;
}

template <typename T>
static int dev_alloc(T **p, size_t n)
{
    HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
    return MCORB_OK;
}
template <typename T>
static int host_alloc(T **p, size_t n)
{
    HIPCHK(hipHostMalloc((void **)p, n * sizeof(T), hipHostMallocMapped | hipHostMallocPortable));
    memset(*p, 0, n * sizeof(T));
    return MCORB_OK;
}
#define TRY(x)                       \
    do {                             \
        int r_ = (x);                \
        if (r_ != MCORB_OK) return r_; \
    } while (0)

int Rig::init(const mcorb_params &p, int ncams_, int W_, int H_, int max_frames_, int nslots)
{
    if (ncams_ < 1 || ncams_ > MCORB_MAX_CAMS || max_frames_ < 1 || nslots < 1 || W_ < 1 || H_ < 1) {
        set_error("bad rig arguments");
        return MCORB_E_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p.device_id < 0 || p.device_id >= ndev) {
        set_error("no usable HIP device (libmcorb has no CPU path)");
        return MCORB_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, p.device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("device is ") + prop.gcnArchName + ", libmcorb is built for gfx950 only");
        return MCORB_E_NODEVICE;
    }
    params = p;
    ncams = ncams_; W = W_; H = H_; max_frames = max_frames_;
    max_images = ncams * max_frames;
    npp = ncams * (ncams - 1) / 2;
    device = p.device_id;
    HIPCHK(hipSetDevice(device));
    TRY(compute_tables(p, tab));
    std::vector<ResizeTap> taps;
    std::vector<uint16_t> lut;
    TRY(build_geometry(p, tab, W, H, geom, taps, &lut));
    for (int l = 0; l < geom.nlevels; l++)
        selp[l] = make_select_params(kMinBorder, geom.lv[l].maxBorderX, kMinBorder, geom.lv[l].maxBorderY, tab.quota[l],
                                     geom.lv[l].wCell, geom.lv[l].hCell);
    if (taps.empty()) taps.push_back(ResizeTap{0, 0, 0, 0});
    // LDS source window of one resize workgroup (256 x kResizeTileH outputs): widest column span / tallest row span per level
    for (int l = 1; l < geom.nlevels; l++) {
        const LevelGeom &D = geom.lv[l];
        int maxc = 16, maxr = 2;
        for (int bx0 = 0; bx0 < D.w; bx0 += 256) {
            const int bx1 = std::min(bx0 + 255, D.w - 1);
            const int a = taps[D.xtab + bx0].s0 & ~15, b = taps[D.xtab + bx1].s1;
            maxc = std::max(maxc, ((b - a) / 16 + 1) * 16);
        }
        for (int by0 = 0; by0 < D.h; by0 += kResizeTileH) {
            const int by1 = std::min(by0 + kResizeTileH - 1, D.h - 1);
            maxr = std::max(maxr, (int)taps[D.ytab + by1].s1 - (int)taps[D.ytab + by0].s0 + 1);
        }
        resize_win[2 * l] = maxc;
        resize_win[2 * l + 1] = maxr;
        if ((size_t)maxc * maxr > 60000) { set_error("scale factor too large for the resize window"); return MCORB_E_ARG; }
    }
    TRY(dev_alloc(&d_taps, taps.size()));
    HIPCHK(hipMemcpy(d_taps, taps.data(), taps.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
    TRY(dev_alloc(&d_lut, lut.size() + 8));
    HIPCHK(hipMemcpy(d_lut, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> ft;
        fast_cell_off = fast_cell_table(geom, ft);   // (16-byte aligned: the records are read 16 bytes at a time)
        TRY(dev_alloc(&d_fasttab, ft.size()));
        HIPCHK(hipMemcpy(d_fasttab, ft.data(), ft.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    HIPCHK(upload_umax(tab.umax));

    // Host threads: the selection workers, one driver per slot (spinning on its events unless the rig sleeps, below) and
    // the submitting thread all draw on the cores this process may use -- the scheduler affinity AND the cgroup CPU quota
    // (a GPU box gives 16 cores per GPU: with 16 workers + 6 spinning drivers the quota ran out in 10 of 40 periods and the
    // whole process was stalled for 5-15 ms each time, 15 % of the throughput).  Default: what is left after the drivers
    // and the caller, at most one worker per image of a batch and at most 16.
    const int cores = usable_cores();
    // How the slot drivers wait for the GPU.  hipEventSynchronize spins: one core per slot, ~60 % of it spent doing
    // nothing at six slots -- cores the selection needs (at 31 k frames/s the selection alone keeps 8.7 cores busy and
    // the GPU box grants 16).  HIP's interrupt-driven wait (hipEventBlockingSync) measured no cheaper in CPU time.
    // Default with several slots: poll hipEventQuery with 20 us sleeps in between (a few % of a core; the added latency
    // is hidden behind the other slots).  One slot = latency mode: spin.  MCORB_SYNC=spin|block|poll overrides.
    const char *sync_env0 = getenv("MCORB_SYNC");
    wait_mode = nslots > 1 ? 2 : 0;
    if (sync_env0) wait_mode = !strcmp(sync_env0, "block") ? 1 : (!strcmp(sync_env0, "poll") ? 2 : 0);
    const bool crowded = wait_mode == 1;   // events created with hipEventBlockingSync
    int nthreads = p.host_threads > 0 ? p.host_threads : (wait_mode == 0 ? cores - nslots - 2 : cores - 4);
    if (p.host_threads <= 0) nthreads = std::max(2, std::min(nthreads, std::min(max_images, 16)));
    if (getenv("MCORB_HOST_THREADS")) nthreads = atoi(getenv("MCORB_HOST_THREADS"));
    nthreads = std::max(1, std::min(nthreads, 64));
    blur_planes = p.orientation != 0 || getenv("MCORB_BLUR_PLANES") != nullptr;
    // Device -> host copies of the tables and the descriptors: the runtime's hipMemcpyAsync (a blit kernel on this stack) or
    // k_copy_to_host (a 24-workgroup kernel of ours).  Measured (profiles/r03_copy_kernel_ab.txt): with six slots in flight the
    // runtime's copy gives 4-6 % more frames/s although k_expand / k_knn2 run stretched beside it; with one slot (latency mode)
    // ours keeps k_expand at 12 us instead of 150 and the frame a few % shorter.  MCORB_COPY_KERNEL=0|1 overrides.
    copy_kernel = getenv("MCORB_COPY_KERNEL") ? atoi(getenv("MCORB_COPY_KERNEL")) != 0 : nslots == 1;
    {   // where DistributeOctTree's list discipline runs (include/mcorb.h, MCORB_SELECT_*)
        const char *e = getenv("MCORB_SELECT");
        int mode = p.selection;
        if (mode == MCORB_SELECT_AUTO && e) mode = !strcmp(e, "host") ? MCORB_SELECT_HOST : (!strcmp(e, "gpu") ? MCORB_SELECT_GPU : MCORB_SELECT_AUTO);
        if (mode != MCORB_SELECT_AUTO && mode != MCORB_SELECT_HOST && mode != MCORB_SELECT_GPU) { set_error("mcorb_params.selection: unknown mode"); return MCORB_E_ARG; }
        gpu_select = mode != MCORB_SELECT_HOST && select_fits(geom);
        // (test knob, read once here -- never from the slot drivers' threads: a getenv per launch raced with a profiler's setenv)
        if (getenv("MCORB_SELECT_DEEP_CAP")) select_deep_cap = std::max(1, atoi(getenv("MCORB_SELECT_DEEP_CAP")));
        // HIP graphs: a single-slot rig (one job at a time, how MC-SLAM calls) replays its job from a captured graph -- 0.35 -> 0.29 ms
        // per rig frame; with several jobs in flight the replay measured 3 - 5 % SLOWER than launch by launch (profiles/r04_overlap.txt)
        gpu_job_limit = getenv("MCORB_GPU_JOBS") ? atoi(getenv("MCORB_GPU_JOBS")) : p.gpu_jobs;
        if (gpu_job_limit < 0 || gpu_job_limit >= nslots) gpu_job_limit = 0;
        upload_pipelined = !(getenv("MCORB_UPLOAD_PIPE") && atoi(getenv("MCORB_UPLOAD_PIPE")) == 0);   // (A/B knob)
        graph_every = !gpu_select ? 0 : getenv("MCORB_GRAPH") ? std::max(0, atoi(getenv("MCORB_GRAPH"))) : (nslots == 1 ? 1 : 0);   // (mcorb_rig_select_mode reports what the rig really runs)
    }
    pool = new WorkerPool(nthreads);
    pool_threads = nthreads;
    for (int i = 0; i < nthreads + nslots; i++) scratch.push_back(new SelectScratch);   // workers, then one per slot's submitting thread

    const int npairs_max = std::max(1, npp * max_frames);
    // an external block usually holds the sets of all slots of a rank (all-to-all) or of all ranks (all-gather)
    ext_cap = (int)align_up(std::max((size_t)4096, (size_t)64 * max_images), 64);
    for (int si = 0; si < nslots; si++) {
        Slot *s = new Slot;
        slots.push_back(s);
        s->rig = this;
        s->index = si;
        if (si > 0 && getenv("MCORB_SHARED_STREAM")) { s->st = slots[0]->st; s->shared_st = true; }
        else if (getenv("MCORB_CU_SPLIT") && atoi(getenv("MCORB_CU_SPLIT")) >= 2) {
            // experiment: the slots' compute streams are confined to disjoint groups of XCDs (CU-masked
            // streams), so that jobs of different groups run side by side instead of taking turns on the whole chip
            const int parts = std::min(8, atoi(getenv("MCORB_CU_SPLIT")));
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> mask(words, 0u);
            const int part = si % parts;
            // CU index c belongs to XCD c % 8 in the mask's numbering on this part (round-robin): give a group whole XCDs
            for (int c = 0; c < ncu; c++)
                if ((c % 8) * parts / 8 == part) mask[c / 32] |= 1u << (c % 32);
            HIPCHK(hipExtStreamCreateWithCUMask(&s->st, (uint32_t)words, mask.data()));
        }
        else HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&s->st_copy, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&s->st_dma, hipStreamNonBlocking));
        // events a driver thread waits on (3 compaction, 10 compute stream, 11 DMA stream).  HIP spins by default, which
        // measured 2-4 % faster than interrupt-driven waits at 6 slots; with many slots the spinning drivers would take
        // the cores the selection workers need, so those rigs sleep instead.  MCORB_SYNC=block|spin overrides.
        const char *sync_env = getenv("MCORB_SYNC");
        const bool blocking = sync_env ? !strcmp(sync_env, "block") : crowded;
        HIPCHK(hipEventCreateWithFlags(&s->ev_x, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_c, hipEventDefault));
        HIPCHK(hipEventCreateWithFlags(&s->ev_e, hipEventDefault));
        for (int e = 0; e < 12; e++) {
            const bool waited = e == 3 || e == 10 || e == 11;
            HIPCHK(hipEventCreateWithFlags(&s->ev[e], waited && blocking ? hipEventBlockingSync : hipEventDefault));
        }
        const size_t M = (size_t)max_images;
        TRY(dev_alloc(&s->d_pyr, M * geom.imgBytes));
        HIPCHK(hipMemset(s->d_pyr, 0, M * geom.imgBytes));
        if (blur_planes) {   // whole blurred planes: only the plane-based descriptor paths write them with every job;
            TRY(dev_alloc(&s->d_blur, M * geom.imgBytes));   // mcorb_rig_get_blurred allocates on first use otherwise
            HIPCHK(hipMemset(s->d_blur, 0, M * geom.imgBytes));
        }
        TRY(dev_alloc(&s->d_cellkp, M * geom.cells * geom.cellCap));
        TRY(dev_alloc(&s->d_cellcnt, M * geom.cells));
        TRY(dev_alloc(&s->d_sorted, M * geom.candCap));
        s->tbl_ints_per_image = tbl_ints(geom.bucketTotal);
        TRY(dev_alloc(&s->d_tbl, M * s->tbl_ints_per_image));
        TRY(host_alloc(&s->h_tbl, M * s->tbl_ints_per_image));
        TRY(dev_alloc(&s->d_desc, M * geom.kcap * 32));
        HIPCHK(hipMemset(s->d_desc, 0, M * geom.kcap * 32));
        TRY(dev_alloc(&s->d_angles, M * geom.kcap));
        TRY(dev_alloc(&s->d_part, knn_part_entries(npairs_max, geom.kcap)));
        TRY(dev_alloc(&s->d_exp, M * geom.kcap * (size_t)kKnnExpandBytes));
        TRY(dev_alloc(&s->d_lcounts, M));
        TRY(host_alloc(&s->h_cand, M * geom.hostCandCap));
        TRY(host_alloc(&s->h_overflow, 16));
        TRY(dev_alloc(&s->d_knn, (size_t)npairs_max * geom.kcap));
        TRY(host_alloc(&s->h_mlist, (size_t)npairs_max * knn_mlist_stride(geom.kcap)));
        TRY(host_alloc(&s->h_mcount, (size_t)npairs_max * knn_qblocks(geom.kcap)));
        {
            const size_t o_nsel = (size_t)ext_cap * sizeof(int);
            const size_t o_setmap = align_up(o_nsel + M * sizeof(int), 64);
            const size_t o_pairs = align_up(o_setmap + M * sizeof(int), 64);
            const size_t o_sel = align_up(o_pairs + (size_t)npairs_max * sizeof(int2), 64);
            s->ctrl_pairs_end = o_sel;
            s->ctrl_nsel_off = o_nsel;
            s->ctrl_bytes = o_sel + M * geom.kcap * sizeof(uint32_t);
            TRY(host_alloc(&s->h_ctrl, s->ctrl_bytes));
            TRY(dev_alloc(&s->d_ctrl, s->ctrl_bytes));
            HIPCHK(hipMemset(s->d_ctrl, 0, s->ctrl_bytes));
            s->h_extcounts = (int *)s->h_ctrl;            s->d_extcounts = (int *)s->d_ctrl;
            s->h_nsel = (int *)(s->h_ctrl + o_nsel);      s->d_nsel = (int *)(s->d_ctrl + o_nsel);
            s->h_setmap = (int *)(s->h_ctrl + o_setmap);  s->d_setmap = (int *)(s->d_ctrl + o_setmap);
            s->h_pairs = (int2 *)(s->h_ctrl + o_pairs);   s->d_pairs = (int2 *)(s->d_ctrl + o_pairs);
            s->h_sel = (uint32_t *)(s->h_ctrl + o_sel);   s->d_sel = (uint32_t *)(s->d_ctrl + o_sel);
        }
        if (gpu_select) {
            TRY(dev_alloc(&s->d_selval, M * geom.nlevels * (size_t)select_cap(geom)));
            TRY(dev_alloc(&s->d_selcnt, M * geom.nlevels));
            s->res_mono_off = 16 * sizeof(int);
            s->res_resp_off = align_up(s->res_mono_off + M * sizeof(int), 64);
            s->res_bytes = align_up(s->res_resp_off + M * geom.kcap, 64);
            TRY(dev_alloc(&s->d_res, s->res_bytes));
            TRY(host_alloc(&s->h_res, s->res_bytes));
            HIPCHK(hipMemset(s->d_res, 0, s->res_bytes));
            TRY(host_alloc(&s->h_sig, M));
            HIPCHK(hipEventCreateWithFlags(&s->ev_s, hipEventDefault));
            HIPCHK(hipEventCreateWithFlags(&s->ev_g, hipEventDefault));
        }
        TRY(host_alloc(&s->h_stage, M * (size_t)W * H));
        TRY(host_alloc(&s->h_desc, M * geom.kcap * 32));
        TRY(host_alloc(&s->h_angles, M * geom.kcap));
        s->kps.resize(M);
        s->mono.assign(M, 0);
        s->sel_val.resize(M * geom.nlevels);
        s->m_idx1.resize(npairs_max);
        s->m_idx2.resize(npairs_max);
        s->tracks.resize(max_frames);
        s->mergeable.assign(max_frames, 0);
        s->th = std::thread([this, s] { driver(s); });
    }
    // The hipMemset calls above run on the null stream and return before the fill kernels have executed; the slots'
    // streams are non-blocking, i.e. NOT ordered against the null stream, so without this an early upload could be
    // zeroed again by a fill that was still queued (seen as an occasional empty first frame).
    HIPCHK(hipDeviceSynchronize());
    return MCORB_OK;
}

Rig::~Rig()
{
    HostProf::report();
    for (Slot *s : slots) {
        if (s->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(s->m);
                s->quit = true;
            }
            s->cv.notify_all();
            s->th.join();
        }
        (void)hipSetDevice(device);
        if (s->st) (void)hipStreamSynchronize(s->st);
        if (s->st_copy) (void)hipStreamSynchronize(s->st_copy);
        if (s->st_dma) (void)hipStreamSynchronize(s->st_dma);
        (void)hipFree(s->d_pyr); (void)hipFree(s->d_blur); (void)hipFree(s->d_desc); (void)hipFree(s->d_cellkp);
        (void)hipFree(s->d_cellcnt); (void)hipFree(s->d_sorted); (void)hipFree(s->d_tbl); (void)hipHostFree(s->h_tbl); (void)hipFree(s->d_angles); (void)hipFree(s->d_part); (void)hipFree(s->d_exp); (void)hipFree(s->d_lcounts); (void)hipFree(s->d_f32);
        (void)hipHostFree(s->h_cand); (void)hipHostFree(s->h_overflow);
        (void)hipFree(s->d_knn); (void)hipHostFree(s->h_mlist); (void)hipHostFree(s->h_mcount); (void)hipHostFree(s->h_ctrl); (void)hipFree(s->d_ctrl);
        (void)hipHostFree(s->h_stage);
        (void)hipHostFree(s->h_desc); (void)hipHostFree(s->h_angles);
        (void)hipFree(s->d_selval); (void)hipFree(s->d_selcnt); (void)hipFree(s->d_res); (void)hipHostFree(s->h_res); (void)hipHostFree(s->h_sig);
        if (s->ev_s) (void)hipEventDestroy(s->ev_s);
        if (s->ev_g) (void)hipEventDestroy(s->ev_g);
        if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
        for (auto &e : s->ev) if (e) (void)hipEventDestroy(e);
        if (s->ev_x) (void)hipEventDestroy(s->ev_x);
        if (s->ev_c) (void)hipEventDestroy(s->ev_c);
        if (s->ev_e) (void)hipEventDestroy(s->ev_e);
        if (s->st && !s->shared_st) (void)hipStreamDestroy(s->st);
        if (s->st_copy) (void)hipStreamDestroy(s->st_copy);
        if (s->st_dma) (void)hipStreamDestroy(s->st_dma);
        delete s;
    }
    slots.clear();
    delete pool;
    for (auto *sc : scratch) delete sc;
    if (d_taps) (void)hipFree(d_taps);
    if (d_lut) (void)hipFree(d_lut);
    if (d_fasttab) (void)hipFree(d_fasttab);
}

// An upload into a slot whose job is still running would overwrite the staging buffer and level 0 between the job's
// GPU phases (the slot's stream is idle while the host selects): refuse it like every other call on a busy slot.
constexpr int kSmallBatch = 8;   // images: at most two 4-camera rig frames

static bool slot_busy(Slot &s)
{
    std::lock_guard<std::mutex> lk(s.m);
    if (s.busy) set_error("slot busy: wait for the submitted job before uploading into its slot");
    return s.busy;
}

// Frame staging: caller memory -> pinned buffer -> hipMemcpy2DAsync into the
// level-0 planes (replaces the clone/convert chain of MultiCameraFrame::setData).
int Rig::upload_u8(int slot, const uint8_t *const *images, int nimg, int stride)
{
    if (slot < 0 || slot >= (int)slots.size() || nimg < 1 || nimg > max_images || !images || stride < W) {
        set_error("upload_u8: bad argument");
        return MCORB_E_ARG;
    }
    Slot &s = *slots[slot];
    if (slot_busy(s)) return MCORB_E_STATE;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(s.st));   // staging buffer free again
    for (int m = 0; m < nimg; m++)
        if (!images[m]) { set_error("upload_u8: empty image"); return MCORB_E_EMPTY; }
    // pageable caller memory -> pinned staging buffer: one pool task per image (the calling thread takes its share)
    auto copy_one = [&](int m, int) {
        uint8_t *dst = s.h_stage + (size_t)m * W * H;
        if (stride == W) memcpy(dst, images[m], (size_t)W * H);
        else for (int y = 0; y < H; y++) memcpy(dst + (size_t)y * W, images[m] + (size_t)y * stride, W);
    };
    if (nimg > 1 && nimg <= kSmallBatch && upload_pipelined) {
        // one rig frame at a time: the staging copy and the DMA overlap -- the planes are copied a quarter at a time, image by image,
        // and whoever finishes an image's last quarter puts its DMA on the stream while the others go on with the next image
        constexpr int Q = 4;
        std::atomic<int> left[kSmallBatch];
        for (int m = 0; m < nimg; m++) left[m].store(Q);
        std::atomic<int> err{(int)hipSuccess};
        const size_t plane = (size_t)W * H;
        auto quarter = [&](int t, int) {
            const int m = t / Q, q = t % Q, y0 = H * q / Q, y1 = H * (q + 1) / Q;
            uint8_t *dst = s.h_stage + (size_t)m * plane;
            if (stride == W) memcpy(dst + (size_t)y0 * W, images[m] + (size_t)y0 * W, (size_t)(y1 - y0) * W);
            else for (int y = y0; y < y1; y++) memcpy(dst + (size_t)y * W, images[m] + (size_t)y * stride, W);
            if (left[m].fetch_sub(1, std::memory_order_acq_rel) != 1) return;
            hipError_t e = hipSetDevice(device);   // (pool threads make no other HIP call)
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(s.d_pyr + (size_t)m * geom.imgBytes + geom.lv[0].off, geom.lv[0].pitch, dst, W, W, H, hipMemcpyHostToDevice, s.st);
            if (e != hipSuccess) err.store((int)e);
        };
        pool->parallel_for(nimg * Q, quarter, pool_threads + s.index);
        HIPCHK((hipError_t)err.load());
        return MCORB_OK;
    }
    if (nimg > 1) pool->parallel_for(nimg, copy_one, pool_threads + s.index);
    else copy_one(0, 0);
    return upload_staged(slot, nimg);
}

// DMA of the slot's pinned staging buffer (image m at m*W*H, row stride W) into level 0 of the pyramid planes.
int Rig::upload_staged(int slot, int nimg)
{
    if (slot < 0 || slot >= (int)slots.size() || nimg < 1 || nimg > max_images) { set_error("upload_staged: bad argument"); return MCORB_E_ARG; }
    Slot &s = *slots[slot];
    if (slot_busy(s)) return MCORB_E_STATE;
    HIPCHK(hipSetDevice(device));
    const size_t plane = (size_t)W * H;
    if (geom.lv[0].pitch == W) {
        // level-0 rows are contiguous: the whole batch is one strided copy (one row = one image)
        HIPCHK(hipMemcpy2DAsync(s.d_pyr + geom.lv[0].off, geom.imgBytes, s.h_stage, plane, plane, nimg, hipMemcpyHostToDevice, s.st));
    } else {
        for (int m = 0; m < nimg; m++)
            HIPCHK(hipMemcpy2DAsync(s.d_pyr + (size_t)m * geom.imgBytes + geom.lv[0].off, geom.lv[0].pitch, s.h_stage + m * plane, W, W, H,
                                    hipMemcpyHostToDevice, s.st));
    }
    return MCORB_OK;
}

int Rig::upload_f32(int slot, const float *const *images, int nimg, int stride_bytes, int channels)
{
    if (slot < 0 || slot >= (int)slots.size() || nimg < 1 || nimg > max_images || !images ||
        (channels != 1 && channels != 3) || stride_bytes < W * channels * 4 || (stride_bytes & 3)) {
        set_error("upload_f32: bad argument");
        return MCORB_E_ARG;
    }
    Slot &s = *slots[slot];
    if (slot_busy(s)) return MCORB_E_STATE;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(s.st));
    const size_t row_f = (size_t)W * channels, img_f = row_f * H;
    const size_t need = img_f * 4 * (size_t)max_images;
    if (s.f32_bytes < need) {
        if (s.d_f32) HIPCHK(hipFree(s.d_f32));
        HIPCHK(hipMalloc((void **)&s.d_f32, need));
        s.f32_bytes = need;
    }
    for (int m = 0; m < nimg; m++) {
        if (!images[m]) { set_error("upload_f32: empty image"); return MCORB_E_EMPTY; }
        HIPCHK(hipMemcpy2DAsync(s.d_f32 + (size_t)m * img_f, row_f * 4, images[m], stride_bytes, row_f * 4, H,
                                hipMemcpyHostToDevice, s.st));
    }
    launch_stage_f32(s.st, s.d_f32, W, H, (int)row_f, channels, img_f, s.d_pyr, geom, nimg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s.st));   // caller memory may be pageable: copies above were staged by the runtime
    return MCORB_OK;
}

int Rig::submit(int slot, const Job &job)
{
    if (slot < 0 || slot >= (int)slots.size()) { set_error("bad slot"); return MCORB_E_ARG; }
    Slot &s = *slots[slot];
    std::unique_lock<std::mutex> lk(s.m);
    if (s.busy) { set_error("slot busy"); return MCORB_E_STATE; }
    s.job = job;
    s.busy = true;
    s.status = MCORB_OK;
    lk.unlock();
    s.cv.notify_all();
    return MCORB_OK;
}

int Rig::wait(int slot)
{
    if (slot < 0 || slot >= (int)slots.size()) { set_error("bad slot"); return MCORB_E_ARG; }
    Slot &s = *slots[slot];
    std::unique_lock<std::mutex> lk(s.m);
    s.cv.wait(lk, [&s] { return !s.busy; });
    if (s.status != MCORB_OK) set_error(s.err);
    return s.status;
}

// MCORB_LAT_PROF=1: where a synchronous PROCESS job spends its wall time (host clock), printed every 50 jobs
namespace LatProf {
static const bool on = getenv("MCORB_LAT_PROF") != nullptr;
static thread_local double t[12];
static thread_local double acc[12];
static thread_local int n = 0;
static inline void mark(int i) { if (on) t[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void flush()
{
    if (!on) return;
    for (int i = 1; i < 9; i++) acc[i] += t[i] - t[i - 1];
    acc[9] += t[9] - t[6]; acc[10] += t[10] - t[9];   // inside finish_match: accept lists unpacked | tracks merged
    if (++n % 50 == 0) {
        fprintf(stderr, "[mcorb lat prof] per job: enqueue A %.0f us, wait tables %.0f, select %.0f, prepare+enqueue B %.0f, wait GPU %.0f, post %.0f, merge %.0f (lists %.0f, tracks %.0f), total %.0f\n",
                acc[1] / 50, acc[2] / 50, acc[3] / 50, acc[4] / 50, acc[5] / 50, acc[6] / 50, acc[7] / 50, acc[9] / 50, acc[10] / 50, (acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7]) / 50);
        for (int i = 0; i < 12; i++) acc[i] = 0;
    }
}
}  // namespace LatProf

int Rig::execute(Slot &s, const Job &j)
{
    // A stale error of this thread is not this job's: e.g. an elapsed-time query (mcorb_rig_last_timing, the fallback path's own
    // accounting) on events that a captured graph holds as nodes and no stream ever recorded leaves "invalid resource handle" behind,
    // and the next hipGetLastError() check -- possibly another rig's job on the same thread -- would trip over it.
    (void)hipGetLastError();
    int st = MCORB_OK;
    switch (j.kind) {
    case Job::EXTRACT:
        if (gpu_select) { st = run_gpu_selected(s, j, false); break; }
        st = run_extract_phaseA(s, j);
        if (st == MCORB_OK) st = run_select_and_describe(s, j, false);
        break;
    case Job::PROCESS:
        LatProf::mark(0);
        if (gpu_select) {
            st = run_gpu_selected(s, j, true);   // (marks 1 .. 5 inside: enqueue | - | - | - | wait GPU | post)
        } else {
            st = run_extract_phaseA(s, j);
            LatProf::mark(1);
            if (st == MCORB_OK) st = run_select_and_describe(s, j, true);
        }
        LatProf::mark(6);
        if (st == MCORB_OK) st = finish_match(s, j);
        LatProf::mark(7);
        LatProf::flush();
        break;
    case Job::MATCH:
        st = enqueue_match(s, j, false);
        if (st == MCORB_OK) {
            hipError_t e = hipEventRecord(s.ev[10], s.st);
            if (e == hipSuccess) e = wait_event(s.ev[10]);
            if (e != hipSuccess) { set_error(hipGetErrorString(e)); st = MCORB_E_HIP; }
        }
        if (st == MCORB_OK) st = finish_match(s, j);
        break;
    default: break;
    }
    return st;
}

// Synchronous entry points run the job on the calling thread: no hand-off to the slot's driver thread and back
// (two futex wake-ups, ~50-100 us of a 0.6 ms single-frame call).
int Rig::run_sync(int slot, const Job &job)
{
    if (slot < 0 || slot >= (int)slots.size()) { set_error("bad slot"); return MCORB_E_ARG; }
    Slot &s = *slots[slot];
    {
        std::lock_guard<std::mutex> lk(s.m);
        if (s.busy) { set_error("slot busy"); return MCORB_E_STATE; }
        s.busy = true;
        s.inline_job = true;   // the driver thread must not pick this one up
        s.status = MCORB_OK;
    }
    (void)hipSetDevice(device);
    const int st = execute(s, job);
    {
        std::lock_guard<std::mutex> lk(s.m);
        s.status = st;
        if (st != MCORB_OK) s.err = get_error();
        s.busy = false;
        s.inline_job = false;
    }
    s.cv.notify_all();
    return st;
}

void Rig::driver(Slot *sp)
{
    Slot &s = *sp;
    (void)hipSetDevice(device);
    // the polling wait sleeps 20 us at a time: without this the kernel's default 50 us timer slack triples that
    if (wait_mode == 2) (void)prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0);
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> lk(s.m);
            s.cv.wait(lk, [&s] { return s.quit || (s.busy && !s.inline_job); });
            if (s.quit) return;
            j = s.job;
        }
        const int st = execute(s, j);
        {
            std::lock_guard<std::mutex> lk(s.m);
            s.status = st;
            if (st != MCORB_OK) s.err = get_error();
            s.busy = false;
        }
        s.cv.notify_all();
    }
}

int Rig::run_extract_phaseA(Slot &s, const Job &j)
{
    if (j.nimg < 1 || j.nimg > max_images) { set_error("extract: bad image count"); return MCORB_E_ARG; }
    s.invalidate_bow();   // tracks / BoW vectors of the previous batch index keypoints that are about to disappear
    // A single rig frame (how MC-SLAM calls, mc_slam_app.cpp:564-572) is launch- and hand-off-bound: ~25 runtime calls and three
    // small copies around 250 us of kernels.  For small batches the three copies go: k_compact writes its tables straight into the
    // host-mapped h_tbl, the describe / k-NN kernels read the control block from host-mapped h_ctrl, k_describe_fused writes the
    // host's descriptor copy itself (a few hundred KB over PCIe in all).
    static const bool side = getenv("MCORB_COMPACT_SIDE") != nullptr;   // round-1 placement of k_compact, for comparison
    s.small_job = j.nimg <= kSmallBatch && params.orientation == 0 && !blur_planes && !side;   // (the knob keeps the copy path: it must not select half of each)
    s.h_overflow[0] = 0;
    HIPCHK(hipEventRecord(s.ev[0], s.st));
    launch_pyramid(s.st, s.d_pyr, geom, d_taps, resize_win, j.nimg);
    HIPCHK(hipEventRecord(s.ev[1], s.st));
    launch_fast(s.st, s.d_pyr, geom, params.ini_th_fast, params.min_th_fast, d_fasttab + fast_cell_off, s.d_cellkp, s.d_cellcnt, j.nimg);
    HIPCHK(hipEventRecord(s.ev[2], s.st));
    // compaction fills the per-image table blocks in device memory (a short kernel: it runs on the compute stream, ahead
    // of whatever comes next); the DMA that takes the blocks to the host runs on the side stream.
    if (side) HIPCHK(hipStreamWaitEvent(s.st_copy, s.ev[2], 0));
    launch_compact(side ? s.st_copy : s.st, s.d_cellkp, s.d_cellcnt, geom, d_lut, s.d_sorted, s.h_cand, s.small_job ? s.h_tbl : s.d_tbl, s.h_overflow, j.nimg);
    HIPCHK(hipEventRecord(s.ev_c, side ? s.st_copy : s.st));
    if (s.small_job) {
        HIPCHK(hipEventRecord(s.ev[3], s.st));   // the tables are in host memory when k_compact is done
        s.blur_valid = false;
        HIPCHK(hipEventRecord(s.ev[4], s.st));
        HIPCHK(hipGetLastError());
        return MCORB_OK;
    }
    if (!side) HIPCHK(hipStreamWaitEvent(s.st_copy, s.ev_c, 0));
    if (!copy_kernel) HIPCHK(hipMemcpyAsync(s.h_tbl, s.d_tbl, (size_t)j.nimg * s.tbl_ints_per_image * sizeof(int), hipMemcpyDeviceToHost, s.st_copy));
    else launch_copy_to_host(s.st_copy, s.d_tbl, s.h_tbl, (size_t)j.nimg * s.tbl_ints_per_image * sizeof(int));
    HIPCHK(hipEventRecord(s.ev[3], s.st_copy));
    // Reference mode blurs inside the descriptor kernel, only around the kept keypoints (k_describe_fused).  Whole
    // blurred planes are made when the rotated taps of the orientation mode need them, when MCORB_BLUR_PLANES asks for
    // the plane-based path (A/B comparison), or later, on demand, for mcorb_rig_get_blurred.
    s.blur_valid = blur_planes;
    if (blur_planes) launch_blur(s.st, s.d_pyr, s.d_blur, geom, j.nimg);
    HIPCHK(hipEventRecord(s.ev[4], s.st));
    HIPCHK(hipGetLastError());
    return MCORB_OK;
}

int Rig::run_select_and_describe(Slot &s, const Job &j, bool then_match)
{
    HIPCHK(wait_event(s.ev[3]));
    LatProf::mark(2);
    if (s.h_overflow[0]) {
        set_error("candidate list of a sparse level does not fit the host buffer (raise mcorb_params.cand_cap)");
        (void)hipStreamSynchronize(s.st);
        return MCORB_E_OVERFLOW;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int L = geom.nlevels, nimg = j.nimg;
    std::atomic<int> bad{0};
    // selection of one level of one image (worker w's scratch)
    auto select_level = [&](int m, int level, int w) {
        const int *tb = s.tbl(m);
        const int *lo = tb + kTblLvlOff, *shp = tb + kTblShipped;
        const int *bst = tb + kTblHead;
        const BucketWin *win = reinterpret_cast<const BucketWin *>(tb + tbl_win_off(geom.bucketTotal));
        const int n = lo[level + 1] - lo[level];
        std::vector<uint32_t> &out = s.sel_val[(size_t)m * L + level];
        out.resize((size_t)tab.quota[level] + 64);
        std::vector<int> &idx = scratch[w]->idx;
        idx.resize(out.size());
        int r = 0;
        if (n > 0)
            r = select_octree(shp[level] ? s.h_cand + (size_t)m * geom.hostCandCap + lo[level] : nullptr,
                              bst + geom.lv[level].bucket0, win + geom.lv[level].bucket0, n, selp[level],
                              idx.data(), out.data(), *scratch[w]);
        if (r == -3) { bad.store(3); r = 0; }
        if (r < 0) { bad.store(1); r = 0; }
        out.resize(r);
    };
    // selection + assembly: one task per image.  (One task per (image, level) for single rig frames was measured slower twice:
    // in round 2 the extra tasks waited for sleeping workers; in round 3, with the workers woken ahead of time and spinning, the
    // selection still went from 83 to 106 us: the per-image task streams its tables into the cache once, 32 small tasks miss
    // them one by one.  Waking the workers ahead of time by itself was worth 9 us of 92 for twelve spinning cores: not kept.
    // Two tasks per image (levels {0, 3, 4, 7} and {1, 2, 5, 6}; the one that finishes second assembles) with workers that
    // spin for 1 ms between jobs: 88 - 92 against 76 - 87 us: not kept either.)
    pool->parallel_for(nimg, [&](int m, int w) {
        HostProf::Scope prof_task(0);
        const int *tb = s.tbl(m);
        const int *lo = tb + kTblLvlOff, *shp = tb + kTblShipped;
        const int *bst = tb + kTblHead;
        const BucketWin *win = reinterpret_cast<const BucketWin *>(tb + tbl_win_off(geom.bucketTotal));
        {
            // the DMA engine wrote these over PCIe, so they sit in DRAM, not in this core's caches: stream them in with one
            // demand load per cache line instead of taking the misses one by one below (software prefetches were
            // measured slower, most get dropped)
            uint32_t touch = 0;
            const uint32_t *c = s.h_cand + (size_t)m * geom.hostCandCap;
            const uint32_t *b1 = reinterpret_cast<const uint32_t *>(bst);
            const uint32_t *b2 = reinterpret_cast<const uint32_t *>(win);
            for (int l = 0; l < L; l++) {
                if (shp[l])
                    for (int i = lo[l], e = lo[l + 1]; i < e; i += 16) touch += c[i];
                const int b0 = geom.lv[l].bucket0, nb = geom.lv[l].nBuckets + 1;
                for (int i = b0; i < b0 + nb; i += 16) touch += b1[i];
                for (int i = 2 * b0; i < 2 * (b0 + nb); i += 16) touch += b2[i];
            }
            s.touch_sink[m & 15] = touch;   // keeps the loads alive
        }
        {
            HostProf::Scope prof_sel(1);
            for (int level = 0; level < L; level++) select_level(m, level, w);
        }
        // assembly (ORBextractor.cpp:1103-1170): final order, lapping partition, coordinate scaling
        int total = 0;
        for (int l = 0; l < L; l++) total += (int)s.sel_val[(size_t)m * L + l].size();
        if (total > geom.kcap) { bad.store(2); total = 0; }
        std::vector<mcorb_keypoint> &K = s.kps[m];
        K.assign(total, mcorb_keypoint{});
        uint32_t *sel = s.h_sel + (size_t)m * geom.kcap;
        int monoIndex = 0, stereoIndex = total - 1;
        if (total) {
            for (int l = 0; l < L; l++) {
                const float scale = tab.scale[l];
                for (const uint32_t c : s.sel_val[(size_t)m * L + l]) {
                    const int xl = cand_x(c) + kMinBorder, yl = cand_y(c) + kMinBorder;
                    mcorb_keypoint kp;
                    kp.x = (float)xl; kp.y = (float)yl;
                    kp.size = (float)tab.scaled_patch[l];
                    kp.angle = 0.f;
                    kp.response = (float)cand_resp(c);
                    kp.octave = l;
                    kp.class_id = -1;
                    if (l != 0) { kp.x *= scale; kp.y *= scale; }
                    int pos;
                    if (kp.x >= (float)j.lap0 && kp.x <= (float)j.lap1) pos = stereoIndex--;
                    else pos = monoIndex++;
                    K[pos] = kp;
                    sel[pos] = pack_sel(l, xl, yl);
                }
            }
        }
        s.mono[m] = monoIndex;
        s.h_nsel[m] = total;
    }, pool_threads + s.index);
    if (bad.load() == 3) { set_error("internal: quad-tree went below the bucketing without the candidate list"); (void)hipStreamSynchronize(s.st); return MCORB_E_STATE; }
    if (bad.load() == 1) { set_error("selection failed: level too tall"); (void)hipStreamSynchronize(s.st); return MCORB_E_SIZE; }
    if (bad.load()) { set_error("keypoint capacity exceeded"); (void)hipStreamSynchronize(s.st); return MCORB_E_CAP; }
    const auto t1 = std::chrono::steady_clock::now();
    s.timing[1] = std::chrono::duration<float, std::micro>(t1 - t0).count();
    LatProf::mark(3);

    s.nimg_done = nimg;
    if (then_match) TRY(prepare_match(s, j));
    if (s.small_job) {
        // no copies: the kernels read the control block from the host-mapped h_ctrl and k_describe_fused writes h_desc itself
        HIPCHK(hipEventRecord(s.ev[5], s.st));
        launch_describe(s.st, s.d_pyr, nullptr, geom, s.h_sel, s.h_nsel, 0, s.d_desc, s.d_angles, nimg, s.h_desc);
        HIPCHK(hipEventRecord(s.ev[6], s.st));
        if (then_match) TRY(enqueue_match(s, j, true));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s.ev[10], s.st));
        LatProf::mark(4);
        HIPCHK(wait_event(s.ev[10]));
        LatProf::mark(5);
        s.nimg_done = nimg;
        float a = 0, c = 0, t = 0;
        ev_elapsed(&a, s.ev[0], s.ev[2]);
        ev_elapsed(&c, s.ev[5], s.ev[6]);
        s.timing[0] = a * 1000.f;
        s.timing[2] = c * 1000.f;
        s.timing[8] = 0.f;
        s.timing[9] = c * 1000.f;
        ev_elapsed(&t, s.ev[0], s.ev[1]); s.timing[4] = t * 1000.f;
        ev_elapsed(&t, s.ev[1], s.ev[2]); s.timing[5] = t * 1000.f;
        ev_elapsed(&t, s.ev[2], s.ev_c); s.timing[6] = t * 1000.f;
        return MCORB_OK;
    }
    // one H2D copy of the control block: counts, pair list, packed selected keypoints
    HIPCHK(hipMemcpyAsync(s.d_ctrl, s.h_ctrl, s.ctrl_bytes, hipMemcpyHostToDevice, s.st));
    HIPCHK(hipEventRecord(s.ev[5], s.st));
    launch_describe(s.st, s.d_pyr, blur_planes ? s.d_blur : nullptr, geom, s.d_sel, s.d_nsel, params.orientation, s.d_desc, s.d_angles, nimg);
    HIPCHK(hipEventRecord(s.ev[6], s.st));
    // descriptors go back to the host on the side stream (DMA) while the matcher already runs
    static const bool d2h_late = getenv("MCORB_D2H_LATE") != nullptr;   // experiment: copy after the k-NN instead of beside it
    static const int d2h_mid_env = getenv("MCORB_D2H_AFTER_EXPAND") ? atoi(getenv("MCORB_D2H_AFTER_EXPAND")) : -1;
    // (experiment, MCORB_D2H_AFTER_EXPAND=1: start the copy behind k_expand.  Whatever kernel of the job ENDS while the copy's
    // host writes are in flight is held until they have drained -- k_expand 12 -> 148 us beside the copy, k_knn2 154 -> 276 us when
    // the copy starts behind k_expand, with the runtime's blit and with k_copy_to_host alike (profiles/r03_copy_kernel_ab.txt) --
    // so the copy stays beside k_expand, whose result nobody needs before k_knn2 anyway: 13.3 k vs 13.0 k frames/s at one slot.)
    const bool d2h_mid = then_match && !d2h_late && d2h_mid_env > 0;
    if ((!d2h_late && !d2h_mid) || !then_match) HIPCHK(hipStreamWaitEvent(s.st_dma, s.ev[6], 0));
    if (then_match && (d2h_late || d2h_mid)) {
        TRY(enqueue_match(s, j, true));
        HIPCHK(hipStreamWaitEvent(s.st_dma, d2h_mid && s.npairs_done > 0 ? s.ev_e : s.ev[9], 0));
    }
    if (!copy_kernel) HIPCHK(hipMemcpyAsync(s.h_desc, s.d_desc, (size_t)nimg * geom.kcap * 32, hipMemcpyDeviceToHost, s.st_dma));
    else launch_copy_to_host(s.st_dma, s.d_desc, s.h_desc, (size_t)nimg * geom.kcap * 32);
    if (params.orientation)
        HIPCHK(hipMemcpyAsync(s.h_angles, s.d_angles, (size_t)nimg * geom.kcap * sizeof(float), hipMemcpyDeviceToHost, s.st_dma));
    if (then_match && !d2h_late && !d2h_mid) TRY(enqueue_match(s, j, true));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s.ev[10], s.st));
    HIPCHK(hipEventRecord(s.ev[11], s.st_dma));
    HIPCHK(wait_event(s.ev[10]));   // events, not streams: the compute stream may be shared between slots
    HIPCHK(wait_event(s.ev[11]));
    if (params.orientation)
        for (int m = 0; m < nimg; m++)
            for (size_t k = 0; k < s.kps[m].size(); k++) s.kps[m][k].angle = s.h_angles[(size_t)m * geom.kcap + k];
    s.nimg_done = nimg;
    float a = 0, b = 0, c = 0, t = 0;
    ev_elapsed(&a, s.ev[0], s.ev[2]);
    ev_elapsed(&b, s.ev_c, s.ev[4]);
    ev_elapsed(&c, s.ev[5], s.ev[6]);
    s.timing[0] = a * 1000.f;
    s.timing[2] = (b + c) * 1000.f;
    s.timing[8] = b * 1000.f;   // k_blur
    s.timing[9] = c * 1000.f;   // k_describe
    ev_elapsed(&t, s.ev[0], s.ev[1]); s.timing[4] = t * 1000.f;   // pyramid launches
    ev_elapsed(&t, s.ev[1], s.ev[2]); s.timing[5] = t * 1000.f;   // k_fast_cells
    ev_elapsed(&t, s.ev[2], s.ev_c); s.timing[6] = t * 1000.f;   // k_compact (the table DMA behind it is not included)
    return MCORB_OK;
}

// MCORB_SELECT_GPU: the whole job -- pyramid, FAST, compaction, selection, assembly, descriptors, matching -- is enqueued in one go;
// the host comes back when the results have landed (descriptors, the control block's sel / nsel, responses, monoIndex, flags) and
// only builds its keypoint records from them.  A batch with a level whose tree goes below the bucketing depth (flag) is redone
// through the host stage: run_select_and_describe on the tables, exactly the MCORB_SELECT_HOST path.
// everything a GPU-selected job puts on the slot's streams, from the control block's head to the result copies
int Rig::enqueue_gpu_job(Slot &s, const Job &j, bool then_match)
{
    const int nimg = j.nimg;
    (void)hipGetLastError();   // (a stale error of this thread -- e.g. an elapsed-time query on an event a replayed graph never recorded -- is not this job's)
    // A small batch (one rig frame at a time) is launch- and copy-bound: its results travel through host-mapped memory -- k_assemble
    // writes the host's sel / responses / counts itself and signals them per image, k_describe_fused writes the host's descriptors, the
    // matcher reads its pair list from the host's control block: no copy is left in the job.
    const bool small = s.gpu_small;
    if (then_match && !small)   // pair list / set map first: k_assemble overwrites the control block's nsel and sel afterwards, in stream order
        HIPCHK(hipMemcpyAsync(s.d_ctrl, s.h_ctrl, s.ctrl_pairs_end, hipMemcpyHostToDevice, s.st));
    int *d_flags = reinterpret_cast<int *>(s.d_res);
    if (!small) HIPCHK(hipMemsetAsync(d_flags, 0, 16 * sizeof(int), s.st));   // (a small batch's flags travel with its per-image signals)
    // (a small batch replayed from its graph: kernels only -- event-record nodes between them split the graph into separately
    // submitted pieces, and nothing reads these events after a replay)
    const bool ev_on = !(small && s.capturing);
    if (ev_on) HIPCHK(hipEventRecord(s.ev[0], s.st));
    launch_pyramid(s.st, s.d_pyr, geom, d_taps, resize_win, nimg);
    if (ev_on) HIPCHK(hipEventRecord(s.ev[1], s.st));
    launch_fast(s.st, s.d_pyr, geom, params.ini_th_fast, params.min_th_fast, d_fasttab + fast_cell_off, s.d_cellkp, s.d_cellcnt, nimg);
    if (ev_on) HIPCHK(hipEventRecord(s.ev[2], s.st));
    launch_compact(s.st, s.d_cellkp, s.d_cellcnt, geom, d_lut, s.d_sorted, s.h_cand, s.d_tbl, s.h_overflow, nimg);
    if (ev_on) HIPCHK(hipEventRecord(s.ev_c, s.st));
    HIPCHK(launch_select(s.st, s.d_tbl, s.d_sorted, geom, s.d_selval, s.d_selcnt, d_flags, nimg, select_deep_cap));
    if (small)
        launch_assemble(s.st, s.d_selval, s.d_selcnt, geom, tab.scale, j.lap0, j.lap1, s.d_sel, s.d_res + s.res_resp_off, s.d_nsel,
                        reinterpret_cast<int *>(s.d_res + s.res_mono_off), d_flags, nimg, s.h_sel, s.h_res + s.res_resp_off, s.h_sig);
    else
        launch_assemble(s.st, s.d_selval, s.d_selcnt, geom, tab.scale, j.lap0, j.lap1, s.d_sel, s.d_res + s.res_resp_off, s.d_nsel,
                        reinterpret_cast<int *>(s.d_res + s.res_mono_off), d_flags, nimg);
    if (ev_on) HIPCHK(hipEventRecord(s.ev_s, s.st));
    if (ev_on) HIPCHK(hipEventRecord(s.ev[3], s.st));
    if (blur_planes) launch_blur(s.st, s.d_pyr, s.d_blur, geom, nimg);
    if (ev_on) HIPCHK(hipEventRecord(s.ev[4], s.st));
    if (ev_on) HIPCHK(hipEventRecord(s.ev[5], s.st));
    launch_describe(s.st, s.d_pyr, blur_planes ? s.d_blur : nullptr, geom, s.d_sel, s.d_nsel, params.orientation, s.d_desc, s.d_angles, nimg,
                    small ? s.h_desc : nullptr);
    if (ev_on) HIPCHK(hipEventRecord(s.ev[6], s.st));
    if (small) {
        if (then_match) TRY(enqueue_match(s, j, true));
        HIPCHK(hipGetLastError());
        if (ev_on) HIPCHK(hipEventRecord(s.ev[11], s.st));
        return MCORB_OK;
    }
    // results to the host on the side stream while the matcher runs: descriptors, the control block from nsel on (nsel, the set
    // map and pair list as uploaded, sel), responses + monoIndex + flags
    HIPCHK(hipStreamWaitEvent(s.st_dma, s.ev[6], 0));
    if (!copy_kernel) HIPCHK(hipMemcpyAsync(s.h_desc, s.d_desc, (size_t)nimg * geom.kcap * 32, hipMemcpyDeviceToHost, s.st_dma));
    else launch_copy_to_host(s.st_dma, s.d_desc, s.h_desc, (size_t)nimg * geom.kcap * 32);
    HIPCHK(hipMemcpyAsync(s.h_ctrl + s.ctrl_nsel_off, s.d_ctrl + s.ctrl_nsel_off, s.ctrl_pairs_end - s.ctrl_nsel_off + (size_t)nimg * geom.kcap * sizeof(uint32_t),
                          hipMemcpyDeviceToHost, s.st_dma));
    HIPCHK(hipMemcpyAsync(s.h_res, s.d_res, s.res_resp_off + (size_t)nimg * geom.kcap, hipMemcpyDeviceToHost, s.st_dma));
    if (params.orientation)
        HIPCHK(hipMemcpyAsync(s.h_angles, s.d_angles, (size_t)nimg * geom.kcap * sizeof(float), hipMemcpyDeviceToHost, s.st_dma));
    if (then_match) TRY(enqueue_match(s, j, true));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s.ev[11], s.st_dma));
    return MCORB_OK;
}

// MCORB_SELECT_GPU: the whole job -- pyramid, FAST, compaction, selection, assembly, descriptors, matching -- is enqueued in one go;
// the host comes back when the results have landed (descriptors, the control block's sel / nsel, responses, monoIndex, flags) and
// only builds its keypoint records from them.  A batch with a level whose tree goes below the bucketing depth (flag) is redone
// through the host stage: run_select_and_describe on the tables, exactly the MCORB_SELECT_HOST path.
void Rig::gpu_job_begin()
{
    if (!gpu_job_limit) return;
    std::unique_lock<std::mutex> lk(gpu_jobs_m);
    gpu_jobs_cv.wait(lk, [this] { return gpu_jobs_running < gpu_job_limit; });
    gpu_jobs_running++;
}
void Rig::gpu_job_end()
{
    if (!gpu_job_limit) return;
    {
        std::lock_guard<std::mutex> lk(gpu_jobs_m);
        gpu_jobs_running--;
    }
    gpu_jobs_cv.notify_one();
}

int Rig::run_gpu_selected(Slot &s, const Job &j, bool then_match)
{
    if (j.nimg < 1 || j.nimg > max_images) { set_error("extract: bad image count"); return MCORB_E_ARG; }
    const int nimg = j.nimg;
    s.invalidate_bow();
    s.small_job = false;
    s.h_overflow[0] = 0;
    s.nimg_done = nimg;
    s.gpu_small = nimg <= kSmallBatch && params.orientation == 0 && !blur_planes && !j.ext_desc && geom.kcap <= kSelSignalMaxCount;
    if (then_match) TRY(prepare_match(s, j));
    if (s.gpu_small)
        for (int m = 0; m < nimg; m++) reinterpret_cast<volatile unsigned long long *>(s.h_sig)[m] = 0;
    s.blur_valid = blur_planes;
    // The job is the same ~20 launches and copies every time: captured once per (slot, shape of the job) into a HIP graph and
    // replayed with one call -- the CPU side of a job drops from ~20 runtime calls to one, the gaps between its kernels shrink.
    // (Per-kernel HIP events do not exist inside a replayed graph: mcorb_rig_last_timing reports the job as a whole then.)
    // graph_every: 0 = never, 1 = every job, K > 1 = all but every K-th job of a slot, which runs launch by launch with its
    // per-kernel events (a sample of the same pipeline for mcorb_rig_last_timing)
    // keypoint records (ORBextractor.cpp:1103-1170's fields) of one image from what k_assemble left
    auto records = [&](int m) {
        const int n = s.h_nsel[m];
        std::vector<mcorb_keypoint> &K = s.kps[m];
        K.resize((size_t)n);
        const uint32_t *sel = s.h_sel + (size_t)m * geom.kcap;
        const uint8_t *rs = s.h_res + s.res_resp_off + (size_t)m * geom.kcap;
        const float *ang = params.orientation ? s.h_angles + (size_t)m * geom.kcap : nullptr;
        for (int k = 0; k < n; k++) {
            const uint32_t v = sel[k];
            const int l = (int)(v >> 28), yl = (int)((v >> 14) & 0x3fffu), xl = (int)(v & 0x3fffu);
            mcorb_keypoint kp;
            kp.x = (float)xl; kp.y = (float)yl;
            kp.size = (float)tab.scaled_patch[l];
            kp.angle = ang ? ang[k] : 0.f;
            kp.response = (float)rs[k];
            kp.octave = l;
            kp.class_id = -1;
            if (l != 0) { kp.x *= tab.scale[l]; kp.y *= tab.scale[l]; }
            K[k] = kp;
        }
        s.mono[m] = reinterpret_cast<const int *>(s.h_res + s.res_mono_off)[m];
    };
    // a small batch: the records are built here, image by image as k_assemble signals them, while the descriptor and matching
    // kernels still run; returns the number of images done (all of them unless the job ended without signalling: an error), -1 on
    // a HIP error
    int records_done = 0, small_flags = 0, early_stale = 0;
    auto records_early = [&](hipEvent_t end) {
        const volatile unsigned long long *sig = s.h_sig;
        for (int m = 0; m < nimg; m++) {
            unsigned spins = 0;
            while (!sig[m]) {
                if ((++spins & (wait_mode == 2 ? 0u : 1023u)) == 0) {
                    const hipError_t e = hipEventQuery(end);
                    if (e == hipSuccess) { if (!sig[m]) return; break; }   // the job is over: nothing more will be signalled
                    if (e != hipErrorNotReady) { set_error(std::string("event query: ") + hipGetErrorString(e)); records_done = -1; return; }
                }
                if (wait_mode == 2) { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }   // (slot drivers that poll: no core per slot)
                else __builtin_ia32_pause();
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            const unsigned long long w = sig[m];   // count and monoIndex travel in the signal word itself (sel_signal())
            small_flags |= sel_signal_bad(w);
            const int n = std::min(sel_signal_count(w), geom.kcap);   // (k_assemble never signals more than kcap; the clamp is for the reads below)
            s.h_nsel[m] = n;
            reinterpret_cast<int *>(s.h_res + s.res_mono_off)[m] = sel_signal_mono(w);
            // ... and so does a checksum of the sel / response values k_assemble sent: values the word does not vouch for have not
            // all landed yet (never seen with the atomic signal; cheap to be sure) -- this image and the ones behind it are expanded
            // after the job's end event instead
            uint32_t x = 0;
            const uint32_t *sel = s.h_sel + (size_t)m * geom.kcap;
            const uint8_t *rs = s.h_res + s.res_resp_off + (size_t)m * geom.kcap;
            for (int k = 0; k < n; k++) x ^= sel_check(sel[k], rs[k], k);
            if (x != sel_signal_check(w)) { early_stale++; return; }
            records(m);
            records_done = m + 1;
        }
    };
    const int ge = graph_every.load(std::memory_order_relaxed);
    const bool graphed = ge > 0 && !j.ext_desc && (ge == 1 || (++s.job_counter % ge) != 0);
    struct GpuTurn {   // this slot's turn on the GPU: from the first launch until the results have landed
        Rig &r; bool held = true;
        explicit GpuTurn(Rig &rig) : r(rig) { r.gpu_job_begin(); }
        void done() { if (held) { held = false; r.gpu_job_end(); } }
        ~GpuTurn() { done(); }
    } turn(*this);
    if (graphed) {
        const Slot::GraphKey key{nimg, then_match ? 1 : 0, j.nframes, j.lap0, j.lap1, j.dist_thresh, j.ratio};
        if (!s.graph_exec || memcmp(&key, &s.graph_key, sizeof(key)) != 0) {
            if (s.graph_exec) { (void)hipGraphExecDestroy(s.graph_exec); s.graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(s.st, hipStreamCaptureModeThreadLocal));
            s.capturing = true;
            int st = enqueue_gpu_job(s, j, then_match);
            s.capturing = false;
            hipError_t e = s.gpu_small ? hipSuccess : hipStreamWaitEvent(s.st, s.ev[11], 0);   // the side stream joins again
            const hipError_t e2 = hipStreamEndCapture(s.st, &graph);
            if (st != MCORB_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
            if (e == hipSuccess) e = e2;
            if (e == hipSuccess) e = hipGraphInstantiate(&s.graph_exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { s.graph_exec = nullptr; set_error(std::string("graph capture: ") + hipGetErrorString(e)); return MCORB_E_HIP; }
            s.graph_key = key;
        }
        HIPCHK(hipEventRecord(s.ev_g, s.st));
        HIPCHK(hipGraphLaunch(s.graph_exec, s.st));
        HIPCHK(hipEventRecord(s.ev[10], s.st));
        LatProf::mark(1); LatProf::mark(2); LatProf::mark(3);
        if (s.gpu_small) records_early(s.ev[10]);
        LatProf::mark(4);
        HIPCHK(wait_event(s.ev[10]));
    } else {
        TRY(enqueue_gpu_job(s, j, then_match));
        HIPCHK(hipEventRecord(s.ev[10], s.st));
        LatProf::mark(1); LatProf::mark(2); LatProf::mark(3);
        if (s.gpu_small) records_early(s.ev[10]);
        LatProf::mark(4);
        HIPCHK(wait_event(s.ev[10]));
        HIPCHK(wait_event(s.ev[11]));
    }
    LatProf::mark(5);
    turn.done();
    if (s.h_overflow[0]) { set_error("candidate list of a sparse level does not fit the host buffer (raise mcorb_params.cand_cap)"); return MCORB_E_OVERFLOW; }
    if (records_done < 0) return MCORB_E_HIP;
    if (s.gpu_small && records_done < nimg) {
        // the job is over: every signal word and everything behind it has landed
        for (int m = records_done; m < nimg; m++) {
            const unsigned long long w = reinterpret_cast<const volatile unsigned long long *>(s.h_sig)[m];
            if (!sel_signal_done(w)) { set_error("extract: the job ended without its results"); return MCORB_E_HIP; }
            small_flags |= sel_signal_bad(w);
            s.h_nsel[m] = std::min(sel_signal_count(w), geom.kcap);
            reinterpret_cast<int *>(s.h_res + s.res_mono_off)[m] = sel_signal_mono(w);
        }
        if (early_stale) {
            s.stale_reads += early_stale;
            static const bool dbg = getenv("MCORB_DEBUG_EMPTY") != nullptr;
            if (dbg) fprintf(stderr, "[mcorb debug] image %d: values read ahead of the signal word's checksum; records redone after the end event\n", records_done);
        }
    }
    const int flags = s.gpu_small ? small_flags : reinterpret_cast<const int *>(s.h_res)[0];
    if (flags) {
        // the host stage on the same tables (bit 0: a tree below the bucketing depth; bit 1: more than kcap keypoints -- the host
        // stage reports that error itself)
        s.fallbacks++;
        s.graph_timing = false;
        s.gpu_small = false;   // (the host stage uploads the whole control block and copies its results back)
        HIPCHK(hipMemcpyAsync(s.h_tbl, s.d_tbl, (size_t)nimg * s.tbl_ints_per_image * sizeof(int), hipMemcpyDeviceToHost, s.st_copy));
        HIPCHK(hipEventRecord(s.ev[3], s.st_copy));
        return run_select_and_describe(s, j, then_match);
    }
    if (records_done < nimg) pool->parallel_for(nimg, [&](int m, int) { records(m); }, pool_threads + s.index);
    {
        // MCORB_DEBUG_EMPTY=1: an image that came back without keypoints is looked at again, after the job: what the device holds
        // (counts of FAST candidates, per-level selections, nsel) against what the host read -- to tell "nothing to find" from a
        // result that was read before it had landed
        static const bool dbg_empty = getenv("MCORB_DEBUG_EMPTY") != nullptr;
        if (dbg_empty)
            for (int m = 0; m < nimg; m++) {
                if (!s.kps[m].empty()) continue;
                (void)hipStreamSynchronize(s.st);
                std::vector<int> cc(geom.cells), sc(geom.nlevels);
                int dn = -1;
                (void)hipMemcpy(cc.data(), s.d_cellcnt + (size_t)m * geom.cells, cc.size() * sizeof(int), hipMemcpyDeviceToHost);
                (void)hipMemcpy(sc.data(), s.d_selcnt + (size_t)m * geom.nlevels, sc.size() * sizeof(int), hipMemcpyDeviceToHost);
                (void)hipMemcpy(&dn, s.d_nsel + m, sizeof(int), hipMemcpyDeviceToHost);
                long long cand = 0;
                for (int v : cc) cand += v;
                long long px = 0;
                std::vector<uint8_t> row(W);
                (void)hipMemcpy(row.data(), s.d_pyr + (size_t)m * geom.imgBytes + geom.lv[0].off + (size_t)(H / 2) * geom.lv[0].pitch, W, hipMemcpyDeviceToHost);
                for (uint8_t v : row) px += v;
                fprintf(stderr, "[mcorb debug empty] image %d of %d: small %d graphed %d | host: sig %d nsel %d | device now: nsel %d, FAST candidates %lld, "
                                "selected per level", m, nimg, (int)s.gpu_small, (int)graphed, s.gpu_small ? (int)(s.h_sig[m] & 0xff) : -1, s.h_nsel[m], dn, cand);
                for (int v : sc) fprintf(stderr, " %d", v);
                fprintf(stderr, " | middle row of level 0 sums to %lld (staging row: ", px);
                long long hx = 0;
                for (int x = 0; x < W; x++) hx += s.h_stage[(size_t)m * W * H + (size_t)(H / 2) * W + x];
                fprintf(stderr, "%lld)\n", hx);
            }
    }
    if (then_match && !j.ext_desc)
        for (size_t i = 0; i < s.match_counts.size(); i++) s.match_counts[i] = s.h_nsel[s.match_sets[i]];
    float a = 0, b = 0, c = 0, t = 0;
    if (graphed) {   // one interval: the whole job
        for (float &v : s.timing) v = 0.f;
        ev_elapsed(&a, s.ev_g, s.ev[10]);
        s.timing[0] = a * 1000.f;
        s.graph_timing = true;
        return MCORB_OK;
    }
    s.graph_timing = false;
    ev_elapsed(&a, s.ev[0], s.ev[2]);
    ev_elapsed(&b, s.ev[3], s.ev[4]);
    ev_elapsed(&c, s.ev[5], s.ev[6]);
    s.timing[0] = a * 1000.f;
    s.timing[2] = (b + c) * 1000.f;
    s.timing[8] = b * 1000.f;
    s.timing[9] = c * 1000.f;
    ev_elapsed(&t, s.ev[0], s.ev[1]); s.timing[4] = t * 1000.f;
    ev_elapsed(&t, s.ev[1], s.ev[2]); s.timing[5] = t * 1000.f;
    ev_elapsed(&t, s.ev[2], s.ev_c); s.timing[6] = t * 1000.f;
    ev_elapsed(&t, s.ev_c, s.ev_s); s.timing[1] = t * 1000.f;   // k_select + k_assemble
    return MCORB_OK;
}

// fills the host side of the control block for a match: per-(frame,cam) sets/counts and the pair list
int Rig::prepare_match(Slot &s, const Job &j)
{
    const bool ext = j.ext_desc != nullptr;
    if (j.ext_pairs) {
        // explicit (query set, train set) pairs of an external block: local set i = the i-th distinct set the list names
        if (!ext || j.ext_npairs < 1 || j.ext_npairs > max_pairs() || j.ext_total < 1 || j.ext_total > ext_cap || (!j.ext_counts && !j.ext_counts_dev)) {
            set_error("match pairs: bad external block or pair count (at most " + std::to_string(max_pairs()) + " pairs per job)");
            return MCORB_E_ARG;
        }
        s.match_external = true;
        s.match_sets.clear();
        s.match_counts.clear();
        std::vector<int> local(j.ext_total, -1);
        if (j.ext_counts)
            for (int i = 0; i < j.ext_total; i++) s.h_extcounts[i] = std::min(std::max(j.ext_counts[i], 0), geom.kcap);
        for (int p = 0; p < j.ext_npairs; p++) {
            int lp[2];
            for (int k = 0; k < 2; k++) {
                const int set = j.ext_pairs[2 * p + k];
                if (set < 0 || set >= j.ext_total) { set_error("match pairs: set index out of range"); return MCORB_E_ARG; }
                if (local[set] < 0) {
                    if ((int)s.match_sets.size() >= max_images) { set_error("match pairs: more than " + std::to_string(max_images) + " distinct sets in one job"); return MCORB_E_ARG; }
                    local[set] = (int)s.match_sets.size();
                    s.h_setmap[local[set]] = set;
                    s.match_sets.push_back(set);
                    s.match_counts.push_back(j.ext_counts ? s.h_extcounts[set] : 0);
                }
                lp[k] = local[set];
            }
            s.h_pairs[p] = int2{lp[0], lp[1]};
        }
        s.nsets_local = (int)s.match_sets.size();
        s.npairs_done = j.ext_npairs;
        s.nframes_done = 0;
        return MCORB_OK;
    }
    if (j.nframes < 1 || j.nframes > max_frames || (!ext && j.nframes * ncams > s.nimg_done)) {
        set_error("match: bad frame count or features not extracted");
        return MCORB_E_STATE;
    }
    const int C = ncams;
    s.match_external = ext;
    s.match_sets.resize((size_t)j.nframes * C);
    s.match_counts.resize((size_t)j.nframes * C);
    if (ext) {
        if (j.ext_total < 1 || j.ext_total > ext_cap || (!j.ext_counts && !j.ext_counts_dev) || !j.ext_sets) {
            set_error("match: bad external block (at most " + std::to_string(ext_cap) + " sets = max(4096, 64 x images per slot))");
            return MCORB_E_ARG;
        }
        if (j.ext_counts)
            for (int i = 0; i < j.ext_total; i++) s.h_extcounts[i] = std::min(std::max(j.ext_counts[i], 0), geom.kcap);
        for (int i = 0; i < j.nframes * C; i++) {
            const int set = j.ext_sets[i];
            if (set < 0 || set >= j.ext_total) { set_error("match: set index out of range"); return MCORB_E_ARG; }
            s.match_sets[i] = set;
            s.match_counts[i] = j.ext_counts ? s.h_extcounts[set] : 0;   // device-resident counts arrive with the job (finish_match)
        }
    } else {
        for (int i = 0; i < j.nframes * C; i++) { s.match_sets[i] = i; s.match_counts[i] = s.h_nsel[i]; }
    }
    s.nframes_done = j.nframes;
    s.nsets_local = j.nframes * C;
    // the k-NN works on LOCAL set indices (frame * cameras + camera): k_expand gathers set setmap[i] into local slot i
    for (int i = 0; i < j.nframes * C; i++) s.h_setmap[i] = s.match_sets[i];
    int p = 0;
    for (int f = 0; f < j.nframes; f++)
        for (int a = 0; a < C - 1; a++)
            for (int b = a + 1; b < C; b++) s.h_pairs[p++] = int2{f * C + a, f * C + b};
    s.npairs_done = p;
    return MCORB_OK;
}

int Rig::enqueue_match(Slot &s, const Job &j, bool ctrl_on_device)
{
    if (!ctrl_on_device) {
        TRY(prepare_match(s, j));
        HIPCHK(hipMemcpyAsync(s.d_ctrl, s.h_ctrl, s.ctrl_pairs_end, hipMemcpyHostToDevice, s.st));
    }
    if (j.after_stream)   // the block is being produced on another stream (a collective): order this stream behind what the
        HIPCHK(hipStreamWaitEvent(s.st, s.ev_x, 0));   // caller had enqueued there at submit time (ev_x, recorded by the submit call)
    if (j.ext_counts_dev) {
        HIPCHK(hipMemcpyAsync(s.d_extcounts, j.ext_counts_dev, (size_t)j.ext_total * sizeof(int), hipMemcpyDeviceToDevice, s.st));
        HIPCHK(hipMemcpyAsync(s.h_extcounts, j.ext_counts_dev, (size_t)j.ext_total * sizeof(int), hipMemcpyDeviceToHost, s.st));
    }
    if (s.npairs_done == 0) return MCORB_OK;
    const bool ext = j.ext_desc != nullptr;
    const bool ev_on = !(s.gpu_small && s.capturing);   // (see enqueue_gpu_job)
    if (ev_on) HIPCHK(hipEventRecord(s.ev[7], s.st));
    const bool hostctrl = ctrl_on_device && (s.small_job || s.gpu_small) && !ext;   // (the fused path of a small batch: no H2D copy was made)
    launch_knn2(s.st, ext ? (const uint8_t *)j.ext_desc : s.d_desc, ext ? s.d_extcounts : (hostctrl && !s.gpu_small ? s.h_nsel : s.d_nsel),
                hostctrl ? s.h_setmap : s.d_setmap, s.nsets_local, hostctrl ? s.h_pairs : s.d_pairs, s.npairs_done, geom.kcap, s.d_exp, s.d_lcounts, s.d_part, j.dist_thresh, j.ratio, s.d_knn, s.h_mlist,
                s.h_mcount, ev_on ? s.ev_e : nullptr, ev_on ? s.ev[8] : nullptr);
    if (ev_on) HIPCHK(hipEventRecord(s.ev[9], s.st));
    HIPCHK(hipGetLastError());
    return MCORB_OK;
}

// The epipolar check of computeIntraMatches(matches, old=true) (MultiCameraFrame.cpp:1178-1207): line in
// image i = F^T * kp2, normalised, squared point-line distance against 3.84 * sigma2[octave].  The mixed
// float/double arithmetic follows the reference's declared types statement by statement.
static bool epipolar_ok(const double *F, const mcorb_keypoint &k1, const mcorb_keypoint &k2, const float *sigma2)
{
    float a = (float)((double)k2.x * F[0] + (double)k2.y * F[3] + F[6]);
    float b = (float)((double)k2.x * F[1] + (double)k2.y * F[4] + F[7]);
    float c = (float)((double)k2.x * F[2] + (double)k2.y * F[5] + F[8]);
    float den = a * a + b * b;
    den = den ? (float)(1. / (double)std::sqrt(den)) : (float)1.;
    a *= den; b *= den; c *= den;
    den = a * a + b * b;
    const float num = a * k1.x + b * k1.y + c;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    const float check_thresh = (float)(3.84 * (double)sigma2[k1.octave]);
    return dsqr < check_thresh;
}

// computeIntraMatches' track merge over the BruteForceMatch lists of one frame
// (MultiCameraFrame.cpp:1167-1268); gate != nullptr adds the old=true epipolar check.
void merge_pair_lists(int C, const int *counts, const uint32_t *const *idx1, const uint32_t *const *idx2, const int *np,
                             const EpipolarGate *gate, std::vector<int32_t> &tr, int &mergeable_out)
{
    tr.clear();
    int ntr = 0, mergeable = 0;
    // keypoint -> track, one flat table for all cameras (per thread: this runs once per rig frame, on pool threads)
    static thread_local std::vector<int> inv_flat;
    int *inv[MCORB_MAX_CAMS];
    {
        size_t total = 0, worst = 0;
        for (int c = 0; c < C; c++) total += (size_t)std::max(counts[c], 0);
        inv_flat.assign(total, -1);
        size_t o = 0;
        for (int c = 0; c < C; c++) { inv[c] = inv_flat.data() + o; o += (size_t)std::max(counts[c], 0); }
        for (int p = 0; p < C * (C - 1) / 2; p++) worst += (size_t)std::max(np[p], 0);
        tr.resize(worst * C);   // a track per accepted match at most; cut to the tracks made at the end
    }
    int32_t *T = tr.data();
    int pl = 0;
    for (int a = 0; a < C - 1; a++) {
        for (int b = a + 1; b < C; b++, pl++) {
            const uint32_t *i1 = idx1[pl], *i2 = idx2[pl];
            for (int k = 0; k < np[pl]; k++) {
                const int fa = (int)i1[k], fb = (int)i2[k];
                const int ma = inv[a][fa], mb = inv[b][fb];
                if (gate && !epipolar_ok(gate->F + 9 * pl, gate->kps[a][fa], gate->kps[b][fb], gate->sigma2)) continue;
                if (ma == -1 && mb == -1) {
                    int32_t *row = T + (size_t)ntr * C;
                    for (int c = 0; c < C; c++) row[c] = -1;
                    row[a] = fa;
                    row[b] = fb;
                    inv[a][fa] = ntr;
                    inv[b][fb] = ntr;
                    ntr++;
                } else {
                    if (ma == -1 && mb != -1) {
                        if (T[(size_t)mb * C + a] == -1) {
                            T[(size_t)mb * C + a] = fa;
                            inv[a][fa] = mb;
                        }
                    }
                    if (ma != -1 && mb != -1) {
                        if (ma != mb) mergeable++;
                    }
                    if (ma != -1 && mb == -1) {
                        T[(size_t)ma * C + b] = fb;
                        inv[b][fb] = ma;
                    }
                }
            }
        }
    }
    tr.resize((size_t)ntr * C);
    mergeable_out = mergeable;
}

void Rig::merge_tracks(Slot &s, int f, const EpipolarGate *gate, std::vector<int32_t> &tr, int &mergeable_out) const
{
    const int C = ncams;
    const uint32_t *i1[MCORB_MAX_CAMS * MCORB_MAX_CAMS], *i2[MCORB_MAX_CAMS * MCORB_MAX_CAMS];
    int np[MCORB_MAX_CAMS * MCORB_MAX_CAMS];
    for (int p = 0; p < npp; p++) {
        i1[p] = s.m_idx1[f * npp + p].data();
        i2[p] = s.m_idx2[f * npp + p].data();
        np[p] = (int)s.m_idx1[f * npp + p].size();
    }
    merge_pair_lists(C, &s.match_counts[(size_t)f * C], i1, i2, np, gate, tr, mergeable_out);
}

// BruteForceMatch's output lists (MultiCameraFrame.cpp:1060-1078) + the track merge, host side,
// after the k-NN tables landed.
int Rig::finish_match(Slot &s, const Job &j)
{
    HostProf::Scope prof(2);
    if (j.ext_counts_dev)   // the counts came over with the job's own stream; the k-NN kernels clamped them the same way
        for (size_t i = 0; i < s.match_sets.size(); i++)
            s.match_counts[i] = std::min(std::max(s.h_extcounts[s.match_sets[i]], 0), geom.kcap);
    (void)ncams;
    auto filter_pair = [&](int pi, int) {   // BruteForceMatch's accept loop for one camera pair (pair index within the job)
        // k_knn2_finalize already compacted the accepted pairs in query order: unpack query << 16 | train
        std::vector<uint32_t> &i1 = s.m_idx1[pi], &i2 = s.m_idx2[pi];
        const int nqb = knn_qblocks(geom.kcap);
        const int *cnt = s.h_mcount + (size_t)pi * nqb;
        const uint32_t *ml = s.h_mlist + (size_t)pi * knn_mlist_stride(geom.kcap);
        int n = 0;
        for (int b = 0; b < nqb; b++) n += cnt[b];
        i1.resize(n); i2.resize(n);
        int k = 0;
        for (int b = 0; b < nqb; b++)
            for (int e = 0; e < cnt[b]; e++, k++) { i1[k] = ml[b * kKnnQueriesPerBlock + e] >> 16; i2[k] = ml[b * kKnnQueriesPerBlock + e] & 0xffffu; }
    };
    auto one_frame = [&](int f, int w) {
        // the accept lists were written by the GPU into pinned host memory: every line is a miss, and the lists are short runs
        // (one per 256 queries) that the hardware prefetcher does not get ahead of -- ask for all of a frame's lines at once
        {
            const int nqb = knn_qblocks(geom.kcap);
            for (int pi = f * npp; pi < (f + 1) * npp; pi++) {
                const int *cnt = s.h_mcount + (size_t)pi * nqb;
                const uint32_t *ml = s.h_mlist + (size_t)pi * knn_mlist_stride(geom.kcap);
                for (int b = 0; b < nqb; b++) {
                    const char *p0 = reinterpret_cast<const char *>(ml + (size_t)b * kKnnQueriesPerBlock);
                    const int bytes = std::min(std::max(cnt[b], 0), kKnnQueriesPerBlock) * 4;
                    for (int o = 0; o < bytes; o += 64) __builtin_prefetch(p0 + o, 0, 0);
                }
            }
        }
        for (int pi = f * npp; pi < (f + 1) * npp; pi++) filter_pair(pi, w);
        LatProf::mark(9);
        merge_tracks(s, f, nullptr, s.tracks[f], s.mergeable[f]);
        LatProf::mark(10);
    };
    // frames are independent (own pair lists, own track table): one pool task each
    if (j.ext_pairs) pool->parallel_for(s.npairs_done, filter_pair, pool_threads + s.index);   // explicit pairs: lists only, the merge is the caller's
    else if (s.nframes_done > 1) pool->parallel_for(s.nframes_done, one_frame, pool_threads + s.index);
    else if (s.nframes_done == 1) one_frame(0, 0);   // (spreading one frame's pairs over the pool was slower: wake-ups)
    if (s.npairs_done > 0 && !s.graph_timing) {
        float m = 0;
        ev_elapsed(&m, s.ev[7], s.ev[9]); s.timing[3] = m * 1000.f;
        ev_elapsed(&m, s.ev_e, s.ev[8]); s.timing[7] = m * 1000.f;   // k_knn2 (k_expand in front of it: timing[3] - [7] - finalize)
    }
    return MCORB_OK;
}

}  // namespace mcorb
