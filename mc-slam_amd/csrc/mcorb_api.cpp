// mcorb_api.cpp -- extern "C" entry points of libmcorb (include/mcorb.h).
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>

#include "mcorb_engine.h"

using namespace mcorb;

struct mcorb_extractor {
    mcorb_params params;
    Rig *rig = nullptr;   // rebuilt when the image size changes
    int w = 0, h = 0;
    // scratch for mcorb_knn2 on host arrays
    uint8_t *d_desc = nullptr, *d_exp = nullptr;
    int *d_lcounts = nullptr;
    uint2 *d_part = nullptr;
    KnnRow *h_rows = nullptr;
    uint32_t *h_mlist = nullptr;
    int *h_mcount = nullptr;
    int *h_counts = nullptr;
    int2 *h_pair = nullptr;
    int kc = 0;
};

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_));                 \
            return MCORB_E_HIP;                                                        \
        }                                                                              \
    } while (0)

extern "C" {

void mcorb_default_params(mcorb_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->nfeatures = 2000;
    p->scale_factor = 1.2f;
    p->nlevels = 8;
    p->ini_th_fast = 20;
    p->min_th_fast = 7;
    p->orientation = MCORB_ORIENT_NONE;
}

const char *mcorb_last_error(void) { return get_error(); }
const char *mcorb_version(void) { return "mcorb 0.1 (gfx950)"; }

int mcorb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

// ---------------------------------------------------------------------------
// rig
// ---------------------------------------------------------------------------
int mcorb_rig_create(const mcorb_params *p, int ncams, int width, int height, int max_frames, int nslots,
                     mcorb_rig **out)
{
    if (!p || !out) { set_error("null argument"); return MCORB_E_ARG; }
    *out = nullptr;
    mcorb_rig *r = new (std::nothrow) mcorb_rig;
    if (!r) { set_error("out of memory"); return MCORB_E_ARG; }
    const int st = r->rig.init(*p, ncams, width, height, max_frames, nslots);
    if (st != MCORB_OK) {
        const std::string keep = get_error();
        delete r;
        set_error(keep);
        return st;
    }
    *out = r;
    return MCORB_OK;
}

void mcorb_rig_destroy(mcorb_rig *r) { delete r; }

int mcorb_rig_upload_u8(mcorb_rig *r, int slot, const uint8_t *const *images, int nimg, int stride)
{
    if (!r) return MCORB_E_ARG;
    return r->rig.upload_u8(slot, images, nimg, stride);
}
int mcorb_rig_staging(mcorb_rig *r, int slot, int m, uint8_t **ptr, int *stride)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size() || m < 0 || m >= r->rig.max_images || !ptr) { set_error("staging: bad argument"); return MCORB_E_ARG; }
    Slot *s = r->rig.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    if (hipStreamSynchronize(s->st) != hipSuccess) { set_error("staging: stream error"); return MCORB_E_HIP; }   // earlier DMA out of this buffer done
    *ptr = s->h_stage + (size_t)m * r->rig.W * r->rig.H;
    if (stride) *stride = r->rig.W;
    return MCORB_OK;
}
int mcorb_rig_upload_staged(mcorb_rig *r, int slot, int nimg)
{
    if (!r) return MCORB_E_ARG;
    return r->rig.upload_staged(slot, nimg);
}
int mcorb_rig_upload_f32(mcorb_rig *r, int slot, const float *const *images, int nimg, int stride_bytes, int channels)
{
    if (!r) return MCORB_E_ARG;
    return r->rig.upload_f32(slot, images, nimg, stride_bytes, channels);
}

int mcorb_rig_extract_submit(mcorb_rig *r, int slot, int nimg, int lap_x0, int lap_x1)
{
    if (!r) return MCORB_E_ARG;
    Job j;
    j.kind = Job::EXTRACT; j.nimg = nimg; j.lap0 = lap_x0; j.lap1 = lap_x1;
    return r->rig.submit(slot, j);
}
int mcorb_rig_extract_wait(mcorb_rig *r, int slot) { return r ? r->rig.wait(slot) : MCORB_E_ARG; }
int mcorb_rig_extract(mcorb_rig *r, int slot, int nimg, int lap_x0, int lap_x1)
{
    if (!r) return MCORB_E_ARG;
    Job j;
    j.kind = Job::EXTRACT; j.nimg = nimg; j.lap0 = lap_x0; j.lap1 = lap_x1;
    return r->rig.run_sync(slot, j);
}

int mcorb_rig_process_submit(mcorb_rig *r, int slot, int nframes, int lap_x0, int lap_x1, float dist_thresh, float ratio)
{
    if (!r) return MCORB_E_ARG;
    Job j;
    j.kind = Job::PROCESS; j.nframes = nframes; j.nimg = nframes * r->rig.ncams; j.lap0 = lap_x0; j.lap1 = lap_x1;
    j.dist_thresh = dist_thresh; j.ratio = ratio;
    return r->rig.submit(slot, j);
}
int mcorb_rig_process_wait(mcorb_rig *r, int slot) { return r ? r->rig.wait(slot) : MCORB_E_ARG; }

int mcorb_rig_match_submit(mcorb_rig *r, int slot, int nframes, float dist_thresh, float ratio)
{
    if (!r) return MCORB_E_ARG;
    Job j;
    j.kind = Job::MATCH; j.nframes = nframes; j.dist_thresh = dist_thresh; j.ratio = ratio;
    return r->rig.submit(slot, j);
}
int mcorb_rig_match_wait(mcorb_rig *r, int slot) { return r ? r->rig.wait(slot) : MCORB_E_ARG; }
int mcorb_rig_match(mcorb_rig *r, int slot, int nframes, float dist_thresh, float ratio)
{
    if (!r) return MCORB_E_ARG;
    Job j;
    j.kind = Job::MATCH; j.nframes = nframes; j.dist_thresh = dist_thresh; j.ratio = ratio;
    return r->rig.run_sync(slot, j);
}
/* extract + match of nframes rig frames in one synchronous call (process_submit + process_wait on the caller's thread) */
int mcorb_rig_process(mcorb_rig *r, int slot, int nframes, int lap_x0, int lap_x1, float dist_thresh, float ratio)
{
    if (!r) return MCORB_E_ARG;
    Job j;
    j.kind = Job::PROCESS; j.nframes = nframes; j.nimg = nframes * r->rig.ncams; j.lap0 = lap_x0; j.lap1 = lap_x1;
    j.dist_thresh = dist_thresh; j.ratio = ratio;
    return r->rig.run_sync(slot, j);
}

static Slot *get_slot(mcorb_rig *r, int slot)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) { set_error("bad rig/slot"); return nullptr; }
    Slot *s = r->rig.slots[slot];
    std::lock_guard<std::mutex> lk(s->m);
    if (s->busy) { set_error("slot busy"); return nullptr; }
    return s;
}

int mcorb_rig_num_keypoints(mcorb_rig *r, int slot, int m)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (m < 0 || m >= s->nimg_done) { set_error("image index out of range"); return MCORB_E_ARG; }
    return (int)s->kps[m].size();
}

int mcorb_rig_get_features(mcorb_rig *r, int slot, int m, mcorb_keypoint *kps, uint8_t *desc, int cap, int *n_out,
                           int *mono_index_out)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (m < 0 || m >= s->nimg_done) { set_error("image index out of range"); return MCORB_E_ARG; }
    const int n = (int)s->kps[m].size();
    if (n_out) *n_out = n;
    if (mono_index_out) *mono_index_out = s->mono[m];
    if (n > cap) { set_error("keypoint buffer too small"); return MCORB_E_CAP; }
    if (kps && n) memcpy(kps, s->kps[m].data(), (size_t)n * sizeof(mcorb_keypoint));
    if (desc && n) memcpy(desc, s->h_desc + (size_t)m * r->rig.geom.kcap * 32, (size_t)n * 32);
    return MCORB_OK;
}

static int pair_index(const Rig &R, int frame, int a, int b)
{
    if (a < 0 || b <= a || b >= R.ncams) return -1;
    int pi = 0;
    for (int i = 0; i < a; i++) pi += R.ncams - 1 - i;
    pi += b - a - 1;
    return frame * R.npp + pi;
}

int mcorb_rig_get_pair_matches(mcorb_rig *r, int slot, int frame, int cam_i, int cam_j, uint32_t *idx1, uint32_t *idx2,
                               int cap, int *n_out)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    const int pi = pair_index(r->rig, frame, cam_i, cam_j);
    if (frame < 0 || frame >= s->nframes_done || pi < 0) { set_error("bad frame/pair"); return MCORB_E_ARG; }
    const int n = (int)s->m_idx1[pi].size();
    if (n_out) *n_out = n;
    if (n > cap) { set_error("match buffer too small"); return MCORB_E_CAP; }
    if (n) {
        memcpy(idx1, s->m_idx1[pi].data(), (size_t)n * 4);
        memcpy(idx2, s->m_idx2[pi].data(), (size_t)n * 4);
    }
    return MCORB_OK;
}

static void decode_rows(const KnnRow *rows, int nq, int32_t *idx, int32_t *dist)
{
    for (int q = 0; q < nq; q++) {
        const KnnRow &k = rows[q];
        idx[2 * q] = knn_idx0(k);
        dist[2 * q] = knn_d0(k);
        idx[2 * q + 1] = knn_idx1(k);
        dist[2 * q + 1] = knn_d1(k);
    }
}

int mcorb_rig_get_pair_knn2(mcorb_rig *r, int slot, int frame, int cam_i, int cam_j, int32_t *idx, int32_t *dist,
                            int cap_rows, int *nq_out)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    const int pi = pair_index(r->rig, frame, cam_i, cam_j);
    if (frame < 0 || frame >= s->nframes_done || pi < 0) { set_error("bad frame/pair"); return MCORB_E_ARG; }
    const int nq = s->match_counts[frame * r->rig.ncams + cam_i];
    if (nq_out) *nq_out = nq;
    if (nq > cap_rows) { set_error("knn buffer too small"); return MCORB_E_CAP; }
    // the k-NN rows stay on the device (the pipeline itself only needs the compacted accept lists): fetch on demand
    std::vector<KnnRow> rows((size_t)std::max(nq, 1));
    HIPCHK(hipSetDevice(r->rig.device));
    if (nq) HIPCHK(hipMemcpy(rows.data(), s->d_knn + (size_t)pi * r->rig.geom.kcap, (size_t)nq * sizeof(KnnRow), hipMemcpyDeviceToHost));
    decode_rows(rows.data(), nq, idx, dist);
    return MCORB_OK;
}

int mcorb_rig_get_tracks(mcorb_rig *r, int slot, int frame, int32_t *tracks, int cap_tracks, int *ntracks_out,
                         int *mergeable_out)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (frame < 0 || frame >= s->nframes_done) { set_error("bad frame"); return MCORB_E_ARG; }
    const int C = r->rig.ncams;
    const int n = (int)(s->tracks[frame].size() / C);
    if (ntracks_out) *ntracks_out = n;
    if (mergeable_out) *mergeable_out = s->mergeable[frame];
    if (n > cap_tracks) { set_error("track buffer too small"); return MCORB_E_CAP; }
    if (n) memcpy(tracks, s->tracks[frame].data(), (size_t)n * C * 4);
    return MCORB_OK;
}

int mcorb_rig_get_tracks_epipolar(mcorb_rig *r, int slot, int frame, const double *F, const mcorb_keypoint *const *kps_undist,
                                  int32_t *tracks, int cap_tracks, int *ntracks_out, int *mergeable_out)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (frame < 0 || frame >= s->nframes_done || !F) { set_error("bad frame / no fundamental matrices"); return MCORB_E_ARG; }
    const int C = r->rig.ncams;
    std::vector<const mcorb_keypoint *> kp(C);
    for (int c = 0; c < C; c++) kp[c] = kps_undist && kps_undist[c] ? kps_undist[c] : s->kps[frame * C + c].data();
    mcorb::EpipolarGate gate{F, kp.data(), r->rig.tab.sigma2};
    std::vector<int32_t> tr;
    int mergeable = 0;
    r->rig.merge_tracks(*s, frame, &gate, tr, mergeable);
    const int n = (int)(tr.size() / C);
    if (ntracks_out) *ntracks_out = n;
    if (mergeable_out) *mergeable_out = mergeable;
    if (n > cap_tracks) { set_error("track buffer too small"); return MCORB_E_CAP; }
    if (n) memcpy(tracks, tr.data(), (size_t)n * C * 4);
    return MCORB_OK;
}

int mcorb_rig_level_size(mcorb_rig *r, int level, int *w, int *h)
{
    if (!r || level < 0 || level >= r->rig.geom.nlevels) return MCORB_E_ARG;
    if (w) *w = r->rig.geom.lv[level].w;
    if (h) *h = r->rig.geom.lv[level].h;
    return MCORB_OK;
}

static int copy_plane(mcorb_rig *r, int slot, int m, int level, bool blurred, uint8_t *dst, int dst_stride)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    const Geom &g = r->rig.geom;
    if (m < 0 || m >= r->rig.max_images || level < 0 || level >= g.nlevels || !dst || dst_stride < g.lv[level].w) {
        set_error("bad plane request");
        return MCORB_E_ARG;
    }
    HIPCHK(hipSetDevice(r->rig.device));
    HIPCHK(hipStreamSynchronize(s->st));
    const LevelGeom &L = g.lv[level];
    if (!blurred) {
        HIPCHK(hipMemcpy2D(dst, dst_stride, s->d_pyr + (size_t)m * g.imgBytes + L.off, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
        return MCORB_OK;
    }
    if (!s->d_blur) {       // reference mode does not even allocate them (k_describe_fused blurs around the keypoints only)
        HIPCHK(hipMalloc((void **)&s->d_blur, (size_t)r->rig.max_images * g.imgBytes));
        HIPCHK(hipMemset(s->d_blur, 0, (size_t)r->rig.max_images * g.imgBytes));
    }
    if (!s->blur_valid) {   // ... nor writes them: make them now from the slot's pyramid
        launch_blur(s->st, s->d_pyr, s->d_blur, g, r->rig.max_images);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->st));
        s->blur_valid = true;
    }
    const uint8_t *src = s->d_blur + (size_t)m * g.imgBytes + L.off;
    // blurred planes live in 16 x 8 tiles on the device (mcorb_common.h): fetch the tiled block, hand back rows
    const size_t rows = ((size_t)L.h + kBlurTileRows - 1) / kBlurTileRows * kBlurTileRows;
    std::vector<uint8_t> tmp(rows * L.pitch);
    HIPCHK(hipMemcpy(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; y++)
        for (int x = 0; x < L.w; x += kBlurTileCols)
            memcpy(dst + (size_t)y * dst_stride + x, tmp.data() + blur_tiled_offset(L.pitch, x, y), (size_t)std::min(kBlurTileCols, L.w - x));
    return MCORB_OK;
}
int mcorb_rig_get_level(mcorb_rig *r, int slot, int m, int level, uint8_t *dst, int dst_stride)
{
    return copy_plane(r, slot, m, level, false, dst, dst_stride);
}
int mcorb_rig_get_blurred(mcorb_rig *r, int slot, int m, int level, uint8_t *dst, int dst_stride)
{
    return copy_plane(r, slot, m, level, true, dst, dst_stride);
}

int mcorb_rig_get_candidates(mcorb_rig *r, int slot, int m, int level, uint32_t *packed, int cap, int *n_out)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    const Geom &g = r->rig.geom;
    if (m < 0 || m >= s->nimg_done || level < 0 || level >= g.nlevels) { set_error("bad candidate request"); return MCORB_E_ARG; }
    // level offsets of the image's table block: in host memory only when the job wrote them there (a small batch of the host
    // selection path) or brought them over; the device copy is what every other job leaves
    int lo_dev[kMaxLevels + 1];
    const int *lo = s->tbl(m) + kTblLvlOff;
    if (!s->small_job) {
        HIPCHK(hipSetDevice(r->rig.device));
        HIPCHK(hipMemcpy(lo_dev, s->d_tbl + (size_t)m * s->tbl_ints_per_image + kTblLvlOff, sizeof(lo_dev), hipMemcpyDeviceToHost));
        lo = lo_dev;
    }
    const int n = lo[level + 1] - lo[level];
    if (n_out) *n_out = n;
    if (n > cap) { set_error("candidate buffer too small"); return MCORB_E_CAP; }
    if (n) {
        // the device hands candidates over bucketed by quad-tree path; restore vToDistributeKeys order
        // (cell row, cell col, y, x) for the caller
        // (read from the device copy: the list is shipped to host memory only when the selection can need it)
        std::vector<uint32_t> tmp(n);
        HIPCHK(hipSetDevice(r->rig.device));
        HIPCHK(hipMemcpy(tmp.data(), s->d_sorted + (size_t)m * g.candCap + lo[level], (size_t)n * 4, hipMemcpyDeviceToHost));
        const uint32_t *src = tmp.data();
        const int wc = g.lv[level].wCell, hc = g.lv[level].hCell;
        std::vector<std::pair<uint64_t, uint32_t>> v(n);
        for (int i = 0; i < n; i++) {
            const int x = cand_x(src[i]), y = cand_y(src[i]);
            v[i] = {((((uint64_t)((y - 3) / hc) << 12 | (uint64_t)((x - 3) / wc)) << 12 | (uint64_t)y) << 12) | (uint64_t)x, src[i]};
        }
        std::sort(v.begin(), v.end());
        for (int i = 0; i < n; i++) packed[i] = v[i].second;
    }
    return MCORB_OK;
}

int mcorb_rig_last_timing(mcorb_rig *r, int slot, float us[10])
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    for (int i = 0; i < 10; i++) us[i] = s->timing[i];
    return MCORB_OK;
}

int mcorb_rig_kcap(mcorb_rig *r) { return r ? r->rig.geom.kcap : MCORB_E_ARG; }
int mcorb_rig_select_mode(mcorb_rig *r) { return r ? (r->rig.gpu_select ? MCORB_SELECT_GPU : MCORB_SELECT_HOST) : MCORB_E_ARG; }
int mcorb_rig_set_graph(mcorb_rig *r, int every)
{
    if (!r || every < 0) return MCORB_E_ARG;
    r->rig.graph_every.store(r->rig.gpu_select ? every : 0);
    return MCORB_OK;
}
int mcorb_rig_select_fallbacks(mcorb_rig *r, int slot)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) return MCORB_E_ARG;
    return r->rig.slots[slot]->fallbacks;
}
int mcorb_rig_early_reads_rejected(mcorb_rig *r, int slot)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) return MCORB_E_ARG;
    return r->rig.slots[slot]->stale_reads;
}
int mcorb_dev_sort_selftest(int device, const uint32_t *keys, int n, uint32_t *perm_dev, uint32_t *perm_std)
{
    if (n < 0 || n > 6000 || (n && (!keys || !perm_dev || !perm_std))) { set_error("sort_selftest: bad argument"); return MCORB_E_ARG; }
    if (n == 0) return MCORB_OK;
    std::vector<uint64_t> a((size_t)n), b((size_t)n);
    for (int i = 0; i < n; i++) a[i] = ((uint64_t)keys[i] << 32) | (uint32_t)i;
    b = a;
    std::sort(b.begin(), b.end(), [](uint64_t x, uint64_t y) { return (x >> 32) < (y >> 32); });
    for (int i = 0; i < n; i++) perm_std[i] = (uint32_t)b[i];
    HIPCHK(hipSetDevice(device));
    uint64_t *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_in, (size_t)n * 8));
    if (hipMalloc((void **)&d_out, (size_t)n * 8) != hipSuccess) { (void)hipFree(d_in); set_error("sort_selftest: out of device memory"); return MCORB_E_HIP; }
    hipError_t e = hipMemcpy(d_in, a.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = sort_selftest(d_in, n, d_out);
    if (e == hipSuccess) e = hipMemcpy(b.data(), d_out, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return MCORB_E_HIP; }
    for (int i = 0; i < n; i++) perm_dev[i] = (uint32_t)b[i];
    return MCORB_OK;
}
int mcorb_rig_host_threads(mcorb_rig *r) { return r ? r->rig.pool_threads : MCORB_E_ARG; }
int mcorb_rig_info(mcorb_rig *r, int32_t out[8])
{
    if (!r || !out) return MCORB_E_ARG;
    const Geom &g = r->rig.geom;
    out[0] = g.kcap; out[1] = g.cells; out[2] = g.tiles; out[3] = g.cellCap; out[4] = g.candCap; out[5] = g.bucketTotal;
    out[6] = (int32_t)g.imgBytes; out[7] = g.nlevels;
    return MCORB_OK;
}
void *mcorb_rig_desc_device_ptr(mcorb_rig *r, int slot)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) return nullptr;
    return r->rig.slots[slot]->d_desc;
}
void *mcorb_rig_stream(mcorb_rig *r, int slot)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) return nullptr;
    return (void *)r->rig.slots[slot]->st;
}

int mcorb_rig_match_external_submit(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts, int ntotal,
                                    const int32_t *sets, int nframes, float dist_thresh, float ratio)
{
    if (!r || !desc_dev || !counts || !sets) { set_error("null argument"); return MCORB_E_ARG; }
    Job j;
    j.kind = Job::MATCH; j.nframes = nframes; j.dist_thresh = dist_thresh; j.ratio = ratio;
    j.ext_desc = desc_dev; j.ext_counts = counts; j.ext_total = ntotal; j.ext_sets = sets;
    return r->rig.submit(slot, j);
}

int mcorb_rig_match_external(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts, int ntotal,
                             const int32_t *sets, int nframes, float dist_thresh, float ratio)
{
    const int st = mcorb_rig_match_external_submit(r, slot, desc_dev, counts, ntotal, sets, nframes, dist_thresh, ratio);
    return st != MCORB_OK ? st : r->rig.wait(slot);
}

int mcorb_rig_match_external_dev_submit(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts_dev, int ntotal,
                                        const int32_t *sets, int nframes, float dist_thresh, float ratio, void *after_stream)
{
    if (!r || !desc_dev || !counts_dev || !sets) { set_error("null argument"); return MCORB_E_ARG; }
    Job j;
    j.kind = Job::MATCH; j.nframes = nframes; j.dist_thresh = dist_thresh; j.ratio = ratio;
    j.ext_desc = desc_dev; j.ext_counts_dev = counts_dev; j.ext_total = ntotal; j.ext_sets = sets;
    j.after_stream = (hipStream_t)after_stream;
    if (after_stream) {
        // "everything enqueued so far on after_stream": the event is recorded HERE, on the caller's thread (the slot is
        // idle, so its hand-off event is free); the slot's driver only makes the slot's stream wait for it.  Recorded by
        // the driver when it dequeues the job, the event also covered whatever the caller had enqueued meanwhile, and a
        // second thread was recording events on a stream the caller owns.
        Slot *s = get_slot(r, slot);
        if (!s) return MCORB_E_STATE;
        HIPCHK(hipSetDevice(r->rig.device));
        HIPCHK(hipEventRecord(s->ev_x, (hipStream_t)after_stream));
    }
    return r->rig.submit(slot, j);
}

// ---- pair-partitioned matching (pair (i, j) of a frame on one GPU, the tables back to one rank for the merge) ----
int mcorb_rig_match_pairs_external_dev_submit(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts_dev, int ntotal,
                                              const int32_t *pair_sets, int npairs, float dist_thresh, float ratio, void *after_stream)
{
    if (!r || !desc_dev || !counts_dev || !pair_sets) { set_error("null argument"); return MCORB_E_ARG; }
    Job j;
    j.kind = Job::MATCH; j.nframes = 0; j.dist_thresh = dist_thresh; j.ratio = ratio;
    j.ext_desc = desc_dev; j.ext_counts_dev = counts_dev; j.ext_total = ntotal; j.ext_pairs = pair_sets; j.ext_npairs = npairs;
    j.after_stream = (hipStream_t)after_stream;
    if (after_stream) {
        Slot *s = get_slot(r, slot);
        if (!s) return MCORB_E_STATE;
        HIPCHK(hipSetDevice(r->rig.device));
        HIPCHK(hipEventRecord(s->ev_x, (hipStream_t)after_stream));
    }
    return r->rig.submit(slot, j);
}

int mcorb_rig_match_pairs_external(mcorb_rig *r, int slot, const void *desc_dev, const int32_t *counts, int ntotal,
                                   const int32_t *pair_sets, int npairs, float dist_thresh, float ratio)
{
    if (!r || !desc_dev || !counts || !pair_sets) { set_error("null argument"); return MCORB_E_ARG; }
    Job j;
    j.kind = Job::MATCH; j.nframes = 0; j.dist_thresh = dist_thresh; j.ratio = ratio;
    j.ext_desc = desc_dev; j.ext_counts = counts; j.ext_total = ntotal; j.ext_pairs = pair_sets; j.ext_npairs = npairs;
    return r->rig.run_sync(slot, j);   // on the calling thread: no hand-off to the slot's driver and back
}

int mcorb_rig_get_pairlist(mcorb_rig *r, int slot, int pair, uint32_t *idx1, uint32_t *idx2, int cap, int *n_out)
{
    if (n_out) *n_out = 0;
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (pair < 0 || pair >= s->npairs_done) { set_error("bad pair index"); return MCORB_E_ARG; }
    const int n = (int)s->m_idx1[pair].size();
    if (n_out) *n_out = n;
    if (n > cap || (n && (!idx1 || !idx2))) { set_error("pair list: output too small"); return MCORB_E_CAP; }
    if (n) {
        memcpy(idx1, s->m_idx1[pair].data(), (size_t)n * sizeof(uint32_t));
        memcpy(idx2, s->m_idx2[pair].data(), (size_t)n * sizeof(uint32_t));
    }
    return MCORB_OK;
}

// ---- device-resident descriptor sets (N1: findInterMatches' knnMatch between the LF descriptors of consecutive keyframes,
//      FrontEnd.cpp:3344-3500: the previous keyframe's set stays in HBM, only the new one is uploaded) ----
struct mcorb_descblock {
    int device = 0, nsets = 0, kcap = 0;
    uint8_t *d_desc = nullptr;
    int32_t *d_counts = nullptr;
    std::vector<int32_t> h_counts;
};

int mcorb_descblock_create(int device, int nsets, int kcap, mcorb_descblock **out)
{
    if (out) *out = nullptr;
    if (!out || nsets < 1 || kcap < 64 || (kcap & 63)) { set_error("descblock: kcap must be a positive multiple of 64"); return MCORB_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { set_error("no such device"); return MCORB_E_NODEVICE; }
    HIPCHK(hipSetDevice(device));
    mcorb_descblock *b = new mcorb_descblock;
    b->device = device; b->nsets = nsets; b->kcap = kcap;
    b->h_counts.assign(nsets, 0);
    if (hipMalloc((void **)&b->d_desc, (size_t)nsets * kcap * 32) != hipSuccess || hipMalloc((void **)&b->d_counts, (size_t)nsets * sizeof(int32_t)) != hipSuccess ||
        hipMemset(b->d_counts, 0, (size_t)nsets * sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(b->d_desc); (void)hipFree(b->d_counts);
        delete b;
        set_error("descblock: out of device memory");
        return MCORB_E_HIP;
    }
    *out = b;
    return MCORB_OK;
}

void mcorb_descblock_destroy(mcorb_descblock *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    (void)hipFree(b->d_desc); (void)hipFree(b->d_counts);
    delete b;
}

int mcorb_descblock_upload(mcorb_descblock *b, int set, const uint8_t *desc, int n)
{
    if (!b || set < 0 || set >= b->nsets || n < 0 || n > b->kcap || (n && !desc)) { set_error("descblock upload: bad argument"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(b->device));
    b->h_counts[set] = n;
    // (the count first, asynchronously from the block's own array; the blocking copy behind it on the same stream covers both)
    HIPCHK(hipMemcpyAsync(b->d_counts + set, &b->h_counts[set], sizeof(int32_t), hipMemcpyHostToDevice, nullptr));
    if (n) HIPCHK(hipMemcpy(b->d_desc + (size_t)set * b->kcap * 32, desc, (size_t)n * 32, hipMemcpyHostToDevice));
    else HIPCHK(hipStreamSynchronize(nullptr));
    return MCORB_OK;
}

void *mcorb_descblock_desc_ptr(mcorb_descblock *b) { return b ? b->d_desc : nullptr; }
int32_t *mcorb_descblock_counts_dev(mcorb_descblock *b) { return b ? b->d_counts : nullptr; }

int mcorb_rig_match_sets(mcorb_rig *r, int slot, mcorb_descblock *b, const int32_t *pair_sets, int npairs, float dist_thresh, float ratio)
{
    if (!r || !b || !pair_sets) { set_error("null argument"); return MCORB_E_ARG; }
    if (b->device != r->rig.device || b->kcap != r->rig.geom.kcap) { set_error("match_sets: the block's device / kcap differ from the rig's (mcorb_rig_kcap)"); return MCORB_E_ARG; }
    return mcorb_rig_match_pairs_external(r, slot, b->d_desc, b->h_counts.data(), b->nsets, pair_sets, npairs, dist_thresh, ratio);
}

int mcorb_rig_get_pairknn2(mcorb_rig *r, int slot, int pair, int32_t *idx, int32_t *dist, int cap_rows, int *nq_out)
{
    if (nq_out) *nq_out = 0;
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (pair < 0 || pair >= s->npairs_done || s->nframes_done != 0) { set_error("bad pair index (explicit-pair matches only)"); return MCORB_E_ARG; }
    const int nq = s->match_counts[s->h_pairs[pair].x];
    if (nq_out) *nq_out = nq;
    if (nq > cap_rows || (nq && (!idx || !dist))) { set_error("knn buffer too small"); return MCORB_E_CAP; }
    std::vector<KnnRow> rows((size_t)std::max(nq, 1));
    HIPCHK(hipSetDevice(r->rig.device));
    if (nq) HIPCHK(hipMemcpy(rows.data(), s->d_knn + (size_t)pair * r->rig.geom.kcap, (size_t)nq * sizeof(KnnRow), hipMemcpyDeviceToHost));
    decode_rows(rows.data(), nq, idx, dist);
    return MCORB_OK;
}

// computeIntraMatches' serial track merge on caller-supplied pair lists (no device): what rank 0 runs on the gathered tables
int mcorb_host_merge_tracks(int ncams, const int32_t *counts, const uint32_t *const *idx1, const uint32_t *const *idx2,
                            const int32_t *npair, int32_t *tracks, int cap_tracks, int *ntracks_out, int *mergeable_out)
{
    if (ntracks_out) *ntracks_out = 0;
    if (mergeable_out) *mergeable_out = 0;
    if (ncams < 2 || ncams > MCORB_MAX_CAMS || !counts || !idx1 || !idx2 || !npair) { set_error("merge_tracks: bad argument"); return MCORB_E_ARG; }
    const int npairs = ncams * (ncams - 1) / 2;
    for (int c = 0; c < ncams; c++)
        if (counts[c] < 0) { set_error("merge_tracks: negative keypoint count"); return MCORB_E_ARG; }
    int pl = 0;
    for (int a = 0; a < ncams - 1; a++)
        for (int b = a + 1; b < ncams; b++, pl++) {
            if (npair[pl] < 0 || (npair[pl] && (!idx1[pl] || !idx2[pl]))) { set_error("merge_tracks: bad pair list"); return MCORB_E_ARG; }
            for (int k = 0; k < npair[pl]; k++)
                if (idx1[pl][k] >= (uint32_t)counts[a] || idx2[pl][k] >= (uint32_t)counts[b]) { /* unsigned: 0x80000000 is not a negative index */ set_error("merge_tracks: index beyond the camera's keypoint count"); return MCORB_E_ARG; }
        }
    (void)npairs;
    std::vector<int32_t> tr;
    int mergeable = 0;
    std::vector<int> cnt(counts, counts + ncams), np(npair, npair + pl);
    merge_pair_lists(ncams, cnt.data(), idx1, idx2, np.data(), nullptr, tr, mergeable);
    const int n = (int)(tr.size() / ncams);
    if (ntracks_out) *ntracks_out = n;
    if (mergeable_out) *mergeable_out = mergeable;
    if (n > cap_tracks || (n && !tracks)) { set_error("merge_tracks: output too small"); return MCORB_E_CAP; }
    if (n) memcpy(tracks, tr.data(), tr.size() * sizeof(int32_t));
    return MCORB_OK;
}

int mcorb_rig_export_descriptors_dev(mcorb_rig *r, int slot, void *dst_dev, int32_t *counts_dev, int nimg, void *then_stream)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (!dst_dev || !counts_dev || nimg < 1 || nimg > s->nimg_done) { set_error("export: bad argument"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(r->rig.device));
    HIPCHK(hipMemcpyAsync(dst_dev, s->d_desc, (size_t)nimg * r->rig.geom.kcap * 32, hipMemcpyDeviceToDevice, s->st));
    HIPCHK(hipMemcpyAsync(counts_dev, s->h_nsel, (size_t)nimg * sizeof(int), hipMemcpyHostToDevice, s->st));   // (pinned; the device mirror is not filled for small batches)
    if (then_stream) {   // whatever the caller enqueues on then_stream next (the collective) runs after the two copies
        HIPCHK(hipEventRecord(s->ev_x, s->st));
        HIPCHK(hipStreamWaitEvent((hipStream_t)then_stream, s->ev_x, 0));
    } else {
        HIPCHK(hipStreamSynchronize(s->st));
    }
    return MCORB_OK;
}

int mcorb_rig_export_descriptors(mcorb_rig *r, int slot, void *dst_dev, int32_t *counts_host, int nimg)
{
    Slot *s = get_slot(r, slot);
    if (!s) return MCORB_E_STATE;
    if (!dst_dev || nimg < 1 || nimg > s->nimg_done) { set_error("export: bad argument"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(r->rig.device));
    HIPCHK(hipMemcpyAsync(dst_dev, s->d_desc, (size_t)nimg * r->rig.geom.kcap * 32, hipMemcpyDeviceToDevice, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    if (counts_host) for (int m = 0; m < nimg; m++) counts_host[m] = s->h_nsel[m];
    return MCORB_OK;
}

// ---------------------------------------------------------------------------
// single-camera extractor
// ---------------------------------------------------------------------------
int mcorb_create(const mcorb_params *p, int max_width, int max_height, mcorb_t **out)
{
    if (!p || !out) { set_error("null argument"); return MCORB_E_ARG; }
    *out = nullptr;
    Tables t;
    int st = compute_tables(*p, t);
    if (st != MCORB_OK) return st;
    mcorb_t *e = new (std::nothrow) mcorb_extractor;
    if (!e) return MCORB_E_ARG;
    e->params = *p;
    if (max_width > 0 && max_height > 0) {
        e->rig = new Rig;
        st = e->rig->init(*p, 1, max_width, max_height, 1, 1);
        if (st != MCORB_OK) {
            const std::string keep = get_error();
            delete e->rig;
            delete e;
            set_error(keep);
            return st;
        }
        e->w = max_width; e->h = max_height;
    } else if (mcorb_device_count() < 1) {
        delete e;
        set_error("no usable gfx950 device (libmcorb has no CPU path)");
        return MCORB_E_NODEVICE;
    }
    *out = e;
    return MCORB_OK;
}

static void free_knn_scratch(mcorb_t *e)
{
    if (e->d_desc) (void)hipFree(e->d_desc);
    if (e->d_part) (void)hipFree(e->d_part);
    if (e->d_exp) (void)hipFree(e->d_exp);
    if (e->d_lcounts) (void)hipFree(e->d_lcounts);
    e->d_exp = nullptr; e->d_lcounts = nullptr;
    if (e->h_rows) (void)hipHostFree(e->h_rows);
    if (e->h_mlist) (void)hipHostFree(e->h_mlist);
    if (e->h_mcount) (void)hipHostFree(e->h_mcount);
    e->h_mlist = nullptr; e->h_mcount = nullptr;
    if (e->h_counts) (void)hipHostFree(e->h_counts);
    if (e->h_pair) (void)hipHostFree(e->h_pair);
    e->d_desc = nullptr; e->d_part = nullptr; e->h_rows = nullptr; e->h_counts = nullptr; e->h_pair = nullptr;
    e->kc = 0;
}

void mcorb_destroy(mcorb_t *e)
{
    if (!e) return;
    free_knn_scratch(e);
    delete e->rig;
    delete e;
}

static int ensure_rig(mcorb_t *e, int w, int h)
{
    if (e->rig && e->w == w && e->h == h) return MCORB_OK;
    delete e->rig;
    e->rig = new Rig;
    const int st = e->rig->init(e->params, 1, w, h, 1, 1);
    if (st != MCORB_OK) {
        const std::string keep = get_error();
        delete e->rig;
        e->rig = nullptr;
        set_error(keep);
        return st;
    }
    e->w = w; e->h = h;
    return MCORB_OK;
}

static int finish_extract(mcorb_t *e, int lap_x0, int lap_x1, mcorb_keypoint *kps, uint8_t *desc, int cap, int *n_out,
                          int *mono_index_out)
{
    Job j;
    j.kind = Job::EXTRACT; j.nimg = 1; j.lap0 = lap_x0; j.lap1 = lap_x1;
    int st = e->rig->submit(0, j);
    if (st == MCORB_OK) st = e->rig->wait(0);
    if (st != MCORB_OK) return st;
    Slot *s = e->rig->slots[0];
    const int n = (int)s->kps[0].size();
    if (n_out) *n_out = n;
    if (mono_index_out) *mono_index_out = s->mono[0];
    if (n > cap) { set_error("keypoint buffer too small"); return MCORB_E_CAP; }
    if (kps && n) memcpy(kps, s->kps[0].data(), (size_t)n * sizeof(mcorb_keypoint));
    if (desc && n) memcpy(desc, s->h_desc, (size_t)n * 32);
    return MCORB_OK;
}

int mcorb_extract(mcorb_t *e, const uint8_t *gray, int w, int h, int stride_bytes, int lap_x0, int lap_x1,
                  mcorb_keypoint *kps, uint8_t *desc, int cap, int *n_out, int *mono_index_out)
{
    if (!e) { set_error("null extractor"); return MCORB_E_ARG; }
    if (n_out) *n_out = 0;
    if (!gray || w <= 0 || h <= 0) { set_error("empty image"); return MCORB_E_EMPTY; }
    int st = ensure_rig(e, w, h);
    if (st != MCORB_OK) return st;
    const uint8_t *imgs[1] = {gray};
    st = e->rig->upload_u8(0, imgs, 1, stride_bytes);
    if (st != MCORB_OK) return st;
    return finish_extract(e, lap_x0, lap_x1, kps, desc, cap, n_out, mono_index_out);
}

int mcorb_extract_f32(mcorb_t *e, const float *img01, int w, int h, int stride_bytes, int channels, int lap_x0,
                      int lap_x1, mcorb_keypoint *kps, uint8_t *desc, int cap, int *n_out, int *mono_index_out)
{
    if (!e) { set_error("null extractor"); return MCORB_E_ARG; }
    if (n_out) *n_out = 0;
    if (!img01 || w <= 0 || h <= 0) { set_error("empty image"); return MCORB_E_EMPTY; }
    int st = ensure_rig(e, w, h);
    if (st != MCORB_OK) return st;
    const float *imgs[1] = {img01};
    st = e->rig->upload_f32(0, imgs, 1, stride_bytes, channels);
    if (st != MCORB_OK) return st;
    return finish_extract(e, lap_x0, lap_x1, kps, desc, cap, n_out, mono_index_out);
}

int mcorb_get_tables(const mcorb_params *p, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                     int *features_per_level)
{
    if (!p) return MCORB_E_ARG;
    Tables t;
    const int st = compute_tables(*p, t);
    if (st != MCORB_OK) return st;
    for (int i = 0; i < t.nlevels; i++) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.inv_scale[i];
        if (sigma2) sigma2[i] = t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = t.inv_sigma2[i];
        if (features_per_level) features_per_level[i] = t.quota[i];
    }
    return MCORB_OK;
}

int mcorb_get_pyramid_level(mcorb_t *e, int level, uint8_t *dst, int dst_stride, int *w, int *h)
{
    if (!e || !e->rig) { set_error("no image processed yet"); return MCORB_E_STATE; }
    const Geom &g = e->rig->geom;
    if (level < 0 || level >= g.nlevels) return MCORB_E_ARG;
    if (w) *w = g.lv[level].w;
    if (h) *h = g.lv[level].h;
    if (!dst) return MCORB_OK;
    if (dst_stride < g.lv[level].w) return MCORB_E_ARG;
    Slot *s = e->rig->slots[0];
    HIPCHK(hipSetDevice(e->rig->device));
    HIPCHK(hipStreamSynchronize(s->st));
    HIPCHK(hipMemcpy2D(dst, dst_stride, s->d_pyr + g.lv[level].off, g.lv[level].pitch, g.lv[level].w, g.lv[level].h,
                       hipMemcpyDeviceToHost));
    return MCORB_OK;
}

// ORBextractor::DescriptorDistance (ORBextractor.cpp:1202-1218)
int mcorb_hamming256(const uint8_t a[32], const uint8_t b[32])
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}

// MultiCameraFrame::computeRepresentativeDesc (MultiCameraFrame.cpp:530-567): among n (<= 16) descriptors
// of one track, the one with the least median Hamming distance to the others; first minimum wins.
int mcorb_representative_desc(const uint8_t *descs, int n)
{
    if (!descs || n < 1 || n > 64) { set_error("representative_desc: bad argument"); return MCORB_E_ARG; }
    int best_median = 0x7fffffff, best_idx = 0;
    std::vector<int> row(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) row[j] = i == j ? 0 : mcorb_hamming256(descs + (size_t)i * 32, descs + (size_t)j * 32);
        std::sort(row.begin(), row.end());
        const int median = row[(size_t)(0.5 * (n - 1))];
        if (median < best_median) { best_median = median; best_idx = i; }
    }
    return best_idx;
}

static int knn2_host_arrays(mcorb_t *e, const uint8_t *q, int nq, const uint8_t *t, int nt, float thr, float ratio)
{
    if (!e || nq < 0 || nt < 0 || (nq && !q) || (nt && !t)) { set_error("knn2: bad argument"); return MCORB_E_ARG; }
    if (nq > 65535 || nt > 65535) { set_error("knn2: more than 65535 descriptors"); return MCORB_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || e->params.device_id >= ndev) {
        set_error("no usable HIP device (libmcorb has no CPU path)");
        return MCORB_E_NODEVICE;
    }
    HIPCHK(hipSetDevice(e->params.device_id));
    const int need = (std::max(std::max(nq, nt), 1) + 63) / 64 * 64;
    if (need > e->kc) {
        free_knn_scratch(e);
        HIPCHK(hipMalloc((void **)&e->d_desc, (size_t)2 * need * 32));
        HIPCHK(hipMalloc((void **)&e->d_part, knn_part_entries(1, need) * sizeof(uint2)));
        HIPCHK(hipMalloc((void **)&e->d_exp, (size_t)2 * need * kKnnExpandBytes));
        HIPCHK(hipMalloc((void **)&e->d_lcounts, 2 * sizeof(int)));
        HIPCHK(hipHostMalloc((void **)&e->h_rows, (size_t)need * sizeof(KnnRow), hipHostMallocMapped));
        HIPCHK(hipHostMalloc((void **)&e->h_mlist, knn_mlist_stride(need) * sizeof(uint32_t), hipHostMallocMapped));
        HIPCHK(hipHostMalloc((void **)&e->h_mcount, (size_t)knn_qblocks(need) * sizeof(int), hipHostMallocMapped));
        HIPCHK(hipHostMalloc((void **)&e->h_counts, 2 * sizeof(int), hipHostMallocMapped));
        HIPCHK(hipHostMalloc((void **)&e->h_pair, sizeof(int2), hipHostMallocMapped));
        e->kc = need;
    }
    if (nq) HIPCHK(hipMemcpy(e->d_desc, q, (size_t)nq * 32, hipMemcpyHostToDevice));
    if (nt) HIPCHK(hipMemcpy(e->d_desc + (size_t)e->kc * 32, t, (size_t)nt * 32, hipMemcpyHostToDevice));
    e->h_counts[0] = nq;
    e->h_counts[1] = nt;
    e->h_pair[0] = int2{0, 1};
    launch_knn2(nullptr, e->d_desc, e->h_counts, nullptr, 2, e->h_pair, 1, e->kc, e->d_exp, e->d_lcounts, e->d_part, thr, ratio, e->h_rows,
                e->h_mlist, e->h_mcount, nullptr, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return MCORB_OK;
}

int mcorb_knn2(mcorb_t *e, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist)
{
    if (!idx || !dist) { set_error("null output"); return MCORB_E_ARG; }
    const int st = knn2_host_arrays(e, q, nq, t, nt, 75.f, 0.85f);
    if (st != MCORB_OK) return st;
    decode_rows(e->h_rows, nq, idx, dist);
    return MCORB_OK;
}

int mcorb_match_ratio(mcorb_t *e, const uint8_t *q, int nq, const uint8_t *t, int nt, float dist_thresh, float ratio,
                      uint32_t *idx1, uint32_t *idx2, int cap, int *n_out)
{
    const int st = knn2_host_arrays(e, q, nq, t, nt, dist_thresh, ratio);
    if (st != MCORB_OK) return st;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        const KnnRow &r = e->h_rows[i];
        if (knn_accept(r)) {
            if (n < cap) { idx1[n] = (uint32_t)i; idx2[n] = (uint32_t)knn_idx0(r); }
            n++;
        }
    }
    if (n_out) *n_out = n;
    if (n > cap) { set_error("match buffer too small"); return MCORB_E_CAP; }
    return MCORB_OK;
}

int mcorb_host_select(const uint32_t *packed, int n, int minX, int maxX, int minY, int maxY, int nfeatures_level,
                      int wCell, int hCell, int32_t *out_idx, int cap)
{
    if (n < 0 || (n && !packed) || !out_idx) { set_error("host_select: bad argument"); return MCORB_E_ARG; }
    if (n == 0) return 0;
    const SelectParams P = make_select_params(minX, maxX, minY, maxY, nfeatures_level, wCell, hCell);
    if (P.nIni < 1) { set_error("host_select: level too tall"); return MCORB_E_SIZE; }
    static thread_local SelectScratch sc;
    std::vector<uint32_t> sorted;
    std::vector<int> perm, bstart;
    std::vector<mcorb::BucketBest> bbest;
    host_bucket_sort(packed, n, P, sorted, perm, bstart, bbest);   // what k_compact does on the device
    std::vector<int> out((size_t)std::max(nfeatures_level, 0) + 64 + 8);
    std::vector<uint32_t> outv(out.size());
    const int r = select_octree(sorted.data(), bstart.data(), bbest.data(), n, P, out.data(), outv.data(), sc);
    if (r == -4) { set_error("host_select: 2^20 candidates or a level 4096 px wide: beyond the packed (count, UL.x) sort key"); return MCORB_E_SIZE; }
    if (r < 0) { set_error("host_select: level too tall"); return MCORB_E_SIZE; }
    if (r > cap) { set_error("host_select: output too small"); return MCORB_E_CAP; }
    for (int i = 0; i < r; i++) {
        out_idx[i] = perm[out[i]];
        if (outv[i] != packed[out_idx[i]]) { set_error("host_select: value/index mismatch"); return MCORB_E_STATE; }
    }
    return r;
}

int mcorb_host_resize_axis(int ssize, int dsize, int is_x, int32_t *quads)
{
    if (ssize < 1 || dsize < 1 || !quads) return MCORB_E_ARG;
    std::vector<ResizeTap> t;
    build_resize_axis(ssize, dsize, is_x != 0, t, 1);
    for (int d = 0; d < dsize; d++) {
        quads[4 * d] = t[d].s0; quads[4 * d + 1] = t[d].s1; quads[4 * d + 2] = t[d].c0; quads[4 * d + 3] = t[d].c1;
    }
    return MCORB_OK;
}

int mcorb_host_geometry(const mcorb_params *p, int w, int h, int32_t *six)
{
    if (!p || !six) return MCORB_E_ARG;
    Tables t;
    int st = compute_tables(*p, t);
    if (st != MCORB_OK) return st;
    Geom g;
    std::vector<ResizeTap> taps;
    st = build_geometry(*p, t, w, h, g, taps);
    if (st != MCORB_OK) return st;
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom &L = g.lv[l];
        int32_t *o = six + 6 * l;
        o[0] = L.w; o[1] = L.h; o[2] = L.nCols; o[3] = L.nRows; o[4] = L.wCell; o[5] = L.hCell;
    }
    return MCORB_OK;
}

}  // extern "C"
