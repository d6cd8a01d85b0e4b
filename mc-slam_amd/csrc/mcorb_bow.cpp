// mcorb_bow.cpp -- DBoW2 vocabulary container + transform (BASELINE config #4).
//
// Replaces, for this path, DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features,
// BowVector&, FeatureVector&, levelsup) as MC-SLAM calls it (MCSlam/src/MultiCameraFrame.cpp:257,
// FrontEnd.cpp:525,929) and loadFromTextFile (FrontEnd.h:137-138).  DBoW2 itself is an un-vendored
// dependency of the reference; the algorithm restated here is the published one:
// the tree descent runs on the GPU (k_bow_descend), the std::map-ordered accumulation of word
// weights / feature lists and the normalisation run on the host in feature order, because the
// floating-point sums are order-defined.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <chrono>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "mcorb_engine.h"

using namespace mcorb;

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_));                 \
            return MCORB_E_HIP;                                                        \
        }                                                                              \
    } while (0)

struct mcorb_vocab {
    int k = 0, L = 0, scoring = 0, weighting = 0, device = 0;
    int nnodes = 0;                              // including the root (node 0)
    std::vector<int> parent, word_id;            // word_id = -1 for inner nodes
    std::vector<double> weight;
    std::vector<int> child_start, child_count;   // into the flattened children arrays
    int nwords = 0;
    // device copies
    int *d_child_start = nullptr, *d_child_count = nullptr, *d_child_id = nullptr;
    uint8_t *d_child_desc = nullptr;
    // scratch for host-array transforms
    uint8_t *d_desc = nullptr;
    int *d_word_id = nullptr; double *d_weight = nullptr;
    mcorb::BowRes *d_out = nullptr, *h_out = nullptr;
    int cap = 0;
    // scratch of mcorb_rig_match_bow (grow-only): device index tables + result table, pinned host mirror
    int *d_mi = nullptr; float *d_my = nullptr; int2 *d_mrg = nullptr; int4 *d_mtab = nullptr, *h_mtab = nullptr;
    size_t mi_cap = 0, my_cap = 0, mrg_cap = 0, mtab_cap = 0;
};

static void free_vocab(mcorb_vocab *v)
{
    if (!v) return;
    (void)hipFree(v->d_child_start); (void)hipFree(v->d_child_count); (void)hipFree(v->d_child_id);
    (void)hipFree(v->d_child_desc); (void)hipFree(v->d_desc); (void)hipFree(v->d_out); (void)hipFree(v->d_word_id); (void)hipFree(v->d_weight);
    (void)hipFree(v->d_mi); (void)hipFree(v->d_my); (void)hipFree(v->d_mrg); (void)hipFree(v->d_mtab);
    if (v->h_mtab) (void)hipHostFree(v->h_mtab);
    if (v->h_out) (void)hipHostFree(v->h_out);
    delete v;
}

static int build_vocab(int k, int L, int scoring, int weighting, const int32_t *parent, const uint8_t *is_leaf,
                       const uint8_t *desc, const double *weight, int n, int device, mcorb_vocab **out)
{
    if (k < 1 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3 || n < 1 ||
        !parent || !is_leaf || !desc || !weight || !out) {
        set_error("vocabulary: bad header or null argument");   // loadFromTextFile's own range check
        return MCORB_E_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("no usable HIP device (libmcorb has no CPU path)");
        return MCORB_E_NODEVICE;
    }
    mcorb_vocab *v = new mcorb_vocab;
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->device = device;
    v->nnodes = n + 1;
    v->parent.assign(n + 1, -1);
    v->word_id.assign(n + 1, -1);
    v->weight.assign(n + 1, 0.0);
    std::vector<std::vector<int>> children(n + 1);
    for (int i = 0; i < n; i++) {
        const int nid = i + 1, pid = parent[i];
        if (pid < 0 || pid >= nid) { delete v; set_error("vocabulary: parent id must precede the node"); return MCORB_E_ARG; }
        v->parent[nid] = pid;
        children[pid].push_back(nid);                 // m_nodes[pid].children.push_back(nid): file order
        v->weight[nid] = weight[i];
        if (is_leaf[i]) v->word_id[nid] = v->nwords++;   // words are numbered in file order
    }
    for (int nid = 1; nid <= n; nid++)
        if ((v->word_id[nid] >= 0) != children[nid].empty()) {
            delete v;
            set_error("vocabulary: leaf flag disagrees with the tree (a leaf with children or an inner node without)");
            return MCORB_E_ARG;
        }
    if (children[0].empty()) { delete v; set_error("vocabulary: root has no children"); return MCORB_E_ARG; }
    v->child_start.assign(n + 1, 0);
    v->child_count.assign(n + 1, 0);
    std::vector<int> child_id;
    std::vector<uint8_t> child_desc;
    child_id.reserve(n);
    child_desc.reserve((size_t)n * 32);
    for (int nid = 0; nid <= n; nid++) {
        v->child_start[nid] = (int)child_id.size();
        v->child_count[nid] = (int)children[nid].size();
        for (int c : children[nid]) {
            child_id.push_back(c);
            child_desc.insert(child_desc.end(), desc + (size_t)(c - 1) * 32, desc + (size_t)c * 32);
        }
    }
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc((void **)&v->d_child_start, (n + 1) * sizeof(int)));
    HIPCHK(hipMalloc((void **)&v->d_child_count, (n + 1) * sizeof(int)));
    HIPCHK(hipMalloc((void **)&v->d_child_id, child_id.size() * sizeof(int)));
    HIPCHK(hipMalloc((void **)&v->d_child_desc, child_desc.size()));
    HIPCHK(hipMemcpy(v->d_child_start, v->child_start.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(v->d_child_count, v->child_count.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(v->d_child_id, child_id.data(), child_id.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(v->d_child_desc, child_desc.data(), child_desc.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&v->d_word_id, (size_t)(n + 1) * sizeof(int)));
    HIPCHK(hipMalloc((void **)&v->d_weight, (size_t)(n + 1) * sizeof(double)));
    HIPCHK(hipMemcpy(v->d_word_id, v->word_id.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(v->d_weight, v->weight.data(), (size_t)(n + 1) * sizeof(double), hipMemcpyHostToDevice));
    *out = v;
    return MCORB_OK;
}

// BowVector::addWeight / addIfNotExist / normalize and FeatureVector::addFeature.  DBoW2 keeps both in std::maps;
// the BowVector here is the equivalent sorted (word id, value) list: entries are gathered in feature order, stably
// sorted by id and folded left to right, so every sum adds its terms in the order addWeight would have (bit-identical
// doubles) and addIfNotExist keeps the first.
typedef std::vector<std::pair<uint32_t, double>> BowList;
static void assemble(const mcorb_vocab *v, const mcorb::BowRes *res, int n, BowList &bow,
                     std::map<uint32_t, std::vector<int32_t>> &fv)
{
    const bool tf_like = v->weighting == 0 || v->weighting == 1;   // TF_IDF, TF
    BowList raw;
    raw.reserve(n);
    for (int i = 0; i < n; i++) {
        const double w = res[i].weight;   // looked up on the device: the host never touches the 1 M-entry tables
        if (w > 0) {   // not stopped
            raw.emplace_back((uint32_t)res[i].word, w);
            fv[(uint32_t)res[i].nodeup].push_back(i);
        }
    }
    std::stable_sort(raw.begin(), raw.end(), [](const std::pair<uint32_t, double> &a, const std::pair<uint32_t, double> &b) { return a.first < b.first; });
    bow.clear();
    bow.reserve(raw.size());
    for (const auto &e : raw) {
        if (!bow.empty() && bow.back().first == e.first) {
            if (tf_like) bow.back().second += e.second;   // addWeight on an existing word
        } else {
            bow.push_back(e);                             // addWeight / addIfNotExist on a new one
        }
    }
    // mustNormalize: L1_NORM, CHI_SQUARE, KL, BHATTACHARYYA -> L1; L2_NORM -> L2; DOT_PRODUCT -> none
    const bool must = v->scoring != 5;
    if (tf_like && !bow.empty() && !must) {
        const double nd = (double)bow.size();
        for (auto &e : bow) e.second /= nd;
    }
    if (must) {
        double norm = 0.0;
        if (v->scoring == 1) {
            for (auto &e : bow) norm += e.second * e.second;
            norm = sqrt(norm);
        } else {
            for (auto &e : bow) norm += fabs(e.second);
        }
        if (norm > 0.0)
            for (auto &e : bow) e.second /= norm;
    }
}

static int emit(const BowList &bow, const std::map<uint32_t, std::vector<int32_t>> &fv, uint32_t *bow_ids,
                double *bow_vals, int bow_cap, int *nbow, uint32_t *fv_nodes, int32_t *fv_offsets, int fv_cap, int *nfv,
                int32_t *fv_feats, int feat_cap)
{
    if (nbow) *nbow = (int)bow.size();
    if (nfv) *nfv = (int)fv.size();
    size_t nf = 0;
    for (auto &e : fv) nf += e.second.size();
    if ((int)bow.size() > bow_cap || (int)fv.size() > fv_cap || (int)nf > feat_cap) { set_error("transform: output too small"); return MCORB_E_CAP; }
    int i = 0;
    for (auto &e : bow) { bow_ids[i] = e.first; bow_vals[i] = e.second; i++; }
    i = 0;
    int off = 0;
    for (auto &e : fv) {
        fv_nodes[i] = e.first;
        fv_offsets[i] = off;
        for (int32_t f : e.second) fv_feats[off++] = f;
        i++;
    }
    fv_offsets[i] = off;
    return MCORB_OK;
}

static int ensure_scratch(mcorb_vocab *v, int n)
{
    if (n <= v->cap) return MCORB_OK;
    (void)hipFree(v->d_desc); (void)hipFree(v->d_out);
    if (v->h_out) (void)hipHostFree(v->h_out);
    v->d_desc = nullptr; v->d_out = nullptr; v->h_out = nullptr; v->cap = 0;
    const int cap = (n + 1023) / 1024 * 1024;
    HIPCHK(hipMalloc((void **)&v->d_desc, (size_t)cap * 32));
    HIPCHK(hipMalloc((void **)&v->d_out, (size_t)cap * sizeof(mcorb::BowRes)));
    HIPCHK(hipHostMalloc((void **)&v->h_out, (size_t)cap * sizeof(mcorb::BowRes), hipHostMallocDefault));
    v->cap = cap;
    return MCORB_OK;
}

extern "C" {

int mcorb_vocab_create(int k, int L, int scoring, int weighting, const int32_t *parent, const uint8_t *is_leaf,
                       const uint8_t *desc, const double *weight, int nnodes, int device, mcorb_vocab **out)
{
    if (out) *out = nullptr;
    return build_vocab(k, L, scoring, weighting, parent, is_leaf, desc, weight, nnodes, device, out);
}

// TemplatedVocabulary::loadFromTextFile (ORB-SLAM's DBoW2 fork; the call at FrontEnd.h:137-138):
// first line "k L scoring weighting", then one node per line: "parent isLeaf d0 .. d31 weight".
int mcorb_vocab_load_text(const char *path, int device, mcorb_vocab **out)
{
    if (out) *out = nullptr;
    if (!path || !out) { set_error("null argument"); return MCORB_E_ARG; }
    std::ifstream f(path);
    if (!f.is_open()) { set_error(std::string("cannot open vocabulary file ") + path); return MCORB_E_ARG; }
    std::string s;
    std::getline(f, s);
    std::stringstream ss(s);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    ss >> k >> L >> n1 >> n2;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        set_error("Vocabulary loading failure: This is not a correct text file!");
        return MCORB_E_ARG;
    }
    std::vector<int32_t> parent;
    std::vector<uint8_t> leaf, desc;
    std::vector<double> weight;
    while (std::getline(f, s)) {
        if (s.empty()) continue;
        std::stringstream sn(s);
        int pid = 0, isleaf = 0;
        sn >> pid >> isleaf;
        parent.push_back(pid);
        leaf.push_back(isleaf > 0);
        for (int i = 0; i < 32; i++) { int b = 0; sn >> b; desc.push_back((uint8_t)b); }
        double w = 0;
        sn >> w;
        weight.push_back(w);
        if (sn.fail()) { set_error("vocabulary: malformed node line"); return MCORB_E_ARG; }
    }
    return build_vocab(k, L, n1, n2, parent.data(), leaf.data(), desc.data(), weight.data(), (int)parent.size(), device, out);
}

void mcorb_vocab_destroy(mcorb_vocab *v) { free_vocab(v); }

int mcorb_vocab_info(const mcorb_vocab *v, int *k, int *L, int *nnodes, int *nwords)
{
    if (!v) return MCORB_E_ARG;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (nnodes) *nnodes = v->nnodes;
    if (nwords) *nwords = v->nwords;
    return MCORB_OK;
}

static int transform_device(mcorb_vocab *v, const uint8_t *d_desc, int n, int levelsup, hipStream_t st, uint32_t *bow_ids,
                            double *bow_vals, int bow_cap, int *nbow, uint32_t *fv_nodes, int32_t *fv_offsets, int fv_cap,
                            int *nfv, int32_t *fv_feats, int feat_cap)
{
    BowList bow;
    std::map<uint32_t, std::vector<int32_t>> fv;
    if (n > 0) {
        const int nid_level = v->L - levelsup;   // <= 0: the feature vector is keyed by the root (node 0)
        launch_bow_descend(st, d_desc, n, v->d_child_start, v->d_child_count, v->d_child_desc, v->d_child_id, v->d_word_id, v->d_weight, nid_level, v->d_out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(v->h_out, v->d_out, (size_t)n * sizeof(mcorb::BowRes), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        assemble(v, v->h_out, n, bow, fv);
    }
    return emit(bow, fv, bow_ids, bow_vals, bow_cap, nbow, fv_nodes, fv_offsets, fv_cap, nfv, fv_feats, feat_cap);
}

int mcorb_vocab_transform(mcorb_vocab *v, const uint8_t *desc, int n, int levelsup, uint32_t *bow_ids, double *bow_vals,
                          int bow_cap, int *nbow, uint32_t *fv_nodes, int32_t *fv_offsets, int fv_cap, int *nfv,
                          int32_t *fv_feats, int feat_cap)
{
    if (!v || n < 0 || (n && !desc)) { set_error("transform: bad argument"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(v->device));
    int st = ensure_scratch(v, std::max(n, 1));
    if (st != MCORB_OK) return st;
    if (n) HIPCHK(hipMemcpy(v->d_desc, desc, (size_t)n * 32, hipMemcpyHostToDevice));
    return transform_device(v, v->d_desc, n, levelsup, nullptr, bow_ids, bow_vals, bow_cap, nbow, fv_nodes, fv_offsets, fv_cap, nfv,
                            fv_feats, feat_cap);
}

// declared in mcorb_api.cpp's translation unit through the header; needs the rig internals
int mcorb_rig_transform_image(mcorb_rig *r, int slot, int m, mcorb_vocab *v, int levelsup, uint32_t *bow_ids, double *bow_vals,
                              int bow_cap, int *nbow, uint32_t *fv_nodes, int32_t *fv_offsets, int fv_cap, int *nfv,
                              int32_t *fv_feats, int feat_cap);

}  // extern "C"

extern "C" int mcorb_rig_transform_image(mcorb_rig *r, int slot, int m, mcorb_vocab *v, int levelsup, uint32_t *bow_ids,
                                         double *bow_vals, int bow_cap, int *nbow, uint32_t *fv_nodes, int32_t *fv_offsets,
                                         int fv_cap, int *nfv, int32_t *fv_feats, int feat_cap)
{
    if (!r || !v || slot < 0 || slot >= (int)r->rig.slots.size()) { set_error("rig transform: bad argument"); return MCORB_E_ARG; }
    Slot *s = r->rig.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    if (m < 0 || m >= s->nimg_done) { set_error("image index out of range"); return MCORB_E_ARG; }
    if (v->device != r->rig.device) { set_error("vocabulary lives on another device"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(v->device));
    const int n = s->h_nsel[m];
    int st = ensure_scratch(v, std::max(n, 1));
    if (st != MCORB_OK) return st;
    // descriptors of image m are still resident in the slot: no host round trip
    return transform_device(v, s->d_desc + (size_t)m * r->rig.geom.kcap * 32, n, levelsup, s->st, bow_ids, bow_vals, bow_cap, nbow,
                            fv_nodes, fv_offsets, fv_cap, nfv, fv_feats, feat_cap);
}

// ---------------------------------------------------------------------------
// computeIntraMatches(matches, words_): the BoW-guided live variant (MultiCameraFrame.cpp:586-943).
// GPU: vocabulary descent of every camera's descriptors + the per-node best/second-best table
// (k_bow_best2).  Host: the reference's serial iteration over common nodes and its track
// bookkeeping (new track / extend / merge into an existing one / override), replayed on that table.
// ---------------------------------------------------------------------------
namespace {
struct BowTrack { int matchIndex[MCORB_MAX_CAMS]; int n_rays; };
}

extern "C" int mcorb_rig_match_bow(mcorb_rig *r, int slot, int frame, mcorb_vocab *v, int levelsup, double max_neighbor_ratio,
                                   int32_t *tracks, int32_t *n_rays, int cap_tracks, int *ntracks_out, uint32_t *words,
                                   int cap_words, int *nwords_out)
{
    if (ntracks_out) *ntracks_out = 0;
    if (nwords_out) *nwords_out = 0;
    if (!r || !v || slot < 0 || slot >= (int)r->rig.slots.size() || !tracks) { set_error("match_bow: bad argument"); return MCORB_E_ARG; }
    Rig &R = r->rig;
    Slot *s = R.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    const int C = R.ncams, kcap = R.geom.kcap;
    if (frame < 0 || (frame + 1) * C > s->nimg_done) { set_error("match_bow: frame not extracted"); return MCORB_E_STATE; }
    if (v->device != R.device) { set_error("vocabulary lives on another device"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(R.device));
    const int TH_LOW = 75;   // ORBextractor.h:27

    static const bool prof = getenv("MCORB_HOST_PROF") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto T0 = now();
    auto T1 = T0, T2 = T0, T3 = T0, T4 = T0;
    // 1. FeatureVector of every camera (transform(..., levelsup), MultiCameraFrame.cpp:257): all descents are
    //    queued back to back, one read-back, one synchronisation
    std::vector<std::map<uint32_t, std::vector<int32_t>>> fvs(C);
    std::vector<int> nfeat(C);
    {
        int st = ensure_scratch(v, C * kcap);
        if (st != MCORB_OK) return st;
        for (int c = 0; c < C; c++) {
            const int m = frame * C + c, n = s->h_nsel[m];
            nfeat[c] = n;
            if (n > 0)
                launch_bow_descend(s->st, s->d_desc + (size_t)m * kcap * 32, n, v->d_child_start, v->d_child_count, v->d_child_desc,
                                   v->d_child_id, v->d_word_id, v->d_weight, v->L - levelsup, v->d_out + (size_t)c * kcap);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(v->h_out, v->d_out, (size_t)C * kcap * sizeof(mcorb::BowRes), hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipStreamSynchronize(s->st));
        T1 = now();
        for (int c = 0; c < C; c++) {
            // FeatureVector only (addFeature for every non-stopped word, in feature order); the BowVector half of
            // transform() is not needed here and its 2000 map insertions would dominate the call
            const mcorb::BowRes *res = v->h_out + (size_t)c * kcap;
            for (int i = 0; i < nfeat[c]; i++)
                if (res[i].weight > 0) fvs[c][(uint32_t)res[i].nodeup].push_back(i);
            if (fvs[c].empty()) return MCORB_OK;   // the reference returns with no matches (:602-603)
        }
    }

    T2 = now();
    // 2. node slots (distinct node ids over all cameras) and per-(slot, camera) feature ranges
    std::map<uint32_t, int> slot_id;
    for (int c = 0; c < C; c++)
        for (auto &e : fvs[c]) slot_id.emplace(e.first, 0);
    int ns = 0;
    for (auto &e : slot_id) e.second = ns++;
    std::vector<int> slot_of((size_t)C * kcap, -1), node_feats((size_t)C * kcap, 0), sets(C);
    std::vector<int2> node_range((size_t)ns * C, int2{0, 0});
    std::vector<float> yv((size_t)C * kcap, 0.f);
    for (int c = 0; c < C; c++) {
        sets[c] = frame * C + c;
        int pos = 0;
        for (auto &e : fvs[c]) {
            const int sl = slot_id[e.first];
            node_range[(size_t)sl * C + c] = int2{pos, (int)e.second.size()};
            for (int32_t f : e.second) { node_feats[(size_t)c * kcap + pos++] = f; slot_of[(size_t)c * kcap + f] = sl; }
        }
        const std::vector<mcorb_keypoint> &K = s->kps[frame * C + c];
        for (size_t k = 0; k < K.size(); k++) yv[(size_t)c * kcap + k] = K[k].y;   // image_kps_undist[c][k].pt.y
    }

    T3 = now();
    // 3. best / second-best table on the GPU (scratch lives in the vocabulary object, grow-only)
    const size_t n_i = (size_t)C * kcap * 2 + C * 2, tab_n = (size_t)C * kcap * C;
    auto grow = [](void **p, size_t &cap, size_t need, size_t elem) -> bool {
        if (need <= cap) return true;
        (void)hipFree(*p);
        *p = nullptr; cap = 0;
        if (hipMalloc(p, need * elem) != hipSuccess) return false;
        cap = need;
        return true;
    };
    if (!grow((void **)&v->d_mi, v->mi_cap, n_i, sizeof(int)) || !grow((void **)&v->d_my, v->my_cap, yv.size(), sizeof(float)) ||
        !grow((void **)&v->d_mrg, v->mrg_cap, std::max<size_t>(node_range.size(), 1), sizeof(int2))) {
        set_error("match_bow: device allocation failed");
        return MCORB_E_HIP;
    }
    if (tab_n > v->mtab_cap) {
        (void)hipFree(v->d_mtab);
        if (v->h_mtab) (void)hipHostFree(v->h_mtab);
        v->d_mtab = nullptr; v->h_mtab = nullptr; v->mtab_cap = 0;
        HIPCHK(hipMalloc((void **)&v->d_mtab, tab_n * sizeof(int4)));
        HIPCHK(hipHostMalloc((void **)&v->h_mtab, tab_n * sizeof(int4), hipHostMallocDefault));
        v->mtab_cap = tab_n;
    }
    int *d_slot_of = v->d_mi, *d_node_feats = v->d_mi + (size_t)C * kcap, *d_sets = v->d_mi + (size_t)C * kcap * 2, *d_nfeat = d_sets + C;
    HIPCHK(hipMemcpyAsync(d_slot_of, slot_of.data(), slot_of.size() * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(d_node_feats, node_feats.data(), node_feats.size() * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(d_sets, sets.data(), C * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(d_nfeat, nfeat.data(), C * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(v->d_my, yv.data(), yv.size() * sizeof(float), hipMemcpyHostToDevice, s->st));
    if (!node_range.empty())
        HIPCHK(hipMemcpyAsync(v->d_mrg, node_range.data(), node_range.size() * sizeof(int2), hipMemcpyHostToDevice, s->st));
    launch_bow_best2(s->st, s->d_desc, d_sets, kcap, C, v->d_my, d_slot_of, v->d_mrg, d_node_feats, d_nfeat, v->d_mtab);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(v->h_mtab, v->d_mtab, tab_n * sizeof(int4), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));   // also covers the pageable host vectors above
    const int4 *tab = v->h_mtab;
    T4 = now();

    // 4. the reference's serial walk over common nodes (:647-929), reading best/second-best from the table
    typedef std::map<uint32_t, std::vector<int32_t>>::const_iterator It;
    std::vector<It> it(C), last(C);
    for (int c = 0; c < C; c++) { it[c] = fvs[c].begin(); last[c] = std::prev(fvs[c].end()); }
    std::vector<BowTrack> matches;
    matches.reserve(10000);   // (:587)
    std::vector<uint32_t> words_;
    int intraMatchInd = 0;
    const uint8_t *hd = s->h_desc;
    auto dist = [&](int ca, int fa, int cb, int fb) {
        return mcorb_hamming256(hd + ((size_t)(frame * C + ca) * kcap + fa) * 32, hd + ((size_t)(frame * C + cb) * kcap + fb) * 32);
    };
    for (;;) {
        bool end = true;   // checkItersEnd (:569-575)
        for (int c = 0; c < C; c++) end = end && it[c]->first >= last[c]->first;
        if (end) break;
        uint32_t min_val = 0x7ffffffeu;
        std::vector<int> selected;
        for (int c = 0; c < C; c++) {
            const uint32_t w = it[c] == last[c] ? 0x7fffffffu : it[c]->first;
            if (w < min_val) { min_val = w; selected.clear(); selected.push_back(c); }
            else if (w == min_val) selected.push_back(c);
        }
        std::vector<std::vector<int>> matchedFlags(selected.size());
        for (size_t i = 0; i < selected.size(); i++) matchedFlags[i].assign(it[selected[i]]->second.size(), -1);
        if (selected.size() >= 2) {
            for (int i = 0; i < (int)selected.size() - 1; i++) {
                const int cam1 = selected[i];
                const std::vector<int32_t> &feat_cam1 = it[cam1]->second;
                for (int a = 0; a < (int)feat_cam1.size(); a++) {
                    bool foundMatch = false;
                    if (matchedFlags[i][a] != -1) continue;
                    BowTrack temp;
                    for (int tt = 0; tt < C; tt++) temp.matchIndex[tt] = -1;
                    temp.matchIndex[cam1] = feat_cam1[a];
                    temp.n_rays = 1;
                    matches.push_back(temp);
                    matchedFlags[i][a] = intraMatchInd;
                    bool updateOnce = true;
                    for (int j = i + 1; j < (int)selected.size(); j++) {
                        const int cam2 = selected[j];
                        const std::vector<int32_t> &feat_cam2 = it[cam2]->second;
                        const int4 t = tab[((size_t)cam1 * kcap + feat_cam1[a]) * C + cam2];
                        const int best_j_now = t.x;
                        const double best_dist_1 = t.x < 0 ? 1e9 : (double)t.y;
                        const double best_dist_2 = t.z == 0x7fffffff ? 1e9 : (double)t.z;
                        if (best_dist_1 <= TH_LOW && best_dist_1 / best_dist_2 <= max_neighbor_ratio) {
                            const int existing = matchedFlags[j][best_j_now];
                            if (existing == intraMatchInd) continue;
                            if (existing == -1) {
                                matches[intraMatchInd].matchIndex[cam2] = feat_cam2[best_j_now];
                                matches[intraMatchInd].n_rays++;
                                matchedFlags[j][best_j_now] = intraMatchInd;
                                foundMatch = true;
                            } else {
                                const int old_cam1 = matches[existing].matchIndex[cam1];
                                if (old_cam1 == -1) {
                                    if (updateOnce) updateOnce = false;
                                    else continue;
                                    bool update_match = true;
                                    int tmp[MCORB_MAX_CAMS];
                                    for (int tt = 0; tt < C; tt++) tmp[tt] = matches[existing].matchIndex[tt];
                                    int inc = 0;
                                    for (int tt = 0; tt < C; tt++) {
                                        if (matches[intraMatchInd].matchIndex[tt] != -1) {
                                            if (matches[existing].matchIndex[tt] != -1) { update_match = false; break; }
                                            tmp[tt] = matches[intraMatchInd].matchIndex[tt];
                                            inc++;
                                        }
                                    }
                                    if (update_match) {
                                        for (int tt = 0; tt < C; tt++) matches[existing].matchIndex[tt] = tmp[tt];
                                        matches[existing].n_rays += inc;
                                        matchedFlags[i][a] = existing;
                                        intraMatchInd = existing;
                                        foundMatch = true;
                                        matches.pop_back();
                                    }
                                    continue;
                                }
                                const double d = dist(cam1, old_cam1, cam2, feat_cam2[best_j_now]);
                                if (best_dist_1 < d) {
                                    matches[existing].matchIndex[cam2] = -1;
                                    matches[existing].n_rays--;
                                    matches[intraMatchInd].matchIndex[cam2] = feat_cam2[best_j_now];
                                    matches[intraMatchInd].n_rays++;
                                    matchedFlags[j][best_j_now] = intraMatchInd;
                                    foundMatch = true;
                                }
                            }
                        }
                    }
                    if (foundMatch) {
                        words_.push_back(it[selected[0]]->first);
                        intraMatchInd = (int)matches.size();
                    } else {
                        matches.pop_back();
                        matchedFlags[i][a] = -1;
                    }
                }
            }
        }
        for (int c : selected) ++it[c];
    }
    if (prof)
        fprintf(stderr, "[mcorb host prof] match_bow: descend+sync %.0f us, assemble %.0f, tables %.0f, best2+copy %.0f, replay %.0f\n",
                us(T0, T1), us(T1, T2), us(T2, T3), us(T3, T4), us(T4, now()));
    if (ntracks_out) *ntracks_out = (int)matches.size();
    if (nwords_out) *nwords_out = (int)words_.size();
    if ((int)matches.size() > cap_tracks || (words && (int)words_.size() > cap_words)) { set_error("match_bow: output too small"); return MCORB_E_CAP; }
    for (size_t m = 0; m < matches.size(); m++) {
        for (int c = 0; c < C; c++) tracks[m * C + c] = matches[m].matchIndex[c];
        if (n_rays) n_rays[m] = matches[m].n_rays;
    }
    if (words) for (size_t w = 0; w < words_.size(); w++) words[w] = words_[w];
    return MCORB_OK;
}
