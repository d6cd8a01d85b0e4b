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
#include <mutex>
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
    // The scratch above belongs to the vocabulary, the launches that use it go to the calling slot's stream: two threads
    // driving two slots with one vocabulary (the reference transforms from per-camera threads) would race on it, and a
    // grow on one would free what the other's kernel still reads.  Every entry point that touches the scratch holds this
    // from ensure_scratch to its final stream synchronisation.
    std::mutex scratch_mu;
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
    // device copies; a failure on the way frees the half-built object (free_vocab tolerates null members)
    auto upload = [&]() -> int {
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
        return MCORB_OK;
    };
    const int st = upload();
    if (st != MCORB_OK) { free_vocab(v); return st; }
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
    std::lock_guard<std::mutex> scratch_lock(v->scratch_mu);
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
    std::lock_guard<std::mutex> scratch_lock(v->scratch_mu);
    int st = ensure_scratch(v, std::max(n, 1));
    if (st != MCORB_OK) return st;
    // descriptors of image m are still resident in the slot: no host round trip
    return transform_device(v, s->d_desc + (size_t)m * r->rig.geom.kcap * 32, n, levelsup, s->st, bow_ids, bow_vals, bow_cap, nbow,
                            fv_nodes, fv_offsets, fv_cap, nfv, fv_feats, feat_cap);
}

// transform() of images [img0, img0 + nimg) of a slot in one go (what extractFeaturesParallel's per-camera threads do,
// MultiCameraFrame.cpp:257): one descent launch, one read-back, the order-defined folds of the images on the worker pool.
extern "C" int mcorb_rig_transform_images(mcorb_rig *r, int slot, int img0, int nimg, mcorb_vocab *v, int levelsup)
{
    if (!r || !v || slot < 0 || slot >= (int)r->rig.slots.size() || nimg < 1) { set_error("rig transform: bad argument"); return MCORB_E_ARG; }
    Rig &R = r->rig;
    Slot *s = R.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    if (img0 < 0 || img0 + nimg > s->nimg_done) { set_error("image index out of range"); return MCORB_E_ARG; }
    if (v->device != R.device) { set_error("vocabulary lives on another device"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(v->device));
    const int kcap = R.geom.kcap;
    std::lock_guard<std::mutex> scratch_lock(v->scratch_mu);
    int st = ensure_scratch(v, nimg * kcap);
    if (st != MCORB_OK) return st;
    launch_bow_descend(s->st, s->d_desc + (size_t)img0 * kcap * 32, nimg * kcap, v->d_child_start, v->d_child_count, v->d_child_desc,
                       v->d_child_id, v->d_word_id, v->d_weight, v->L - levelsup, v->d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(v->h_out, v->d_out, (size_t)nimg * kcap * sizeof(mcorb::BowRes), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    if ((int)s->bowvec.size() < R.max_images) { s->bowvec.resize(R.max_images); s->bowvec_ok.assign(R.max_images, 0); }
    R.pool->parallel_for(nimg, [&](int i, int) {
        BowList bow;
        std::map<uint32_t, std::vector<int32_t>> fv;
        assemble(v, v->h_out + (size_t)i * kcap, s->h_nsel[img0 + i], bow, fv);
        BowImageOut &o = s->bowvec[img0 + i];
        o.bow_ids.clear(); o.bow_vals.clear(); o.fv_nodes.clear(); o.fv_offsets.clear(); o.fv_feats.clear();
        for (auto &e : bow) { o.bow_ids.push_back(e.first); o.bow_vals.push_back(e.second); }
        for (auto &e : fv) {
            o.fv_nodes.push_back(e.first);
            o.fv_offsets.push_back((int32_t)o.fv_feats.size());
            o.fv_feats.insert(o.fv_feats.end(), e.second.begin(), e.second.end());
        }
        o.fv_offsets.push_back((int32_t)o.fv_feats.size());
        s->bowvec_ok[img0 + i] = 1;
    }, R.pool_threads + s->index);
    return MCORB_OK;
}

extern "C" int mcorb_rig_get_transform(mcorb_rig *r, int slot, int m, uint32_t *bow_ids, double *bow_vals, int bow_cap, int *nbow,
                                       uint32_t *fv_nodes, int32_t *fv_offsets, int fv_cap, int *nfv, int32_t *fv_feats, int feat_cap)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) { set_error("get transform: bad argument"); return MCORB_E_ARG; }
    Slot *s = r->rig.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    if (m < 0 || m >= (int)s->bowvec_ok.size() || !s->bowvec_ok[m]) { set_error("get transform: image not transformed since the slot's last extraction"); return MCORB_E_STATE; }
    const BowImageOut &o = s->bowvec[m];
    if (nbow) *nbow = (int)o.bow_ids.size();
    if (nfv) *nfv = (int)o.fv_nodes.size();
    if ((int)o.bow_ids.size() > bow_cap || (int)o.fv_nodes.size() > fv_cap || (int)o.fv_feats.size() > feat_cap) { set_error("transform: output too small"); return MCORB_E_CAP; }
    if (!o.bow_ids.empty()) { memcpy(bow_ids, o.bow_ids.data(), o.bow_ids.size() * 4); memcpy(bow_vals, o.bow_vals.data(), o.bow_vals.size() * 8); }
    if (!o.fv_nodes.empty()) memcpy(fv_nodes, o.fv_nodes.data(), o.fv_nodes.size() * 4);
    memcpy(fv_offsets, o.fv_offsets.data(), o.fv_offsets.size() * 4);
    if (!o.fv_feats.empty()) memcpy(fv_feats, o.fv_feats.data(), o.fv_feats.size() * 4);
    return MCORB_OK;
}

// ---------------------------------------------------------------------------
// computeIntraMatches(matches, words_): the BoW-guided live variant (MultiCameraFrame.cpp:586-943).
// GPU: vocabulary descent of every camera's descriptors + the per-node best/second-best table
// (k_bow_best2), both for ALL frames of the batch in one launch each.  Host: the reference's serial
// iteration over common nodes and its track bookkeeping (new track / extend / merge into an existing one /
// override), replayed on that table, one worker-pool task per frame.
// A FeatureVector is DBoW2's std::map<node id, vector<feature index>>; here it is the equivalent sorted list of
// (node id, run of feature indices in feature order) -- iteration order and contents are the map's.
// ---------------------------------------------------------------------------
namespace {
struct BowTrack { int matchIndex[MCORB_MAX_CAMS]; int n_rays; };
struct FvRun { uint32_t node; int beg, cnt; };          // features fv_feats[beg .. beg + cnt)
struct FrameTables {                                    // per frame, built on the host between the two GPU phases
    std::vector<std::vector<FvRun>> fv;                 // per camera
    std::vector<std::vector<int32_t>> feats;            // per camera: feature indices, node-major
    std::vector<uint32_t> slots;                        // distinct node ids over all cameras, ascending
    bool empty = false;                                 // some camera has no feature vector: the reference returns no matches
};
}  // namespace

static int pair_of(int C, int c1, int c2) { return c1 * C - c1 * (c1 + 1) / 2 + (c2 - c1 - 1); }

extern "C" int mcorb_rig_match_bow_frames(mcorb_rig *r, int slot, int frame0, int nframes, mcorb_vocab *v, int levelsup,
                                          double max_neighbor_ratio, const float *const *y_undist)
{
    if (!r || !v || slot < 0 || slot >= (int)r->rig.slots.size() || nframes < 1) { set_error("match_bow: bad argument"); return MCORB_E_ARG; }
    Rig &R = r->rig;
    Slot *s = R.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    const int C = R.ncams, kcap = R.geom.kcap, npairs = C * (C - 1) / 2;
    if (frame0 < 0 || (frame0 + nframes) * C > s->nimg_done) { set_error("match_bow: frames not extracted"); return MCORB_E_STATE; }
    if (v->device != R.device) { set_error("vocabulary lives on another device"); return MCORB_E_ARG; }
    HIPCHK(hipSetDevice(R.device));
    const int TH_LOW = 75;   // ORBextractor.h:27
    const int img0 = frame0 * C, nimg = nframes * C;
    if ((int)s->bow.size() < R.max_frames) { s->bow.resize(R.max_frames); s->bow_ok.assign(R.max_frames, 0); }

    static const bool prof = getenv("MCORB_HOST_PROF") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto T0 = now();

    // 1. vocabulary descent of every descriptor of the batch (transform(..., levelsup), MultiCameraFrame.cpp:257): one
    //    launch over the slot's kcap-strided descriptor block (rows past an image's count are descended too and ignored)
    std::lock_guard<std::mutex> scratch_lock(v->scratch_mu);
    int st = ensure_scratch(v, nimg * kcap);
    if (st != MCORB_OK) return st;
    launch_bow_descend(s->st, s->d_desc + (size_t)img0 * kcap * 32, nimg * kcap, v->d_child_start, v->d_child_count, v->d_child_desc,
                       v->d_child_id, v->d_word_id, v->d_weight, v->L - levelsup, v->d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(v->h_out, v->d_out, (size_t)nimg * kcap * sizeof(mcorb::BowRes), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));
    const auto T1 = now();

    // 2. per frame: FeatureVectors (addFeature for every non-stopped word, in feature order; the BowVector half of
    //    transform() is not needed here), node slots and the index tables of k_bow_best2
    std::vector<FrameTables> ft(nframes);
    std::vector<int> h_slot_of((size_t)nimg * kcap, -1), h_node_feats((size_t)nimg * kcap, 0), h_nfeat(nimg, 0);
    std::vector<float> h_yv((size_t)nimg * kcap, 0.f);
    R.pool->parallel_for(nframes, [&](int f, int) {
        FrameTables &F = ft[f];
        F.fv.resize(C); F.feats.resize(C);
        std::vector<std::pair<uint32_t, int32_t>> tmp;
        for (int c = 0; c < C; c++) {
            const int m = f * C + c, n = s->h_nsel[img0 + m];
            h_nfeat[m] = n;
            const mcorb::BowRes *res = v->h_out + (size_t)m * kcap;
            tmp.clear();
            for (int i = 0; i < n; i++)
                if (res[i].weight > 0) tmp.emplace_back((uint32_t)res[i].nodeup, i);
            std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<uint32_t, int32_t> &a, const std::pair<uint32_t, int32_t> &b) { return a.first < b.first; });
            F.feats[c].resize(tmp.size());
            for (size_t i = 0; i < tmp.size(); i++) {
                F.feats[c][i] = tmp[i].second;
                if (i == 0 || tmp[i].first != tmp[i - 1].first) F.fv[c].push_back(FvRun{tmp[i].first, (int)i, 0});
                F.fv[c].back().cnt++;
            }
            if (F.fv[c].empty()) F.empty = true;   // the reference returns with no matches (:602-603)
            // image_kps_undist[c][k].pt.y (:708-716): the caller's undistorted rows, or the raw ones (RECTIFY / zero distortion)
            const std::vector<mcorb_keypoint> &K = s->kps[img0 + m];
            const float *yu = y_undist ? y_undist[img0 + m] : nullptr;
            for (int k = 0; k < n; k++) h_yv[(size_t)m * kcap + k] = yu ? yu[k] : K[k].y;
        }
        for (int c = 0; c < C; c++)
            for (const FvRun &e : F.fv[c]) F.slots.push_back(e.node);
        std::sort(F.slots.begin(), F.slots.end());
        F.slots.erase(std::unique(F.slots.begin(), F.slots.end()), F.slots.end());
        for (int c = 0; c < C; c++) {
            const int m = f * C + c;
            for (const FvRun &e : F.fv[c]) {
                const int sl = (int)(std::lower_bound(F.slots.begin(), F.slots.end(), e.node) - F.slots.begin());
                for (int k = 0; k < e.cnt; k++) {
                    h_node_feats[(size_t)m * kcap + e.beg + k] = F.feats[c][e.beg + k];
                    h_slot_of[(size_t)m * kcap + F.feats[c][e.beg + k]] = sl;
                }
            }
        }
    }, R.pool_threads + s->index);
    std::vector<int> h_rgbase(nframes + 1, 0);
    for (int f = 0; f < nframes; f++) h_rgbase[f + 1] = h_rgbase[f] + (int)ft[f].slots.size();
    std::vector<int2> h_rg((size_t)std::max(h_rgbase[nframes], 1) * C, int2{0, 0});
    for (int f = 0; f < nframes; f++)
        for (int c = 0; c < C; c++)
            for (const FvRun &e : ft[f].fv[c]) {
                const int sl = (int)(std::lower_bound(ft[f].slots.begin(), ft[f].slots.end(), e.node) - ft[f].slots.begin());
                h_rg[(size_t)(h_rgbase[f] + sl) * C + c] = int2{e.beg, e.cnt};
            }
    const auto T2 = now();

    // 3. best / second-best tables on the GPU (scratch lives in the vocabulary object, grow-only)
    const size_t n_i = (size_t)nimg * kcap * 2 + nimg + nframes + 1, tab_n = (size_t)nframes * npairs * kcap;
    auto grow = [](void **p, size_t &cap, size_t need, size_t elem) -> bool {
        if (need <= cap) return true;
        (void)hipFree(*p);
        *p = nullptr; cap = 0;
        if (hipMalloc(p, need * elem) != hipSuccess) return false;
        cap = need;
        return true;
    };
    if (!grow((void **)&v->d_mi, v->mi_cap, n_i, sizeof(int)) || !grow((void **)&v->d_my, v->my_cap, h_yv.size(), sizeof(float)) ||
        !grow((void **)&v->d_mrg, v->mrg_cap, h_rg.size(), sizeof(int2))) {
        set_error("match_bow: device allocation failed");
        return MCORB_E_HIP;
    }
    if (std::max<size_t>(tab_n, 1) > v->mtab_cap) {
        (void)hipFree(v->d_mtab);
        if (v->h_mtab) (void)hipHostFree(v->h_mtab);
        v->d_mtab = nullptr; v->h_mtab = nullptr; v->mtab_cap = 0;
        HIPCHK(hipMalloc((void **)&v->d_mtab, std::max<size_t>(tab_n, 1) * sizeof(int4)));
        HIPCHK(hipHostMalloc((void **)&v->h_mtab, std::max<size_t>(tab_n, 1) * sizeof(int4), hipHostMallocDefault));
        v->mtab_cap = std::max<size_t>(tab_n, 1);
    }
    int *d_slot_of = v->d_mi, *d_node_feats = d_slot_of + (size_t)nimg * kcap, *d_nfeat = d_node_feats + (size_t)nimg * kcap, *d_rgbase = d_nfeat + nimg;
    HIPCHK(hipMemcpyAsync(d_slot_of, h_slot_of.data(), h_slot_of.size() * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(d_node_feats, h_node_feats.data(), h_node_feats.size() * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(d_nfeat, h_nfeat.data(), nimg * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(d_rgbase, h_rgbase.data(), (nframes + 1) * sizeof(int), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(v->d_my, h_yv.data(), h_yv.size() * sizeof(float), hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(v->d_mrg, h_rg.data(), h_rg.size() * sizeof(int2), hipMemcpyHostToDevice, s->st));
    launch_bow_best2(s->st, s->d_desc, img0, kcap, C, nframes, v->d_my, d_slot_of, v->d_mrg, d_rgbase, d_node_feats, d_nfeat, v->d_mtab);
    HIPCHK(hipGetLastError());
    if (tab_n) HIPCHK(hipMemcpyAsync(v->h_mtab, v->d_mtab, tab_n * sizeof(int4), hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipStreamSynchronize(s->st));   // also covers the pageable host vectors above
    const auto T3 = now();

    // 4. per frame: the reference's serial walk over common nodes (:647-929), reading best/second-best from the table
    const uint8_t *hd = s->h_desc;
    R.pool->parallel_for(nframes, [&](int f, int) {
        BowFrameOut &out = s->bow[frame0 + f];
        out.tracks.clear(); out.n_rays.clear(); out.words.clear();
        const FrameTables &F = ft[f];
        if (F.empty) return;
        const int4 *tab = v->h_mtab + (size_t)f * npairs * kcap;
        std::vector<int> it(C, 0), last(C);
        for (int c = 0; c < C; c++) last[c] = (int)F.fv[c].size() - 1;   // std::prev(end()): the last node is never visited (reference quirk)
        std::vector<BowTrack> matches;
        matches.reserve(10000);   // (:587)
        int intraMatchInd = 0;
        auto dist = [&](int ca, int fa, int cb, int fb) {
            return mcorb_hamming256(hd + ((size_t)(img0 + f * C + ca) * kcap + fa) * 32, hd + ((size_t)(img0 + f * C + cb) * kcap + fb) * 32);
        };
        std::vector<int> selected;
        std::vector<std::vector<int>> matchedFlags;
        for (;;) {
            bool end = true;   // checkItersEnd (:569-575)
            for (int c = 0; c < C; c++) end = end && F.fv[c][it[c]].node >= F.fv[c][last[c]].node;
            if (end) break;
            uint32_t min_val = 0x7ffffffeu;
            selected.clear();
            for (int c = 0; c < C; c++) {
                const uint32_t w = it[c] == last[c] ? 0x7fffffffu : F.fv[c][it[c]].node;
                if (w < min_val) { min_val = w; selected.clear(); selected.push_back(c); }
                else if (w == min_val) selected.push_back(c);
            }
            matchedFlags.assign(selected.size(), std::vector<int>());
            for (size_t i = 0; i < selected.size(); i++) matchedFlags[i].assign(F.fv[selected[i]][it[selected[i]]].cnt, -1);
            if (selected.size() >= 2) {
                for (int i = 0; i < (int)selected.size() - 1; i++) {
                    const int cam1 = selected[i];
                    const FvRun &run1 = F.fv[cam1][it[cam1]];
                    const int32_t *feat_cam1 = F.feats[cam1].data() + run1.beg;
                    for (int a = 0; a < run1.cnt; a++) {
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
                            const int32_t *feat_cam2 = F.feats[cam2].data() + F.fv[cam2][it[cam2]].beg;
                            const int4 t = tab[(size_t)pair_of(C, cam1, cam2) * kcap + feat_cam1[a]];
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
                            out.words.push_back(F.fv[selected[0]][it[selected[0]]].node);
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
        out.tracks.resize(matches.size() * C);
        out.n_rays.resize(matches.size());
        for (size_t m = 0; m < matches.size(); m++) {
            for (int c = 0; c < C; c++) out.tracks[m * C + c] = matches[m].matchIndex[c];
            out.n_rays[m] = matches[m].n_rays;
        }
    }, R.pool_threads + s->index);
    for (int f = 0; f < nframes; f++) s->bow_ok[frame0 + f] = 1;   // exactly the frames matched by this call
    if (prof)
        fprintf(stderr, "[mcorb host prof] match_bow x%d frames: descend+sync %.0f us, feature vectors + tables %.0f, best2+copy %.0f, replay %.0f\n",
                nframes, us(T0, T1), us(T1, T2), us(T2, T3), us(T3, now()));
    return MCORB_OK;
}

extern "C" int mcorb_rig_get_bow_tracks(mcorb_rig *r, int slot, int frame, int32_t *tracks, int32_t *n_rays, int cap_tracks,
                                        int *ntracks_out, uint32_t *words, int cap_words, int *nwords_out)
{
    if (ntracks_out) *ntracks_out = 0;
    if (nwords_out) *nwords_out = 0;
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) { set_error("bow tracks: bad argument"); return MCORB_E_ARG; }
    Slot *s = r->rig.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    if (frame < 0 || frame >= (int)s->bow_ok.size() || !s->bow_ok[frame]) { set_error("bow tracks: frame not matched since the slot's last extraction"); return MCORB_E_STATE; }
    const BowFrameOut &o = s->bow[frame];
    const int C = r->rig.ncams, n = (int)o.n_rays.size();
    if (ntracks_out) *ntracks_out = n;
    if (nwords_out) *nwords_out = (int)o.words.size();
    if (n > cap_tracks || (words && (int)o.words.size() > cap_words) || (n && !tracks)) { set_error("bow tracks: output too small"); return MCORB_E_CAP; }
    if (n) memcpy(tracks, o.tracks.data(), (size_t)n * C * sizeof(int32_t));
    if (n_rays && n) memcpy(n_rays, o.n_rays.data(), (size_t)n * sizeof(int32_t));
    if (words && !o.words.empty()) memcpy(words, o.words.data(), o.words.size() * sizeof(uint32_t));
    return MCORB_OK;
}

// one frame, raw keypoint rows: the round-1 entry point
extern "C" int mcorb_rig_match_bow(mcorb_rig *r, int slot, int frame, mcorb_vocab *v, int levelsup, double max_neighbor_ratio,
                                   int32_t *tracks, int32_t *n_rays, int cap_tracks, int *ntracks_out, uint32_t *words,
                                   int cap_words, int *nwords_out)
{
    if (ntracks_out) *ntracks_out = 0;
    if (nwords_out) *nwords_out = 0;
    if (!tracks) { set_error("match_bow: bad argument"); return MCORB_E_ARG; }
    const int st = mcorb_rig_match_bow_frames(r, slot, frame, 1, v, levelsup, max_neighbor_ratio, nullptr);
    if (st != MCORB_OK) return st;
    return mcorb_rig_get_bow_tracks(r, slot, frame, tracks, n_rays, cap_tracks, ntracks_out, words, cap_words, nwords_out);
}
