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
    int2 *d_out = nullptr, *h_out = nullptr;
    int cap = 0;
};

static void free_vocab(mcorb_vocab *v)
{
    if (!v) return;
    (void)hipFree(v->d_child_start); (void)hipFree(v->d_child_count); (void)hipFree(v->d_child_id);
    (void)hipFree(v->d_child_desc); (void)hipFree(v->d_desc); (void)hipFree(v->d_out);
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
    *out = v;
    return MCORB_OK;
}

// BowVector::addWeight / addIfNotExist / normalize and FeatureVector::addFeature, on std::map like DBoW2
static void assemble(const mcorb_vocab *v, const int2 *res, int n, std::map<uint32_t, double> &bow,
                     std::map<uint32_t, std::vector<int32_t>> &fv)
{
    const bool tf_like = v->weighting == 0 || v->weighting == 1;   // TF_IDF, TF
    for (int i = 0; i < n; i++) {
        const int node = res[i].x;
        const uint32_t id = (uint32_t)v->word_id[node];
        const double w = v->weight[node];
        if (w > 0) {   // not stopped
            auto it = bow.lower_bound(id);
            if (it != bow.end() && !(id < it->first)) {
                if (tf_like) it->second += w;          // addWeight
            } else {
                bow.insert(it, std::make_pair(id, w));   // addWeight / addIfNotExist
            }
            fv[(uint32_t)res[i].y].push_back(i);
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

static int emit(const std::map<uint32_t, double> &bow, const std::map<uint32_t, std::vector<int32_t>> &fv, uint32_t *bow_ids,
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
    HIPCHK(hipMalloc((void **)&v->d_out, (size_t)cap * sizeof(int2)));
    HIPCHK(hipHostMalloc((void **)&v->h_out, (size_t)cap * sizeof(int2), hipHostMallocDefault));
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
    std::map<uint32_t, double> bow;
    std::map<uint32_t, std::vector<int32_t>> fv;
    if (n > 0) {
        const int nid_level = v->L - levelsup;   // <= 0: the feature vector is keyed by the root (node 0)
        launch_bow_descend(st, d_desc, n, v->d_child_start, v->d_child_count, v->d_child_desc, v->d_child_id, nid_level, v->d_out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(v->h_out, v->d_out, (size_t)n * sizeof(int2), hipMemcpyDeviceToHost, st));
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
