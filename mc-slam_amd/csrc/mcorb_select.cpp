// mcorb_select.cpp -- host stage: quad-tree keypoint selection on GPU-bucketed candidates.
//
// Functional equivalent of ORBextractor::DistributeOctTree + ExtractorNode::DivideNode +
// compareNodes (MCSlam/src/ORBextractor.cpp:479-778).  The reference's result is order-defined
// by (a) std::list push_front/erase order and (b) std::sort's placement of equivalent
// (count, UL.x) entries, so the LIST DISCIPLINE stays here, on the host, and std::sort is called
// on the same sequence with the same predicate.  What moved to the GPU is the data-parallel part:
// k_compact sorts each level's candidates by quad-tree path code (path_code(), mcorb_common.h)
// and ships the bucket start offsets, so for the first `depth` splits a node's key set is a
// contiguous range whose size is one subtraction; deeper nodes (a handful of keys each) are
// partitioned here.  The final "max response, first one wins" pick (:757-775) resolves ties from
// the coordinates, which encode vToDistributeKeys order (cell row, cell col, y, x).
#include "mcorb_select.h"

#include <math.h>

#include <algorithm>
#include <utility>

namespace mcorb {

namespace {
struct Node {
    int x0, y0, x1, y1;   // UL = (x0,y0), BR = (x1,y1); UR/BL follow (nodes stay rectangles)
    int d;                // depth below the root
    uint32_t code;        // path code prefix (root index, then 2 bits per split)
    int beg, cnt;         // key range: in the sorted candidate array (arena == false) or in the arena
    bool arena;
    int prev, next;
    bool noMore;
};
}  // namespace

struct SelectScratch::Impl {
    std::vector<Node> nodes;
    std::vector<int> arena;
    std::vector<uint8_t> quad;
    // nodes that can still be divided: (key count << 12 | UL.x) in the upper half, the node's index in the lower -- compareNodes
    // (:537-552) looks at exactly those two fields, so the sort compares upper halves and never touches the node array
    std::vector<uint64_t> expand, prevExpand;
    static uint64_t expand_entry(int cnt, int x0, int id) { return ((uint64_t)(((uint32_t)cnt << 12) | (uint32_t)x0) << 32) | (uint32_t)id; }
    int head = -1, tail = -1, count = 0;

    int new_node() { nodes.emplace_back(); return (int)nodes.size() - 1; }
    void push_back(int id)
    {
        Node &n = nodes[id];
        n.prev = tail; n.next = -1;
        if (tail >= 0) nodes[tail].next = id; else head = id;
        tail = id; count++;
    }
    void push_front(int id)
    {
        Node &n = nodes[id];
        n.prev = -1; n.next = head;
        if (head >= 0) nodes[head].prev = id; else tail = id;
        head = id; count++;
    }
    int erase(int id)   // returns the next node, like std::list::erase
    {
        Node &n = nodes[id];
        const int nx = n.next;
        if (n.prev >= 0) nodes[n.prev].next = n.next; else head = n.next;
        if (n.next >= 0) nodes[n.next].prev = n.prev; else tail = n.prev;
        count--;
        return nx;
    }
};

SelectScratch::SelectScratch() : impl(new Impl) {}
SelectScratch::~SelectScratch() { delete impl; }

// DivideNode (:479-535): children in n1..n4 order; child[q] = -1 where a child has no keys.
static bool divide(SelectScratch::Impl &S, const uint32_t *cand, const int *bstart, int D, int id, int child[4])
{
    const Node P = S.nodes[id];
    const int sx = P.x0 + ((P.x1 - P.x0 + 1) >> 1);   // UL.x + ceil((UR.x-UL.x)/2)
    const int sy = P.y0 + ((P.y1 - P.y0 + 1) >> 1);
    int cnt[4], beg[4];
    bool in_arena = false;
    if (P.d < D) {
        // children are bucket ranges of the GPU-sorted array: five consecutive boundaries
        const int shift = 2 * (D - P.d - 1);
        const int *b = bstart + ((size_t)(P.code << 2) << shift);
        const int e0 = b[0], e1 = b[(size_t)1 << shift], e2 = b[(size_t)2 << shift], e3 = b[(size_t)3 << shift], e4 = b[(size_t)4 << shift];
        beg[0] = e0; beg[1] = e1; beg[2] = e2; beg[3] = e3;
        cnt[0] = e1 - e0; cnt[1] = e2 - e1; cnt[2] = e3 - e2; cnt[3] = e4 - e3;
    } else {
        if (!cand) return false;   // the candidate list was not shipped: cannot happen when nz >= N (see k_compact)
        in_arena = true;
        cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
        if ((int)S.quad.size() < P.cnt) S.quad.resize(P.cnt);
        for (int i = 0; i < P.cnt; i++) {
            const uint32_t c = cand[P.arena ? S.arena[P.beg + i] : P.beg + i];
            // kp.pt.x < n1.UR.x ? (kp.pt.y < n1.BR.y ? n1 : n3) : (kp.pt.y < n1.BR.y ? n2 : n4)
            const int q = (cand_x(c) < sx ? 0 : 1) + (cand_y(c) < sy ? 0 : 2);
            S.quad[i] = (uint8_t)q;
            cnt[q]++;
        }
        const size_t base = S.arena.size();
        S.arena.resize(base + P.cnt);
        beg[0] = (int)base; beg[1] = beg[0] + cnt[0]; beg[2] = beg[1] + cnt[1]; beg[3] = beg[2] + cnt[2];
        int pos[4] = {beg[0], beg[1], beg[2], beg[3]};
        for (int i = 0; i < P.cnt; i++) S.arena[pos[S.quad[i]]++] = P.arena ? S.arena[P.beg + i] : P.beg + i;
    }
    const int bx0[4] = {P.x0, sx, P.x0, sx}, bx1[4] = {sx, P.x1, sx, P.x1};
    const int by0[4] = {P.y0, P.y0, sy, sy}, by1[4] = {sy, sy, P.y1, P.y1};
    for (int q = 0; q < 4; q++) {
        child[q] = -1;
        if (cnt[q] == 0) continue;
        Node n;
        n.x0 = bx0[q]; n.x1 = bx1[q]; n.y0 = by0[q]; n.y1 = by1[q];
        n.d = P.d + 1;
        n.code = (P.code << 2) | (uint32_t)q;
        n.beg = beg[q]; n.cnt = cnt[q];
        n.arena = in_arena;
        n.prev = n.next = -1;
        n.noMore = (cnt[q] == 1);
        child[q] = (int)S.nodes.size();
        S.nodes.push_back(n);
    }
    return true;
}

static inline int win_pos(const BucketBest &b) { return (int)b.pos; }
static inline int win_pos(const BucketWin &) { return -1; }

template <typename BB>
int select_octree(const uint32_t *cand, const int *bstart, const BB *bbest, int n, const SelectParams &P, int *out_idx,
                  uint32_t *out_val, SelectScratch &scratch)
{
    SelectScratch::Impl &S = *scratch.impl;
    S.nodes.clear(); S.arena.clear(); S.expand.clear(); S.prevExpand.clear();
    S.head = S.tail = -1; S.count = 0;
    const int N = P.N, D = P.depth;
    if (P.nIni < 1) return -2;
    if (n <= 0) return 0;
    // the sort key packs (key count << 12 | UL.x) into 32 bits: counts below 2^20, level widths below 4096 (the rig geometry's
    // limit); anything larger would silently reorder compareNodes' ties
    if (n >= (1 << 20) || P.maxX - P.minX >= 4096) return -4;
    const float hX = P.hX;

    // root nodes (:567-578); their key sets are the top-level bucket ranges
    for (int i = 0; i < P.nIni; i++) {
        const int id = S.new_node();
        Node &nd = S.nodes[id];
        nd.x0 = (int)(hX * (float)i);
        nd.x1 = (int)(hX * (float)(i + 1));
        nd.y0 = 0;
        nd.y1 = P.maxY - P.minY;
        nd.d = 0;
        nd.code = (uint32_t)i;
        nd.beg = bstart[(uint32_t)i << (2 * D)];
        nd.cnt = bstart[(uint32_t)(i + 1) << (2 * D)] - nd.beg;
        nd.arena = false;
        nd.noMore = false;
        S.push_back(id);
    }
    // (:587-600)
    for (int it = S.head; it >= 0;) {
        Node &nd = S.nodes[it];
        if (nd.cnt == 1) { nd.noMore = true; it = nd.next; }
        else if (nd.cnt == 0) it = S.erase(it);
        else it = nd.next;
    }

    bool bFinish = false;
    while (!bFinish) {
        int prevSize = S.count;
        int nToExpand = 0;
        S.expand.clear();
        for (int it = S.head; it >= 0;) {
            if (S.nodes[it].noMore) { it = S.nodes[it].next; continue; }
            int ch[4];
            if (!divide(S, cand, bstart, D, it, ch)) return -3;
            for (int q = 0; q < 4; q++) {
                if (ch[q] < 0) continue;
                S.push_front(ch[q]);
                if (S.nodes[ch[q]].cnt > 1) {
                    nToExpand++;
                    S.expand.push_back(SelectScratch::Impl::expand_entry(S.nodes[ch[q]].cnt, S.nodes[ch[q]].x0, ch[q]));
                }
            }
            it = S.erase(it);
        }
        if (S.count >= N || S.count == prevSize) {
            bFinish = true;
        } else if (S.count + nToExpand * 3 > N) {
            while (!bFinish) {
                prevSize = S.count;
                S.prevExpand = S.expand;
                S.expand.clear();
                // compareNodes (:537-552): (count, UL.x) ascending; equivalent entries land where std::sort puts them -- the
                // same algorithm on the same sequence with a predicate that answers the same for every pair
                std::sort(S.prevExpand.begin(), S.prevExpand.end(), [](uint64_t e1, uint64_t e2) { return (e1 >> 32) < (e2 >> 32); });
                for (int j = (int)S.prevExpand.size() - 1; j >= 0; j--) {
                    const int id = (int)(uint32_t)S.prevExpand[j];
                    int ch[4];
                    if (!divide(S, cand, bstart, D, id, ch)) return -3;
                    for (int q = 0; q < 4; q++) {
                        if (ch[q] < 0) continue;
                        S.push_front(ch[q]);
                        if (S.nodes[ch[q]].cnt > 1) S.expand.push_back(SelectScratch::Impl::expand_entry(S.nodes[ch[q]].cnt, S.nodes[ch[q]].x0, ch[q]));
                    }
                    S.erase(id);
                    if (S.count >= N) break;
                }
                if (S.count >= N || S.count == prevSize) bFinish = true;
            }
        }
    }

    // best response per node; among equal responses the first key in vToDistributeKeys order wins
    // (:757-775), i.e. the smallest (cell row, cell col, y, x)
    int m = 0;
    for (int it = S.head; it >= 0; it = S.nodes[it].next) {
        const Node &nd = S.nodes[it];
        if (!nd.arena) {
            // the node is a run of buckets whose winners the GPU already found: the largest key wins
            // (a single-key node is the one non-empty bucket of its run)
            const int shift = 2 * (D - nd.d);
            const BB *bb = bbest + ((size_t)nd.code << shift);
            const int nb = 1 << shift;
            uint32_t bestKey = 0, bestVal = 0;
            int bestPos = -1;
            for (int b = 0; b < nb; b++)
                if (bb[b].key > bestKey) { bestKey = bb[b].key; bestPos = win_pos(bb[b]); bestVal = bb[b].val; }
            out_val[m] = bestVal;
            out_idx[m++] = bestPos;
            continue;
        }
        if (nd.cnt == 1) { out_val[m] = cand[S.arena[nd.beg]]; out_idx[m++] = S.arena[nd.beg]; continue; }
        int best = -1, bestR = -1;
        uint64_t bestO = 0;
        for (int k = 0; k < nd.cnt; k++) {
            const int idx = nd.arena ? S.arena[nd.beg + k] : nd.beg + k;
            const uint32_t c = cand[idx];
            const int r = cand_resp(c);
            if (r < bestR) continue;
            const int x = cand_x(c), y = cand_y(c);
            const uint64_t o = ((((uint64_t)((y - 3) / P.hCell) << 12 | (uint64_t)((x - 3) / P.wCell)) << 12 | (uint64_t)y) << 12) |
                               (uint64_t)x;
            if (r > bestR || o < bestO) { bestR = r; best = idx; bestO = o; }
        }
        out_val[m] = cand[best];
        out_idx[m++] = best;
    }
    return m;
}

template int select_octree<BucketBest>(const uint32_t *, const int *, const BucketBest *, int, const SelectParams &, int *, uint32_t *,
                                       SelectScratch &);
template int select_octree<BucketWin>(const uint32_t *, const int *, const BucketWin *, int, const SelectParams &, int *, uint32_t *,
                                      SelectScratch &);

// CPU statement of what k_compact does on the GPU for one level (used by the host-only test hook):
// counting sort of the candidates by path code, bucket starts out.
void host_bucket_sort(const uint32_t *cand, int n, const SelectParams &P, std::vector<uint32_t> &sorted,
                      std::vector<int> &perm, std::vector<int> &bstart, std::vector<BucketBest> &bbest)
{
    const int B = P.nIni << (2 * P.depth);
    bstart.assign((size_t)B + 1, 0);
    std::vector<uint32_t> code(n);
    const int W0 = P.maxX - P.minX, H0 = P.maxY - P.minY;
    for (int i = 0; i < n; i++) {
        code[i] = path_code(cand_x(cand[i]), cand_y(cand[i]), W0, H0, P.nIni, P.hX, P.depth);
        bstart[code[i] + 1]++;
    }
    for (int b = 0; b < B; b++) bstart[b + 1] += bstart[b];
    std::vector<int> pos(bstart.begin(), bstart.end() - 1);
    sorted.resize(n);
    perm.resize(n);
    bbest.assign((size_t)B, BucketBest{0, 0, 0});
    for (int i = 0; i < n; i++) {
        const int s = pos[code[i]]++;
        sorted[s] = cand[i];
        perm[s] = i;
        const uint32_t key = ((uint32_t)cand_resp(cand[i]) << 23) | (uint32_t)(kPickOrderMask - i);
        if (key > bbest[code[i]].key) bbest[code[i]] = BucketBest{key, (uint32_t)s, cand[i]};
    }
}

SelectParams make_select_params(int minX, int maxX, int minY, int maxY, int N, int wCell, int hCell)
{
    SelectParams P;
    P.minX = minX; P.maxX = maxX; P.minY = minY; P.maxY = maxY; P.N = N;
    P.wCell = wCell > 0 ? wCell : (1 << 20);
    P.hCell = hCell > 0 ? hCell : (1 << 20);
    P.nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));   // (:558)
    P.hX = P.nIni >= 1 ? (float)(maxX - minX) / P.nIni : 0.f;      // (:560)
    // depth of the GPU bucketing: about log4(N / nIni), i.e. where a uniform spread reaches the quota
    int d = 1;
    while (d < 5 && (P.nIni > 0 ? P.nIni : 1) * (1 << (2 * d)) < N) d++;
    P.depth = d;
    return P;
}

}  // namespace mcorb
