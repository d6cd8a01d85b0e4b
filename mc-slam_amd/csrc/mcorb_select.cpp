// mcorb_select.cpp -- host stage: quad-tree keypoint selection.
//
// Functional equivalent of ORBextractor::DistributeOctTree + ExtractorNode::
// DivideNode + compareNodes (MCSlam/src/ORBextractor.cpp:479-778), rebuilt for
// speed: keys are indices into the packed candidate list, node key sets are
// ranges of an append-only arena, the node list is an index-linked list.  The
// reference's result is order-defined by (a) std::list push_front/erase order
// and (b) std::sort's placement of equivalent (count, UL.x) entries, so this
// stage keeps the same list discipline and calls std::sort on the same sequence
// with the same predicate.  It is serial per (image, level) and runs on the
// engine's worker pool between the two GPU phases.
#include "mcorb_select.h"

#include <math.h>

#include <algorithm>
#include <utility>

namespace mcorb {

namespace {
struct Node {
    int x0, y0, x1, y1;   // UL = (x0,y0), BR = (x1,y1); UR/BL follow (nodes stay rectangles)
    int kbeg, kcnt;       // key range in the arena
    int prev, next;
    bool noMore;
};
}  // namespace

struct SelectScratch::Impl {
    std::vector<Node> nodes;
    std::vector<int> arena;
    std::vector<uint8_t> quad;
    std::vector<std::pair<int, int>> expand, prevExpand;
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

// DivideNode (:479-535): splits node `id` into up to four children appended to
// the arena in n1..n4 order; returns child ids (-1 where a child has no keys).
static void divide(SelectScratch::Impl &S, const uint32_t *cand, int id, int child[4])
{
    const Node P = S.nodes[id];
    const int halfX = (int)ceilf((float)(P.x1 - P.x0) / 2);
    const int halfY = (int)ceilf((float)(P.y1 - P.y0) / 2);
    const int sx = P.x0 + halfX, sy = P.y0 + halfY;
    int cnt[4] = {0, 0, 0, 0};
    if ((int)S.quad.size() < P.kcnt) S.quad.resize(P.kcnt);
    for (int i = 0; i < P.kcnt; i++) {
        const uint32_t c = cand[S.arena[P.kbeg + i]];
        // kp.pt.x < n1.UR.x ? (kp.pt.y < n1.BR.y ? n1 : n3) : (kp.pt.y < n1.BR.y ? n2 : n4)
        const int q = (cand_x(c) < sx ? 0 : 1) + (cand_y(c) < sy ? 0 : 2);
        S.quad[i] = (uint8_t)q;
        cnt[q]++;
    }
    const size_t base = S.arena.size();
    S.arena.resize(base + P.kcnt);
    int pos[4] = {(int)base, (int)base + cnt[0], (int)base + cnt[0] + cnt[1], (int)base + cnt[0] + cnt[1] + cnt[2]};
    const int beg[4] = {pos[0], pos[1], pos[2], pos[3]};
    for (int i = 0; i < P.kcnt; i++) S.arena[pos[S.quad[i]]++] = S.arena[P.kbeg + i];
    const int bx0[4] = {P.x0, sx, P.x0, sx}, bx1[4] = {sx, P.x1, sx, P.x1};
    const int by0[4] = {P.y0, P.y0, sy, sy}, by1[4] = {sy, sy, P.y1, P.y1};
    for (int q = 0; q < 4; q++) {
        child[q] = -1;
        if (cnt[q] == 0) continue;
        const int c = S.new_node();
        Node &n = S.nodes[c];
        n.x0 = bx0[q]; n.x1 = bx1[q]; n.y0 = by0[q]; n.y1 = by1[q];
        n.kbeg = beg[q]; n.kcnt = cnt[q];
        n.noMore = (cnt[q] == 1);
        child[q] = c;
    }
}

int select_octree(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N, int *out_idx,
                  SelectScratch &scratch)
{
    SelectScratch::Impl &S = *scratch.impl;
    S.nodes.clear(); S.arena.clear(); S.expand.clear(); S.prevExpand.clear();
    S.head = S.tail = -1; S.count = 0;
    S.nodes.reserve(4 * (size_t)(N > 64 ? N : 64) + 64);
    S.arena.reserve((size_t)n * 12 + 64);

    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) return -2;
    const float hX = (float)(maxX - minX) / nIni;

    // root nodes (:567-578) and key assignment (:581-585), stable by construction
    std::vector<int> rootCnt(nIni, 0);
    if ((int)S.quad.size() < n) S.quad.resize(n);
    std::vector<int> rootOf(n);
    for (int i = 0; i < n; i++) {
        int r = (int)((float)cand_x(cand[i]) / hX);
        if (r >= nIni) r = nIni - 1;   // the reference would index out of range here; cannot happen for x < maxX-minX
        rootOf[i] = r;
        rootCnt[r]++;
    }
    S.arena.resize(n);
    std::vector<int> rootPos(nIni, 0);
    for (int r = 1; r < nIni; r++) rootPos[r] = rootPos[r - 1] + rootCnt[r - 1];
    {
        std::vector<int> p = rootPos;
        for (int i = 0; i < n; i++) S.arena[p[rootOf[i]]++] = i;
    }
    for (int i = 0; i < nIni; i++) {
        const int id = S.new_node();
        Node &nd = S.nodes[id];
        nd.x0 = (int)(hX * (float)i);
        nd.x1 = (int)(hX * (float)(i + 1));
        nd.y0 = 0;
        nd.y1 = maxY - minY;
        nd.kbeg = rootPos[i]; nd.kcnt = rootCnt[i];
        nd.noMore = false;
        S.push_back(id);
    }
    // (:587-600)
    for (int it = S.head; it >= 0;) {
        Node &nd = S.nodes[it];
        if (nd.kcnt == 1) { nd.noMore = true; it = nd.next; }
        else if (nd.kcnt == 0) it = S.erase(it);
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
            divide(S, cand, it, ch);
            for (int q = 0; q < 4; q++) {
                if (ch[q] < 0) continue;
                S.push_front(ch[q]);
                if (S.nodes[ch[q]].kcnt > 1) {
                    nToExpand++;
                    S.expand.emplace_back(S.nodes[ch[q]].kcnt, ch[q]);
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
                const std::vector<Node> &nodes = S.nodes;
                // compareNodes (:537-552); equivalent entries land where std::sort puts them
                std::sort(S.prevExpand.begin(), S.prevExpand.end(),
                          [&nodes](const std::pair<int, int> &e1, const std::pair<int, int> &e2) {
                              if (e1.first < e2.first) return true;
                              else if (e1.first > e2.first) return false;
                              else return nodes[e1.second].x0 < nodes[e2.second].x0;
                          });
                for (int j = (int)S.prevExpand.size() - 1; j >= 0; j--) {
                    const int id = S.prevExpand[j].second;
                    int ch[4];
                    divide(S, cand, id, ch);
                    for (int q = 0; q < 4; q++) {
                        if (ch[q] < 0) continue;
                        S.push_front(ch[q]);
                        if (S.nodes[ch[q]].kcnt > 1) S.expand.emplace_back(S.nodes[ch[q]].kcnt, ch[q]);
                    }
                    S.erase(id);
                    if (S.count >= N) break;
                }
                if (S.count >= N || S.count == prevSize) bFinish = true;
            }
        }
    }

    // best response per node, first maximum wins (:757-775)
    int m = 0;
    for (int it = S.head; it >= 0; it = S.nodes[it].next) {
        const Node &nd = S.nodes[it];
        int best = S.arena[nd.kbeg];
        int bestR = cand_resp(cand[best]);
        for (int k = 1; k < nd.kcnt; k++) {
            const int idx = S.arena[nd.kbeg + k];
            const int r = cand_resp(cand[idx]);
            if (r > bestR) { bestR = r; best = idx; }
        }
        out_idx[m++] = best;
    }
    return m;
}

}  // namespace mcorb
