// mcorb_select_gpu.hip -- DistributeOctTree's list discipline on the GPU (ORBextractor.cpp:554-778), one wave per (image, level).
//
// k_compact has already counting-sorted a level's FAST candidates by quad-tree path code and found every bucket's winner
// (mcorb_kernels.hip); what the host stage (mcorb_select.cpp) did with those tables -- the std::list push_front / erase sequence,
// the std::sort of (key count, UL.x) and the largest-first divisions until N nodes exist -- happens here, so that a batch is ONE
// submission: pyramid, FAST, compaction, selection, assembly, descriptors, matching, with no host round trip in the middle.
//
// * A full pass (:615-678) divides every node with more than one key; children are pushed to the FRONT in n1..n4 order, nodes
//   with one key stay where they are.  The new list is therefore: the children of the dividable nodes in REVERSE node order, each
//   node's non-empty children in n4, n3, n2, n1 order, then the single-key nodes in their old order -- positions are prefix sums.
// * The careful phase (:688-752) sorts the dividable nodes by (key count, UL.x) with std::sort and divides from the back until N
//   nodes exist.  std::sort is restated in mcorb_sortmodel.h (closed-form partitions, block-wise stable finish, heap fallback --
//   pinned against libstdc++ on the CPU); how many nodes get divided is a prefix sum over the sorted order, the new list again
//   a set of positions.
// * The final pick (:757-775) is the best of the node's bucket winners.
// A node at the bucketing depth that has to be divided again needs the candidates themselves (clustered corners): the kernel
// raises the image's fallback flag and the engine runs the host stage for that batch (mcorb_engine.cpp), exactly as before.
//
// k_assemble then does ORBextractor::operator()'s assembly (:1103-1170): level order, coordinate scaling, the lapping partition
// (mono keypoints from the front, stereo ones from the back).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mcorb_common.h"
#include "mcorb_kernels.h"
#include "mcorb_sortmodel.h"

namespace mcorb {

namespace {

__device__ __forceinline__ int sel_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int sel_rank(unsigned long long mask, int acc = 0)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, (uint32_t)acc));
}
__device__ __forceinline__ void sel_sync()   // LDS written by some lanes of this wave, read by others
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ int sel_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// inclusive prefix sum over the wave for values 0 .. 7 (children per node): three ballots, no cross-lane traffic; tot = the wave's sum
__device__ __forceinline__ int sel_scan3(int v, int &tot)
{
    const unsigned long long b0 = __ballot(v & 1), b1 = __ballot(v & 2), b2 = __ballot(v & 4);
    tot = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    return sel_rank(b0) + 2 * sel_rank(b1) + 4 * sel_rank(b2) + v;
}

// a tree node: x = path code (root index, then 2 bits per split) | depth << 28, y = UL.x | UR.x << 16 (level coordinates relative
// to minX), z = its key count.  Down to the bucketing depth D a node's key set is a run of k_compact's buckets and every count is
// a difference of bucket starts; a node BELOW D (corners so clustered that fewer than N buckets are non-empty) lives inside one
// bucket: its key set is that bucket's candidates filtered by the split lines on its path below D, found by scanning the bucket
// (in device memory) whenever the node is divided or its best key is picked.
typedef uint4 SelNode;
constexpr int kSelDepthShift = 28;
__device__ __forceinline__ uint32_t node_code(SelNode n) { return n.x & ((1u << kSelDepthShift) - 1u); }
__device__ __forceinline__ int node_depth(SelNode n) { return (int)(n.x >> kSelDepthShift); }
__device__ __forceinline__ int node_count(SelNode n) { return (int)n.z; }

struct SelCtx {
    const int *bst;          // bucket starts (LDS)
    const uint32_t *cand;    // the level's candidates, bucket-sorted (device memory)
    int D, maxDepth, H0, wCell, hCell;
    int deepCap;             // a bucket with more candidates than this is not scanned node by node: the level goes to the host stage
    float hX;
};
// geometry of a node below the bucketing depth, recomputed from its path: rectangle (splits happen at its middle) and the
// split lines below D that bound its key set inside the bucket (a side that no split below D produced is open: root / bucket
// membership is not a rectangle test -- path_code(), mcorb_common.h)
struct DeepGeom { int x0, x1, y0, y1, mx0, mx1, my0, my1; };
__device__ inline DeepGeom deep_geom(const SelCtx &C, SelNode n)
{
    const int d = node_depth(n);
    const uint32_t code = node_code(n);
    const int r = (int)(code >> (2 * d));
    DeepGeom G;
    G.x0 = (int)__fmul_rn(C.hX, (float)r); G.x1 = (int)__fmul_rn(C.hX, (float)(r + 1));
    G.y0 = 0; G.y1 = C.H0;
    G.mx0 = 0; G.mx1 = 1 << 20; G.my0 = 0; G.my1 = 1 << 20;
    for (int j = 1; j <= d; j++) {
        const int q = (int)((code >> (2 * (d - j))) & 3u);
        const int sx = G.x0 + ((G.x1 - G.x0 + 1) >> 1), sy = G.y0 + ((G.y1 - G.y0 + 1) >> 1);
        if (q & 1) { G.x0 = sx; if (j > C.D) G.mx0 = sx; } else { G.x1 = sx; if (j > C.D) G.mx1 = sx; }
        if (q & 2) { G.y0 = sy; if (j > C.D) G.my0 = sy; } else { G.y1 = sy; if (j > C.D) G.my1 = sy; }
    }
    return G;
}
// DivideNode (:479-535): key counts of n1..n4.  Returns false when the node cannot be divided here (path code out of bits, or a
// bucket too large to scan): the level goes to the host stage.
__device__ inline bool node_kids(const SelCtx &C, SelNode n, int cnt[4])
{
    const int d = node_depth(n);
    if (d < C.D) {
        const int sh = 2 * (C.D - d - 1);
        const int base = (int)((node_code(n) << 2) << sh);
        const int e0 = C.bst[base], e1 = C.bst[base + (1 << sh)], e2 = C.bst[base + (2 << sh)], e3 = C.bst[base + (3 << sh)], e4 = C.bst[base + (4 << sh)];
        cnt[0] = e1 - e0; cnt[1] = e2 - e1; cnt[2] = e3 - e2; cnt[3] = e4 - e3;
        return true;
    }
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
    if (d >= C.maxDepth) return false;
    const int b = (int)(node_code(n) >> (2 * (d - C.D)));
    const int k0 = C.bst[b], k1 = C.bst[b + 1];
    if (k1 - k0 > C.deepCap) return false;
    const DeepGeom G = deep_geom(C, n);
    const int sx = G.x0 + ((G.x1 - G.x0 + 1) >> 1), sy = G.y0 + ((G.y1 - G.y0 + 1) >> 1);
    for (int k = k0; k < k1; k++) {
        const uint32_t c = C.cand[k];
        const int x = cand_x(c), y = cand_y(c);
        if (x >= G.mx0 && x < G.mx1 && y >= G.my0 && y < G.my1) cnt[(x >= sx ? 1 : 0) + (y >= sy ? 2 : 0)]++;   // (:505-530)
    }
    return true;
}
__device__ __forceinline__ SelNode node_child(SelNode n, int q, int cnt)
{
    const int x0 = (int)(n.y & 0xffffu), x1 = (int)(n.y >> 16);
    const int sx = x0 + ((x1 - x0 + 1) >> 1);   // UL.x + ceil((UR.x - UL.x) / 2)
    const int cx0 = (q & 1) ? sx : x0, cx1 = (q & 1) ? x1 : sx;
    return SelNode{((node_code(n) << 2) | (uint32_t)q) | ((uint32_t)(node_depth(n) + 1) << kSelDepthShift), (uint32_t)cx0 | ((uint32_t)cx1 << 16), (uint32_t)cnt, 0u};
}
// the node's best key (:757-775): largest response, the first in vToDistributeKeys order among equals (cell row, cell column, y, x)
__device__ inline uint32_t node_best(const SelCtx &C, const uint2 *win, SelNode n)
{
    const int d = node_depth(n);
    if (d <= C.D) {   // a run of buckets: the best of their winners (k_compact)
        const int sh = 2 * (C.D - d);
        const int b0 = (int)(node_code(n) << sh), nb = 1 << sh;
        uint32_t bestKey = 0, bestVal = 0;
        for (int b = 0; b < nb; b++) {
            const uint2 w = win[b0 + b];
            if (w.x > bestKey) { bestKey = w.x; bestVal = w.y; }
        }
        return bestVal;
    }
    const int b = (int)(node_code(n) >> (2 * (d - C.D)));
    const DeepGeom G = deep_geom(C, n);
    uint32_t best = 0;
    int bestR = -1;
    unsigned long long bestO = 0;
    for (int k = C.bst[b]; k < C.bst[b + 1]; k++) {
        const uint32_t c = C.cand[k];
        const int x = cand_x(c), y = cand_y(c), r = cand_resp(c);
        if (!(x >= G.mx0 && x < G.mx1 && y >= G.my0 && y < G.my1) || r < bestR) continue;
        const unsigned long long o = (((((unsigned long long)((y - 3) / C.hCell) << 12) | (unsigned long long)((x - 3) / C.wCell)) << 12 | (unsigned long long)y) << 12) |
                                     (unsigned long long)x;
        if (r > bestR || o < bestO) { bestR = r; best = c; bestO = o; }
    }
    return best;
}
// per-node summary of a division: non-empty children (bits 0-3), children with more than one key (bits 4-7), 0x100 = dividable
__device__ __forceinline__ uint32_t kids_info(const int cnt[4])
{
    uint32_t m = 0x100u;
#pragma unroll
    for (int q = 0; q < 4; q++) m |= (cnt[q] > 0 ? 1u << q : 0u) | (cnt[q] > 1 ? 16u << q : 0u);
    return m;
}

struct SelLds {
    int *bst;                 // bucket starts of this level (B + 1)
    const uint2 *win;         // the buckets' winners {key, candidate} (k_compact), staged while the tree is built
    SelNode *list[2];         // node lists, ping-pong
    uint64_t *exp[2];         // (key count << 12 | UL.x) << 32 | list position of the nodes that can still be divided
    uint16_t *ta, *tb;        // scratch: partition positions (sort) / per-node and per-entry division summaries (passes)
    uint32_t *blk;            // sort: block of every position (first | last << 16)
    int *stk;                 // sort: range stack
};
constexpr int kSelStack = 3 * 48;
__host__ __device__ inline size_t sel_lds_bytes(int B, int cap) { return (size_t)(B + 1 + 3) / 4 * 16 + (size_t)(B + 1) / 2 * 16 + (size_t)cap * (2 * 16 + 2 * 8 + 2 * 2 + 4) + kSelStack * 4; }

// std::sort(a, a + n) on the upper halves, libstdc++'s permutation (mcorb_sortmodel.h); the result is in `out`
__device__ void wave_std_sort(uint64_t *a, int n, uint64_t *out, SelLds &S)
{
    const int lane = sel_lane();
    uint16_t *lp = S.ta, *rp = S.tb;
    if (n <= 0) return;
    int sp = 0;
    if (lane == 0) { S.stk[0] = 0; S.stk[1] = n; S.stk[2] = 2 * sm_lg(n); }
    sp = 1;
    sel_sync();
    while (sp > 0) {
        sp--;
        int f = sel_uni(S.stk[3 * sp]), l = sel_uni(S.stk[3 * sp + 1]), dl = sel_uni(S.stk[3 * sp + 2]);
        while (l - f > 16) {
            if (dl == 0) {   // depth budget used up: heap sort, one lane (rare: median-of-three killers)
                if (lane == 0) sm_heap_sort(a, f, l);
                sel_sync();
                break;
            }
            dl--;
            const int mid = f + (l - f) / 2;
            const uint32_t ka = sm_key(a[f + 1]), kb = sm_key(a[mid]), kc = sm_key(a[l - 1]);
            const int mi = sm_median3(ka, kb, kc, f + 1, mid, l - 1);
            if (lane == 0) { const uint64_t t = a[f]; a[f] = a[mi]; a[mi] = t; }
            sel_sync();
            const uint32_t p = mi == f + 1 ? ka : (mi == mid ? kb : kc);   // = key(a[f]) after the swap
            if (l - f - 1 <= 64) {
                // the range fits the wave: one position per lane, one read; lane j learns L_j (ascending) and R_j (descending)
                // through the crossbar (ds_permute: an L lane sends its position to lane #(L lanes below it), the others to the
                // unused lanes behind), and the cut comes out of registers -- a third of the LDS round trips of the general path
                const int i = f + 1 + lane;
                const bool in = i < l;
                const uint32_t k = sm_key(a[min(i, l - 1)]);
                const bool isL = in && !(k < p), isR = in && !(p < k);
                const unsigned long long bl = __ballot(isL), br = __ballot(isR);
                const int nL = __popcll(bl), nR = __popcll(br);
                const int below = sel_rank(bl);                                            // L lanes below this one
                const int above = __popcll(lane == 63 ? 0ull : br >> (lane + 1));          // R lanes above this one
                const int Lj = __builtin_amdgcn_ds_permute((isL ? below : nL + (lane - below)) << 2, i);
                const int Rj = __builtin_amdgcn_ds_permute((isR ? above : nR + (63 - lane - above)) << 2, i);
                const bool ok = lane < min(nL, nR) && Lj < Rj;
                const int s = __popcll(__ballot(ok));   // (a prefix of the lanes: L ascends, R descends)
                const int nextL = s < nL ? __builtin_amdgcn_readlane(Lj, s) : (1 << 30), lastR = s > 0 ? __builtin_amdgcn_readlane(Rj, s - 1) : (1 << 30);
                const int cut = min(nextL, lastR);
                if (lane < s) {
                    const uint64_t ex = a[Lj], ey = a[Rj];
                    a[Lj] = ey; a[Rj] = ex;
                }
                if (lane == 0) { S.stk[3 * sp] = cut; S.stk[3 * sp + 1] = l; S.stk[3 * sp + 2] = dl; }
                sp++;
                l = cut;
                sel_sync();
                continue;
            }
            // L: positions of (f, l) with key >= p, ascending; R: positions with key <= p, descending
            int nL = 0, nR = 0;
            for (int c0 = f + 1; c0 < l; c0 += 64) {
                const int i = c0 + lane;
                const bool isL = i < l && !(sm_key(a[min(i, l - 1)]) < p);
                const unsigned long long b = __ballot(isL);
                if (isL) lp[sel_rank(b, nL)] = (uint16_t)i;
                nL += __popcll(b);
            }
            for (int c0 = 0; c0 < l - f - 1; c0 += 64) {
                const int i = l - 1 - (c0 + lane);
                const bool isR = i > f && !(p < sm_key(a[max(i, f + 1)]));
                const unsigned long long b = __ballot(isR);
                if (isR) rp[sel_rank(b, nR)] = (uint16_t)i;
                nR += __popcll(b);
            }
            sel_sync();
            const int m = min(nL, nR);
            int s = 0;
            for (int c0 = 0; c0 < m; c0 += 64) {   // the swapped pairs are a prefix: L ascends, R descends
                const int j = c0 + lane;
                const bool ok = j < m && lp[min(j, m - 1)] < rp[min(j, m - 1)];
                const unsigned long long b = __ballot(ok);
                s += __popcll(b);
                if (b != ~0ull) break;
            }
            const int nextL = s < nL ? (int)lp[s] : (1 << 30), lastR = s > 0 ? (int)rp[s - 1] : (1 << 30);
            const int cut = min(nextL, lastR);
            for (int c0 = 0; c0 < s; c0 += 64) {
                const int j = c0 + lane;
                if (j < s) {
                    const int x = lp[j], y = rp[j];
                    const uint64_t ex = a[x], ey = a[y];
                    a[x] = ey; a[y] = ex;
                }
            }
            if (lane == 0) { S.stk[3 * sp] = cut; S.stk[3 * sp + 1] = l; S.stk[3 * sp + 2] = dl; }
            sp++;
            l = cut;
            sel_sync();
        }
        for (int t = lane; t < l - f; t += 64) S.blk[f + t] = (uint32_t)f | ((uint32_t)l << 16);
    }
    sel_sync();
    // __final_insertion_sort = stable rank inside each block
    for (int i = lane; i < n; i += 64) {
        const uint32_t b = S.blk[i];
        const int bf = (int)(b & 0xffffu), bl = (int)(b >> 16);
        const uint64_t e = a[i];
        const uint32_t ki = sm_key(e);
        int r = 0;
        for (int j = bf; j < bl; j++) {
            const uint32_t kj = sm_key(a[j]);
            r += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
        }
        out[bf + r] = e;
    }
    sel_sync();
}

// one full pass over the list (:615-678); returns the new length, m = nodes that can be divided again (nToExpand)
__device__ int full_pass(const SelLds &S, const SelCtx &C, const SelNode *in, int n, SelNode *out, uint64_t *exp, int cap, int &m, int &fb)
{
    const int lane = sel_lane();
    uint16_t *info = S.ta;
    int H = 0, bad = 0, singles = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        int nc = 0;
        uint32_t inf = 0x100u;
        if (i < n) {
            const SelNode nd = in[i];
            inf = 0;
            if (node_count(nd) > 1) {
                int cnt[4];
                if (!node_kids(C, nd, cnt)) bad = 1;
                inf = kids_info(cnt);
                nc = __popc(inf & 15u);
            }
            info[i] = (uint16_t)inf;
        }
        int tot;
        (void)sel_scan3(nc, tot);
        H += tot;
        singles += __popcll(__ballot(!(inf & 0x100u)));
    }
    if (__ballot(bad != 0) != 0ull || H + singles > cap) { fb = 1; m = 0; return n; }
    sel_sync();
    int crun = 0, erun = 0, trun = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const uint32_t inf = i < n ? info[i] : 0u;
        const uint32_t mask = inf & 15u, emask = (inf >> 4) & 15u;
        const int nc = __popc(mask), ne = __popc(emask);
        int ctot, etot;
        const int cin = sel_scan3(nc, ctot), ein = sel_scan3(ne, etot);
        const bool single = i < n && !(inf & 0x100u);
        const unsigned long long bs = __ballot(single);
        if (inf & 0x100u) {
            const SelNode nd = in[i];
            int cnt[4];
            (void)node_kids(C, nd, cnt);
            const int cincl = crun + cin, eexcl = erun + ein - ne;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!((mask >> q) & 1u)) continue;
                const SelNode ch = node_child(nd, q, cnt[q]);
                const int pos = H - cincl + __popc(mask >> (q + 1));   // children pushed later (q' > q) sit further to the front
                out[pos] = ch;
                if ((emask >> q) & 1u)
                    exp[eexcl + __popc(emask & ((1u << q) - 1u))] = ((uint64_t)(((uint32_t)cnt[q] << 12) | (ch.y & 0xffffu)) << 32) | (uint32_t)pos;
            }
        } else if (single) {
            out[H + sel_rank(bs, trun)] = in[i];
        }
        crun += ctot;
        erun += etot;
        trun += __popcll(bs);
    }
    sel_sync();
    m = erun;
    return H + trun;
}

// one round of the careful phase (:688-752): `srt` = the dividable nodes sorted by std::sort; divides from the back until N nodes
__device__ int careful_round(const SelLds &S, const SelCtx &C, const SelNode *in, int n, SelNode *out, const uint64_t *srt, int m, uint64_t *exp, int N, int cap,
                             int &mOut, int &fb)
{
    const int lane = sel_lane();
    uint16_t *divided = S.ta, *einfo = S.tb;   // per old list position: divided in this round; per entry (division order): summary
    for (int i = lane; i < n; i += 64) divided[i] = 0;
    // division order t = 0 .. m-1 is the sorted order from the back; k = how many divisions until the list holds N nodes
    int k = -1, srun = 0;
    for (int t0 = 0; t0 < m && k < 0; t0 += 64) {
        const int t = t0 + lane;
        int inc = 0;
        if (t < m) {
            const SelNode nd = in[(uint32_t)srt[m - 1 - t]];
            int cnt[4];
            const bool ok = node_kids(C, nd, cnt);
            const uint32_t inf = kids_info(cnt) | (ok ? 0u : 0x200u);   // 0x200: cannot be divided here
            einfo[t] = (uint16_t)inf;
            inc = __popc(inf & 15u);
            if (!ok) inc = 1;   // (counts as "no change"; if it is among the divided ones the level goes to the host)
        }
        int itot;
        const int sin = sel_scan3(inc, itot) - (lane + 1);   // a division replaces one node by its non-empty children
        const unsigned long long reach = __ballot(t < m && n + srun + sin >= N);
        if (reach) k = t0 + (int)__builtin_ctzll(reach) + 1;
        srun += itot - min(64, m - t0);
    }
    if (k < 0) k = m;
    sel_sync();
    int Hc = 0, badk = 0;
    for (int t0 = 0; t0 < k; t0 += 64) {
        const int t = t0 + lane;
        const uint32_t inf = t < k ? einfo[t] : 0u;
        if (inf & 0x200u) badk = 1;
        int tot;
        (void)sel_scan3(__popc(inf & 15u), tot);
        Hc += tot;
    }
    if (__ballot(badk != 0) != 0ull || Hc + n - k > cap) { fb = 1; mOut = 0; return n; }
    int crun = 0, erun = 0;
    for (int t0 = 0; t0 < k; t0 += 64) {
        const int t = t0 + lane;
        const uint32_t inf = t < k ? einfo[t] : 0u;
        const uint32_t mask = inf & 15u, emask = (inf >> 4) & 15u;
        const int nc = __popc(mask), ne = __popc(emask);
        int ctot, etot;
        const int cin = sel_scan3(nc, ctot), ein = sel_scan3(ne, etot);
        if (t < k) {
            const uint32_t pos0 = (uint32_t)srt[m - 1 - t];
            const SelNode nd = in[pos0];
            divided[pos0] = 1;
            int cnt[4];
            (void)node_kids(C, nd, cnt);
            const int cincl = crun + cin, eexcl = erun + ein - ne;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!((mask >> q) & 1u)) continue;
                const SelNode ch = node_child(nd, q, cnt[q]);
                const int pos = Hc - cincl + __popc(mask >> (q + 1));
                out[pos] = ch;
                if ((emask >> q) & 1u)
                    exp[eexcl + __popc(emask & ((1u << q) - 1u))] = ((uint64_t)(((uint32_t)cnt[q] << 12) | (ch.y & 0xffffu)) << 32) | (uint32_t)pos;
            }
        }
        crun += ctot;
        erun += etot;
    }
    sel_sync();
    int trun = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {   // the nodes that were not divided keep their order behind the new children
        const int i = c0 + lane;
        const bool keep = i < n && divided[i] == 0;
        const unsigned long long b = __ballot(keep);
        if (keep) out[Hc + sel_rank(b, trun)] = in[i];
        trun += __popcll(b);
    }
    sel_sync();
    mOut = erun;
    return Hc + trun;
}

}  // namespace

// grid (images, levels), 64 threads.  out_val[(img * nlevels + level) * selcap + i]: the retained candidates (packed y | x | response)
// in DistributeOctTree's result order; out_cnt[img * nlevels + level]: how many, or -1 = this level needs the host stage.
__global__ __launch_bounds__(64) void k_select(const int *__restrict__ tbl, const uint32_t *__restrict__ sorted, Geom g, uint32_t *__restrict__ out_val, int *__restrict__ out_cnt,
                                               int selcap, int ldsB, int ldsCap, int deepCap, int *__restrict__ fallback, unsigned long long *__restrict__ prof)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sel_sh[];
    const int lane = sel_lane();
    const int img = blockIdx.x, level = blockIdx.y;
    int prof_i = 0;
    auto stamp = [&]() {   // (MCORB_SELECT_PROF: shader-clock stamps of image 0's waves at the phase boundaries)
        if (prof && img == 0 && lane == 0 && prof_i < 32) prof[level * 32 + prof_i++] = __builtin_amdgcn_s_memtime();
    };
    stamp();
    const LevelGeom &L = g.lv[level];
    const int N = L.quota, D = L.depth, B = L.nBuckets, cap = ldsCap;
    const int *tb = tbl + (size_t)img * tbl_ints(g.bucketTotal);
    const int ncand = tb[kTblLvlOff + level + 1] - tb[kTblLvlOff + level];
    int *ocnt = out_cnt + (size_t)img * g.nlevels + level;
    uint32_t *oval = out_val + ((size_t)img * g.nlevels + level) * selcap;
    if (ncand <= 0) {
        if (lane == 0) *ocnt = 0;
        return;
    }
    SelLds S;
    {
        uint8_t *p = sel_sh;
        S.bst = reinterpret_cast<int *>(p); p += (size_t)(ldsB + 1 + 3) / 4 * 16;
        S.win = reinterpret_cast<const uint2 *>(p); p += (size_t)(ldsB + 1) / 2 * 16;
        S.list[0] = reinterpret_cast<SelNode *>(p); p += (size_t)cap * 16;
        S.list[1] = reinterpret_cast<SelNode *>(p); p += (size_t)cap * 16;
        S.exp[0] = reinterpret_cast<uint64_t *>(p); p += (size_t)cap * 8;
        S.exp[1] = reinterpret_cast<uint64_t *>(p); p += (size_t)cap * 8;
        S.blk = reinterpret_cast<uint32_t *>(p); p += (size_t)cap * 4;
        S.ta = reinterpret_cast<uint16_t *>(p); p += (size_t)cap * 2;
        S.tb = reinterpret_cast<uint16_t *>(p); p += (size_t)cap * 2;
        S.stk = reinterpret_cast<int *>(p);
    }
    SelCtx C;
    C.bst = S.bst;
    C.cand = sorted + (size_t)img * g.candCap + tb[kTblLvlOff + level];
    C.D = D;
    {   // path codes keep the root index above 2 bits per split, in 28 bits
        int rootBits = 0;
        while ((1 << rootBits) < L.nIni) rootBits++;
        C.maxDepth = min(14, (kSelDepthShift - rootBits) / 2);
    }
    C.H0 = L.maxBorderY - kMinBorder;
    C.wCell = L.wCell > 0 ? L.wCell : (1 << 20);
    C.hCell = L.hCell > 0 ? L.hCell : (1 << 20);
    C.hX = L.hX;
    C.deepCap = deepCap;
    // the level's bucket starts and bucket winners: global -> LDS, all loads in flight at once (the winners are not needed before
    // the tree is finished)
    {
        const int *bsrc = tb + kTblHead + L.bucket0;
        const uint2 *wsrc = reinterpret_cast<const uint2 *>(tb + tbl_win_off(g.bucketTotal)) + L.bucket0;
        uint2 *wdst = const_cast<uint2 *>(S.win);
        for (int b = lane; b <= B; b += 64) S.bst[b] = bsrc[b];
        for (int b = lane; b < B; b += 64) wdst[b] = wsrc[b];
    }
    sel_sync();
    stamp();
    // root nodes (:567-600): empty ones are erased
    int n, m = 0, fb = 0, cur = 0;
    {
        const int i = lane;
        bool keep = false;
        SelNode nd{0, 0, 0, 0};
        if (i < L.nIni) {
            nd.x = (uint32_t)i;
            nd.y = (uint32_t)(int)__fmul_rn(L.hX, (float)i) | ((uint32_t)(int)__fmul_rn(L.hX, (float)(i + 1)) << 16);
            nd.z = (uint32_t)(S.bst[(i + 1) << (2 * D)] - S.bst[i << (2 * D)]);
            keep = nd.z > 0;
        }
        const unsigned long long b = __ballot(keep);
        if (keep) S.list[0][sel_rank(b)] = nd;
        n = __popcll(b);
    }
    sel_sync();
    if (L.nIni > 64 || L.nIni < 1 || N + 8 > cap || C.maxDepth < D) fb = 1;
    bool finish = fb != 0;
    while (!finish) {
        const int prev = n;
        n = full_pass(S, C, S.list[cur], n, S.list[cur ^ 1], S.exp[0], cap, m, fb);
        stamp();
        if (fb) break;
        cur ^= 1;
        if (n >= N || n == prev) break;
        if (n + 3 * m > N) {
            int ecur = 0;
            while (true) {
                const int prev2 = n;
                wave_std_sort(S.exp[ecur], m, S.exp[ecur ^ 1], S);
                stamp();
                int m2 = 0;
                n = careful_round(S, C, S.list[cur], n, S.list[cur ^ 1], S.exp[ecur ^ 1], m, S.exp[ecur], N, cap, m2, fb);
                stamp();
                if (fb) break;
                cur ^= 1;
                m = m2;
                if (n >= N || n == prev2) break;
            }
            break;
        }
    }
    if (fb || n > selcap) {
        if (lane == 0) { *ocnt = -1; atomicOr(fallback, 1); }
        return;
    }
    // best response per node (:757-775)
    const SelNode *list = S.list[cur];
    for (int i = lane; i < n; i += 64) oval[i] = node_best(C, S.win, list[i]);
    if (lane == 0) *ocnt = n;
    stamp();
}

// ORBextractor::operator()'s assembly (:1103-1170), one workgroup per image, one wave per level: levels in order, every level's keypoints in
// DistributeOctTree's order; keypoints whose scaled x lies in [lap0, lap1] fill the output from the back (stereo), the others from
// the front (mono).  sel[img][pos] = level | y | x (what the descriptor kernel reads), resp[img][pos] = FAST response,
// nsel[img] = total, mono[img] = monoIndex.  An image with a level the GPU could not select gets nsel = 0.
struct AssembleParams { float scale[kMaxLevels]; int lap0, lap1; };
__global__ __launch_bounds__(64 * kMaxLevels) void k_assemble(const uint32_t *__restrict__ sel_val, const int *__restrict__ sel_cnt, Geom g, AssembleParams P,
                                                               int selcap, uint32_t *__restrict__ sel, uint8_t *__restrict__ resp, int *__restrict__ nsel,
                                                               int *__restrict__ mono, int *__restrict__ fallback, uint32_t *__restrict__ sel_h,
                                                               uint8_t *__restrict__ resp_h, unsigned long long *sig_h)
{
    // sel_h, resp_h, sig_h (small batches, one rig frame at a time): the host's copies of sel / responses are written here too
    // (host-mapped memory) and sig_h[img] is set behind them, so that the host builds its keypoint records while the descriptor and
    // matching kernels still run: no result copies at the end of the job.  The signal word carries everything else the host
    // needs (sel_signal(), mcorb_kernels.h: done, fallback code, keypoint count, monoIndex, and a checksum of the sel / response
    // values sent) and is written with a system-scope atomic exchange behind a system-scope fence: over PCIe an atomic is a
    // non-posted request, which may not pass the posted writes in front of it, whereas a plain store behind the same fence was seen by
    // the host BEFORE the count written just ahead of it (once in ~10^4 jobs: scripts/stress_consistency.py).  The checksum makes
    // the hand-off self-checking whatever the fabric does: a host that reads values the word does not vouch for redoes the image's
    // records after the job's end event.
    // one workgroup per image, one wave per level: every wave counts its level's stereo keypoints, the counts of the levels before
    // it give its first mono / stereo position
    __shared__ int s_cnt[kMaxLevels], s_st[kMaxLevels];
    __shared__ uint32_t s_chk[kMaxLevels];   // (its own array: a slow wave may still be reading s_st when a fast one gets here)
    const int lane = sel_lane(), level = threadIdx.x >> 6, img = blockIdx.x;
    const int c = sel_cnt[(size_t)img * g.nlevels + level];
    const uint32_t *v = sel_val + ((size_t)img * g.nlevels + level) * selcap;
    const float sc = P.scale[level], lo = (float)P.lap0, hi = (float)P.lap1;
    auto is_stereo = [&](uint32_t cd) {
        float kx = (float)(cand_x(cd) + kMinBorder);
        if (level != 0) kx = __fmul_rn(kx, sc);
        return kx >= lo && kx <= hi;
    };
    // the level's first 64 * kAsmRegs winners stay in registers between the counting and the placing pass (all loads in flight at
    // once: one workgroup per image is latency, not bandwidth); longer lists go on from memory
    constexpr int kAsmRegs = 8;
    uint32_t vr[kAsmRegs];
#pragma unroll
    for (int u = 0; u < kAsmRegs; u++) vr[u] = u * 64 + lane < c ? v[u * 64 + lane] : 0u;
    int nst = 0;
    if (hi >= (float)kMinBorder) {   // (no keypoint lies left of the border: the default lapping area (0, 0) has no stereo keypoints)
#pragma unroll
        for (int u = 0; u < kAsmRegs; u++) nst += __popcll(__ballot(u * 64 + lane < c && is_stereo(vr[u])));
        for (int c0 = 64 * kAsmRegs; c0 < c; c0 += 64) {
            const int i = c0 + lane;
            nst += __popcll(__ballot(i < c && is_stereo(v[min(i, c - 1)])));
        }
    }
    if (lane == 0) { s_cnt[level] = c; s_st[level] = nst; }
    __syncthreads();
    int total = 0, bad = 0, monoBefore = 0, stereoBefore = 0, monoAll = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int cl = s_cnt[l];
        if (cl < 0) { bad = 1; continue; }
        total += cl;
        monoAll += cl - s_st[l];
        if (l < level) { monoBefore += cl - s_st[l]; stereoBefore += s_st[l]; }
    }
    if (!bad && total > g.kcap) {
        bad = 2;
        if (threadIdx.x == 0) atomicOr(fallback, 2);
    }
    if (bad) {   // a level the GPU could not select (1) or too many keypoints (2): the host stage redoes the batch
        if (threadIdx.x == 0) {
            nsel[img] = 0; mono[img] = 0;
            if (sig_h) __hip_atomic_exchange(&sig_h[img], sel_signal(bad, 0, 0, 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    uint32_t *so = sel + (size_t)img * g.kcap;
    uint8_t *ro = resp + (size_t)img * g.kcap;
    int monoIndex = monoBefore, stereoIndex = total - 1 - stereoBefore;
    uint32_t chk = 0;   // XOR of what this lane sends to the host (sel_check): the host recomputes it over what it reads
    auto place = [&](int i, uint32_t cd) {
        const bool valid = i < c;
        const bool stereo = valid && is_stereo(cd);
        const unsigned long long bs = __ballot(stereo), bm = __ballot(valid && !stereo);
        if (valid) {
            const int pos = stereo ? stereoIndex - sel_rank(bs) : monoIndex + sel_rank(bm);
            const uint32_t ps = pack_sel(level, cand_x(cd) + kMinBorder, cand_y(cd) + kMinBorder);
            so[pos] = ps;
            ro[pos] = (uint8_t)cand_resp(cd);
            if (sel_h) {
                sel_h[(size_t)img * g.kcap + pos] = ps;
                resp_h[(size_t)img * g.kcap + pos] = (uint8_t)cand_resp(cd);
                chk ^= sel_check(ps, (uint8_t)cand_resp(cd), pos);
            }
        }
        stereoIndex -= __popcll(bs);
        monoIndex += __popcll(bm);
    };
#pragma unroll
    for (int u = 0; u < kAsmRegs; u++)
        if (u * 64 < c) place(u * 64 + lane, vr[u]);
    for (int c0 = 64 * kAsmRegs; c0 < c; c0 += 64) place(c0 + lane, c0 + lane < c ? v[c0 + lane] : 0u);
    if (threadIdx.x == 0) { nsel[img] = total; mono[img] = monoAll; }
    if (sig_h) {   // every wave's host writes are out before the image is signalled
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) chk ^= __shfl_xor(chk, o);
        if (lane == 0) s_chk[level] = chk;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t x = 0;
            for (int l = 0; l < g.nlevels; l++) x ^= s_chk[l];
            __hip_atomic_exchange(&sig_h[img], sel_signal(0, total, monoAll, x), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// --- launch wrappers -------------------------------------------------------------------------------------------------------------

int select_cap(const Geom &g)   // retained candidates per (image, level): DistributeOctTree stops at N .. N + 2 nodes
{
    int q = 1;
    for (int l = 0; l < g.nlevels; l++) q = q > g.lv[l].quota ? q : g.lv[l].quota;
    return (q + 64 + 63) & ~63;
}

static size_t select_lds(const Geom &g, int &B)
{
    B = 1;
    for (int l = 0; l < g.nlevels; l++) B = B > g.lv[l].nBuckets ? B : g.lv[l].nBuckets;
    return sel_lds_bytes(B, select_cap(g));
}

// k_select keeps a level's tree, its sort scratch and its bucket tables in one wave's LDS: feature budgets beyond ~2 500 per
// level (or more than 64 root nodes) do not fit 160 KiB -- such rigs select on the host
bool select_fits(const Geom &g)
{
    int B;
    for (int l = 0; l < g.nlevels; l++)
        if (g.lv[l].nIni > 64) return false;
    return select_cap(g) <= 65535 && select_lds(g, B) <= 160 * 1024;
}

hipError_t launch_select(hipStream_t st, const int *tbl, const uint32_t *sorted, const Geom &g, uint32_t *sel_val, int *sel_cnt, int *fallback, int nimg, int deep_cap)
{
    int B;
    const int cap = select_cap(g);
    const size_t lds = select_lds(g, B);
    if (!select_fits(g)) return hipErrorInvalidValue;
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    static unsigned long long *prof = nullptr;
    static const bool prof_on = getenv("MCORB_SELECT_PROF") != nullptr;
    if (prof_on && !prof) (void)hipHostMalloc((void **)&prof, kMaxLevels * 32 * 8, hipHostMallocMapped);
    if (prof_on && prof) {
        static int calls = 0;
        if (++calls == 40) {   // a steady-state job: print the previous launch's stamps (cycles between phase boundaries)
            (void)hipStreamSynchronize(st);
            for (int l = 0; l < g.nlevels; l++) {
                fprintf(stderr, "[select prof] level %d:", l);
                for (int i = 1; i < 32 && prof[l * 32 + i]; i++) fprintf(stderr, " %llu", prof[l * 32 + i] - prof[l * 32 + i - 1]);
                fprintf(stderr, "\n");
            }
        }
        for (int i = 0; i < kMaxLevels * 32; i++) prof[i] = 0;
    }
    hipLaunchKernelGGL(k_select, dim3(nimg, g.nlevels), dim3(64), lds, st, tbl, sorted, g, sel_val, sel_cnt, cap, B, cap, deep_cap, fallback, prof_on ? prof : nullptr);
    return hipGetLastError();
}

void launch_assemble(hipStream_t st, const uint32_t *sel_val, const int *sel_cnt, const Geom &g, const float *scale, int lap0, int lap1,
                     uint32_t *sel, uint8_t *resp, int *nsel, int *mono, int *fallback, int nimg, uint32_t *sel_h, uint8_t *resp_h,
                     unsigned long long *sig_h)
{
    AssembleParams P;
    for (int l = 0; l < kMaxLevels; l++) P.scale[l] = l < g.nlevels ? scale[l] : 1.f;
    P.lap0 = lap0; P.lap1 = lap1;
    hipLaunchKernelGGL(k_assemble, dim3(nimg), dim3(64 * g.nlevels), 0, st, sel_val, sel_cnt, g, P, select_cap(g), sel, resp, nsel, mono, fallback, sel_h, resp_h, sig_h);
}

// test hook (mcorb_dev_sort_selftest): one wave sorts n entries with wave_std_sort
__global__ __launch_bounds__(64) void k_sort_selftest(const uint64_t *__restrict__ in, int n, uint64_t *__restrict__ out, int cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sel_sh[];
    SelLds S;
    uint8_t *p = sel_sh;
    S.bst = reinterpret_cast<int *>(p); p += 16;
    S.list[0] = S.list[1] = nullptr;
    S.exp[0] = reinterpret_cast<uint64_t *>(p); p += (size_t)cap * 8;
    S.exp[1] = reinterpret_cast<uint64_t *>(p); p += (size_t)cap * 8;
    S.blk = reinterpret_cast<uint32_t *>(p); p += (size_t)cap * 4;
    S.ta = reinterpret_cast<uint16_t *>(p); p += (size_t)cap * 2;
    S.tb = reinterpret_cast<uint16_t *>(p); p += (size_t)cap * 2;
    S.stk = reinterpret_cast<int *>(p);
    for (int i = sel_lane(); i < n; i += 64) S.exp[0][i] = in[i];
    sel_sync();
    wave_std_sort(S.exp[0], n, S.exp[1], S);
    for (int i = sel_lane(); i < n; i += 64) out[i] = S.exp[1][i];
}

hipError_t sort_selftest(const uint64_t *in_dev, int n, uint64_t *out_dev)
{
    const int cap = (n + 63) & ~63;
    const size_t lds = 16 + (size_t)cap * (8 + 8 + 4 + 2 + 2) + kSelStack * 4;
    if (n < 0 || cap > 6000 || lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_sort_selftest, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_sort_selftest, dim3(1), dim3(64), lds, 0, in_dev, n, out_dev, cap);
    return hipDeviceSynchronize();
}

}  // namespace mcorb
