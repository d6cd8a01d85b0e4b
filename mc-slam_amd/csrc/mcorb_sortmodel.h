// mcorb_sortmodel.h -- libstdc++'s std::sort, restated so that a GPU wave can run it.
//
// DistributeOctTree's second phase sorts (key count, UL.x) pairs with std::sort and walks the result from the back
// (ORBextractor.cpp:697, compareNodes :537-552).  Entries that compare equal are common (small counts, nodes of one column), and
// where an unstable sort leaves them decides which nodes are divided first, i.e. which keypoints come out and in what order.  The
// selection kernel (mcorb_select_gpu.hip) therefore has to produce the permutation std::sort produces -- for libstdc++
// (bits/stl_algo.h: __introsort_loop with _S_threshold 16, median-of-three to the front, __unguarded_partition, heap sort when
// 2 * floor(log2 n) partition levels are used up, __final_insertion_sort) -- and it does so with three observations:
//
//  1. __unguarded_partition has a closed form.  With pivot p = a[first], L = the positions of (first, last) whose key is >= p in
//     ascending order and R = those whose key is <= p in descending order, the loop swaps (L_i, R_i) for i = 1 .. s, where s is the
//     number of leading i with L_i < R_i, and returns cut = min(L_{s+1}, R_s)  (L_1 when s = 0; R_s when L has no further entry):
//     both scans only ever see elements that have not been swapped yet, except for the one a scan stops on when it runs into the
//     other side's last swap.  Ranks, s and the swaps are ballots and prefix counts: one wave-parallel step per partition.
//  2. The sub-ranges of a partition are independent: the order in which they are worked off does not matter, only each range's
//     remaining depth budget.
//  3. __final_insertion_sort is a stable sort, and after the introsort loop the array is a sequence of blocks (the ranges that
//     ended at <= 16 elements, or were heap-sorted) with every element of a block <= every element of the next: the final position
//     of an element is its block's start plus its stable rank inside the block.
//
// This header is the HOST statement of exactly those steps (sort_model), checked against std::sort itself on random multisets full
// of ties and on median-of-three killer sequences that force the heap-sort branch (tests/cpp/test_sortmodel.cpp, run by the CPU
// suite), plus the sequential heap sort both sides share.  Entries are 64-bit: key in the upper half, payload in the lower; only the
// key is compared.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MCORB_HD __host__ __device__
#else
#define MCORB_HD
#endif

namespace mcorb {

MCORB_HD inline uint32_t sm_key(uint64_t e) { return (uint32_t)(e >> 32); }

// libstdc++'s __adjust_heap + __push_heap (bits/stl_heap.h) on a[first .. first + len), comparing keys
MCORB_HD inline void sm_adjust_heap(uint64_t *a, int first, int hole, int len, uint64_t value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (sm_key(a[first + child]) < sm_key(a[first + child - 1])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && sm_key(a[first + parent]) < sm_key(value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

// __partial_sort(first, last, last) = __heap_select (make_heap, nothing to select) + __sort_heap
MCORB_HD inline void sm_heap_sort(uint64_t *a, int first, int last)
{
    const int len = last - first;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; parent--) {
        const uint64_t v = a[first + parent];
        sm_adjust_heap(a, first, parent, len, v);
        if (parent == 0) break;
    }
    for (int l = last; l - first > 1;) {
        --l;
        const uint64_t v = a[l];
        a[l] = a[first];
        sm_adjust_heap(a, first, 0, l - first, v);
    }
}

MCORB_HD inline int sm_lg(int n)
{
    int k = 0;
    while (n > 1) { n >>= 1; k++; }
    return k;
}

// which of a[ia], a[ib], a[ic] __move_median_to_first swaps to the front
MCORB_HD inline int sm_median3(uint32_t ka, uint32_t kb, uint32_t kc, int ia, int ib, int ic)
{
    if (ka < kb) {
        if (kb < kc) return ib;
        if (ka < kc) return ic;
        return ia;
    }
    if (ka < kc) return ia;
    if (kb < kc) return ic;
    return ib;
}

#if !defined(__HIP_DEVICE_COMPILE__)
}  // namespace mcorb
#include <utility>
#include <vector>
namespace mcorb {

// the closed form of __unguarded_partition(first + 1, last, first): performs the swaps, returns the cut
inline int sm_partition_closed_form(uint64_t *a, int first, int last)
{
    const uint32_t p = sm_key(a[first]);
    std::vector<int> L, R;
    for (int i = first + 1; i < last; i++)
        if (!(sm_key(a[i]) < p)) L.push_back(i);
    for (int i = last - 1; i > first; i--)
        if (!(p < sm_key(a[i]))) R.push_back(i);
    size_t s = 0;
    while (s < L.size() && s < R.size() && L[s] < R[s]) s++;
    for (size_t i = 0; i < s; i++) std::swap(a[L[i]], a[R[i]]);
    const int inf = 1 << 30;
    const int nextL = s < L.size() ? L[s] : inf, lastR = s > 0 ? R[s - 1] : inf;
    return nextL < lastR ? nextL : lastR;
}

// std::sort(a, a + n, key <) as the selection kernel runs it: same permutation as libstdc++'s, entry for entry
inline void sort_model(uint64_t *a, int n, int *heap_ranges = nullptr)   // heap_ranges (tests): how many ranges took the heap-sort branch
{
    if (heap_ranges) *heap_ranges = 0;
    if (n < 2) return;
    struct Range { int f, l, dl; };
    std::vector<Range> stack, blocks;
    stack.push_back({0, n, 2 * sm_lg(n)});
    while (!stack.empty()) {
        Range r = stack.back();
        stack.pop_back();
        bool heap = false;
        while (r.l - r.f > 16) {
            if (r.dl == 0) { sm_heap_sort(a, r.f, r.l); heap = true; if (heap_ranges) ++*heap_ranges; break; }
            r.dl--;
            const int mid = r.f + (r.l - r.f) / 2;
            const int m = sm_median3(sm_key(a[r.f + 1]), sm_key(a[mid]), sm_key(a[r.l - 1]), r.f + 1, mid, r.l - 1);
            std::swap(a[r.f], a[m]);
            const int cut = sm_partition_closed_form(a, r.f, r.l);
            stack.push_back({cut, r.l, r.dl});
            r.l = cut;
        }
        (void)heap;
        blocks.push_back({r.f, r.l, 0});
    }
    // final insertion sort = stable rank inside each block
    std::vector<uint64_t> out(a, a + n);
    for (const Range &b : blocks)
        for (int i = b.f; i < b.l; i++) {
            int rank = 0;
            for (int j = b.f; j < b.l; j++)
                rank += sm_key(a[j]) < sm_key(a[i]) || (sm_key(a[j]) == sm_key(a[i]) && j < i);
            out[b.f + rank] = a[i];
        }
    for (int i = 0; i < n; i++) a[i] = out[i];
}
#endif

}  // namespace mcorb
