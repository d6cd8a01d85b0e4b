// mcorb_select.h -- host quad-tree keypoint selection stage (see mcorb_select.cpp).
#pragma once
#include <stdint.h>
#include <vector>

#include "mcorb_common.h"

namespace mcorb {

struct SelectScratch {
    struct Impl;
    Impl *impl;
    std::vector<int> idx;   // caller's index buffer (select_octree's out_idx)
    SelectScratch();
    ~SelectScratch();
    SelectScratch(const SelectScratch &) = delete;
    SelectScratch &operator=(const SelectScratch &) = delete;
};

// per-level constants of DistributeOctTree's call (ORBextractor.cpp:876-877) plus the bucketing depth
struct SelectParams {
    int minX, maxX, minY, maxY;   // (:788-791)
    int N;                        // mnFeaturesPerLevel[level]
    int nIni;                     // (:558)
    float hX;                     // (:560)
    int depth;                    // path-code depth used by k_compact for this level
    int wCell, hCell;             // cell grid of the detection loop: recovers vToDistributeKeys order from (x,y)
};
SelectParams make_select_params(int minX, int maxX, int minY, int maxY, int N, int wCell, int hCell);

// cand: one level's candidates sorted by path code (any order inside a bucket); bstart: the
// nIni*4^depth + 1 bucket start offsets.  out_idx: indices INTO cand of the retained candidates
// in the reference's result order; needs room for N + 64 entries.  Returns the count, or -2 if
// the level is too tall for a root node (the reference divides by zero there).
// out_val: the retained candidates themselves (packed).  cand may be null when the level's list was not shipped
// (k_compact ships it only if fewer than N buckets are non-empty); -3 if the tree then wanted to go deeper.
// BB = BucketWin (what the GPU ships: key + candidate) or BucketBest (host statement, also the position; out_idx of a
// bucket-resolved node is -1 with BucketWin).
template <typename BB>
int select_octree(const uint32_t *cand, const int *bstart, const BB *bbest, int n, const SelectParams &P, int *out_idx,
                  uint32_t *out_val, SelectScratch &scratch);

// host statement of k_compact's counting sort for one level (test hook only)
void host_bucket_sort(const uint32_t *cand, int n, const SelectParams &P, std::vector<uint32_t> &sorted,
                      std::vector<int> &perm, std::vector<int> &bstart, std::vector<BucketBest> &bbest);

}  // namespace mcorb
