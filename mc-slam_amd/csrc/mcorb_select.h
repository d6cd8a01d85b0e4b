// mcorb_select.h -- host quad-tree keypoint selection stage (see mcorb_select.cpp).
#pragma once
#include <stdint.h>
#include <vector>

#include "mcorb_common.h"

namespace mcorb {

struct SelectScratch {
    struct Impl;
    Impl *impl;
    SelectScratch();
    ~SelectScratch();
    SelectScratch(const SelectScratch &) = delete;
    SelectScratch &operator=(const SelectScratch &) = delete;
};

// cand: packed candidates of one (image, level) in vToDistributeKeys order.
// out_idx: indices of the retained candidates in the reference's result order;
// needs room for N + 4 entries.  Returns the count, or -2 if the level is too
// tall for a root node (the reference divides by zero there).
int select_octree(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N, int *out_idx,
                  SelectScratch &scratch);

}  // namespace mcorb
