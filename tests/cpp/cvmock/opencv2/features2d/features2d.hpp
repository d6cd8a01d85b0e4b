// cvmock (syntax only, see ../core/core.hpp): cv::KeyPoint lives in core.hpp here; FAST / BFMatcher are DECLARED for
// tools/crosscheck/crosscheck_opencv.cpp's syntax check and have no bodies: nothing can link or run against them.
#pragma once
#include <vector>

#include "../core/core.hpp"

namespace cv {
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
};
void FAST(InputArray image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true);
class BFMatcher {
public:
    explicit BFMatcher(int normType = 4, bool crossCheck = false);
    void knnMatch(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<std::vector<DMatch>> &matches, int k) const;
};
}  // namespace cv
