// cvmock (syntax only, see ../core/core.hpp): DECLARATIONS of the three imgproc calls tools/crosscheck/crosscheck_opencv.cpp
// makes; no bodies -- nothing can link or run against them, and no parity claim may cite them.
#pragma once
#include "../core/core.hpp"

namespace cv {
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_REFLECT_101);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType);
}  // namespace cv
