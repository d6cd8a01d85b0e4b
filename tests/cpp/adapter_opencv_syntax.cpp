// Compiles the -DMCORB_WITH_OPENCV flavour of include/mcorb_adapter.hpp against tests/cpp/cvmock (this repository's own
// stand-in for the OpenCV declarations it touches).  SYNTAX ONLY: it shows that the code a maintainer would compile is
// well-formed C++ against those signatures; it pins nothing about OpenCV.  The one thing that can run without a GPU --
// setCalibration's cv::Mat algebra -- is executed on a made-up rig and printed, so the Python test can compare it with
// numpy float64 (the mock's matrix routines are plain double loops; again: not OpenCV's).
#define MCORB_WITH_OPENCV 1
#include "mcorb_adapter.hpp"

#include <stdio.h>

int main()
{
    // instantiate what the flavour adds (never executed: no GPU in the CPU suite)
    int (mcorb::ORBextractor::*op)(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint> &, cv::OutputArray, std::vector<int> &) =
        &mcorb::ORBextractor::operator();
    int (mcorb::ORBextractor::*dd)(const cv::Mat &, const cv::Mat &) = &mcorb::ORBextractor::DescriptorDistance;
    void (mcorb::MultiCameraFrontEnd::*sc)(const std::vector<cv::Mat> &, const std::vector<cv::Mat> &, const std::vector<cv::Mat> &) =
        &mcorb::MultiCameraFrontEnd::setCalibration;
    (void)op; (void)dd; (void)sc;
    printf("adapter_opencv_syntax ok\n");
    return 0;
}
