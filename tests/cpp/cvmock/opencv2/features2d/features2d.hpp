// cvmock (syntax only, see ../core/core.hpp): cv::KeyPoint lives in core.hpp here.
#pragma once
#include "../core/core.hpp"
