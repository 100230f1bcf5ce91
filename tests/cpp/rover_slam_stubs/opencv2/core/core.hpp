#pragma once
#include "rfe/cv_compat.h"
#if !RFE_HAVE_OPENCV
namespace cv { struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0; }; }
#endif
