#pragma once
// stand-in: the members the matcher reads (reference include/Frame.h has them among many others)
#include <vector>
#include "rfe/cv_compat.h"
namespace ORB_SLAM3 {
class MapPoint;
class KeyFrame;
class Frame {
public:
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors;
    cv::Mat imgLeft;
};
}
