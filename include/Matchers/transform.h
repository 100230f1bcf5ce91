// transform.h -- NormalizeImage / NormalizeKeypoints / RGB2Grayscale of the reference
// (src/Matchers/transform.cpp:3-32,85-90).  NormalizeImage's arithmetic (u8 * 1/255) is fused into
// the first HIP convolution; the host version here exists for callers that use it directly.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>
#include "../rfe/cv_compat.h"

inline cv::Mat NormalizeImage(cv::Mat& Image) {
    if (Image.channels() != 1) throw std::invalid_argument("[ERROR] Not an image");  // 3-channel inputs: convert first
    cv::Mat out(Image.rows, Image.cols, CV_32F);
    for (int r = 0; r < Image.rows; ++r) {
        const unsigned char* s = Image.ptr<unsigned char>(r);
        float* d = out.ptr<float>(r);
        for (int c = 0; c < Image.cols; ++c) d[c] = (float)s[c] * (float)(1.0 / 255.0);
    }
    return out;
}

inline std::vector<cv::Point2f> NormalizeKeypoints(std::vector<cv::Point2f> kpts, int h, int w) {
    const cv::Point2f shift(static_cast<float>(w) / 2, static_cast<float>(h) / 2);
    const float scale = static_cast<float>((std::max)(w, h)) / 2;
    std::vector<cv::Point2f> out;
    out.reserve(kpts.size());
    for (const cv::Point2f& k : kpts) out.push_back((k - shift) / scale);
    return out;
}
