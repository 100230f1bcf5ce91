// transform.h -- NormalizeImage / NormalizeKeypoints / RGB2Grayscale of the reference
// (src/Matchers/transform.cpp:3-32, 85-90; declarations include/Matchers/transform.h).  NormalizeImage's arithmetic for
// the 8-bit grayscale frames of the live path (u8 * 1/255) is fused into the first HIP convolution; the host versions
// here serve callers that use them directly (SPextractor.cc:596-597 calls NormalizeImage on a clone).
// With OpenCV present cv::cvtColor / convertTo do the work exactly as in the reference; without it (this build image)
// the same arithmetic is spelled out on the POD mirror of rfe/cv_compat.h.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>
#include "../rfe/cv_compat.h"
#if RFE_HAVE_OPENCV
#include <opencv2/imgproc.hpp>
#endif

// transform.cpp:3-17: 3 channels -> BGR2RGB then * 1/255 (CV_32FC3); 1 channel -> * 1/255 (CV_32FC1); else throws
inline cv::Mat NormalizeImage(cv::Mat& Image) {
#if RFE_HAVE_OPENCV
    cv::Mat normalizedImage = Image.clone();
    if (Image.channels() == 3) {
        cv::cvtColor(normalizedImage, normalizedImage, cv::COLOR_BGR2RGB);
        normalizedImage.convertTo(normalizedImage, CV_32F, 1.0 / 255.0);
    } else if (Image.channels() == 1) {
        Image.convertTo(normalizedImage, CV_32F, 1.0 / 255.0);
    } else {
        throw std::invalid_argument("[ERROR] Not an image");
    }
    return normalizedImage;
#else
    const int ch = Image.channels();
    if ((ch != 1 && ch != 3) || Image.depth() != CV_8U) throw std::invalid_argument("[ERROR] Not an image");
    cv::Mat out(Image.rows, Image.cols, ch == 3 ? CV_32FC3 : CV_32F);
    for (int r = 0; r < Image.rows; ++r) {
        const unsigned char* s = Image.ptr<unsigned char>(r);
        float* d = out.ptr<float>(r);
        for (int c = 0; c < Image.cols; ++c)
            for (int k = 0; k < ch; ++k)   // convertTo(CV_32F, 1/255) = fp32 product v * (float)(1/255.) (OpenCV's cvtScale8u32f), as in the HIP kernel; BGR -> RGB first
                d[c * ch + k] = (float)s[c * ch + (ch == 3 ? 2 - k : k)] * (float)(1.0 / 255.0);
    }
    return out;
#endif
}

// transform.cpp:85-90: cv::cvtColor(RGB2GRAY).  8-bit: OpenCV's 14-bit fixed point (R 4899, G 9617, B 1868, + 8192 >> 14);
// float: 0.299 R + 0.587 G + 0.114 B.
inline cv::Mat RGB2Grayscale(cv::Mat& Image) {
#if RFE_HAVE_OPENCV
    cv::Mat resultImage;
    cv::cvtColor(Image, resultImage, cv::COLOR_RGB2GRAY);
    return resultImage;
#else
    if (Image.channels() != 3) throw std::invalid_argument("[ERROR] RGB2Grayscale needs a 3-channel image");
    const bool f32 = Image.depth() == CV_32F;
    cv::Mat out(Image.rows, Image.cols, f32 ? CV_32F : CV_8U);
    for (int r = 0; r < Image.rows; ++r)
        for (int c = 0; c < Image.cols; ++c) {
            if (f32) {
                const float* s = Image.ptr<float>(r) + 3 * c;
                out.ptr<float>(r)[c] = s[0] * 0.299f + s[1] * 0.587f + s[2] * 0.114f;
            } else {
                const unsigned char* s = Image.ptr<unsigned char>(r) + 3 * c;
                out.ptr<unsigned char>(r)[c] = (unsigned char)((s[0] * 4899 + s[1] * 9617 + s[2] * 1868 + 8192) >> 14);
            }
        }
    return out;
#endif
}

// transform.cpp:19-32
inline std::vector<cv::Point2f> NormalizeKeypoints(std::vector<cv::Point2f> kpts, int h, int w) {
    const cv::Point2f shift(static_cast<float>(w) / 2, static_cast<float>(h) / 2);
    const float scale = static_cast<float>((std::max)(w, h)) / 2;
    std::vector<cv::Point2f> out;
    out.reserve(kpts.size());
    for (size_t i = 0; i < kpts.size(); ++i) out.push_back((kpts[i] - shift) / scale);
    return out;
}
