// SPmatcher.h -- the LightGlue half of ORB_SLAM3::SPmatcher (reference include/Matchers/SPmatcher.h:48-63,
// src/Matchers/SPmatcher.cc:17-27, :359-542): constructor, the four MatchingPoints_onnx overloads and
// DescriptorDistance_sp, forwarding to librover_fe.so.  The classic projection / BoW searches of the
// reference class (SPmatcher.cc:49-357, :695-2189, CPU L2 loops) are out of scope.
//
// Quirk kept on purpose: three of the four overloads normalise keypoints with a hard-coded 300 x 400 image
// size (SPmatcher.cc:360-361, 376-377, 414-415); only the Frame overload uses the real size (:463-464).
// Define RFE_COMPAT_NORMALIZE=0 to pass a true size via SetImageSize() instead.
#ifndef SPMATCHER_H
#define SPMATCHER_H
#include <cmath>
#include <vector>
#include "lightglue_onnx.h"

#ifndef RFE_COMPAT_NORMALIZE
#define RFE_COMPAT_NORMALIZE 1
#endif

namespace ORB_SLAM3 {

class SPmatcher {
public:
    explicit SPmatcher(float thre = 0.0f) {
        Configuration cfg;
        cfg.device = "cuda";
        featureMatcher = new LightGlueDecoupleOnnxRunner();
        featureMatcher->InitOrtEnv(cfg);           // return value ignored, as in SPmatcher.cc:24
        featureMatcher->SetMatchThresh(thre);
    }
    ~SPmatcher() { delete featureMatcher; }
    SPmatcher(const SPmatcher&) = delete;
    SPmatcher& operator=(const SPmatcher&) = delete;

    static float DescriptorDistance_sp(const cv::Mat& a, const cv::Mat& b) {   // SPmatcher.cc:2184-2189
        const float* pa = a.ptr<float>(0); const float* pb = b.ptr<float>(0);
        double s = 0;
        for (int i = 0; i < a.cols; ++i) { const double d = (double)pa[i] - pb[i]; s += d * d; }
        return (float)std::sqrt(s);
    }

    void SetImageSize(int rows, int cols) { rows_ = rows; cols_ = cols; }

    // SPmatcher.cc:359-371
    int MatchingPoints_onnx(std::vector<cv::Point2f> kpts0, std::vector<cv::Point2f> kpts1, float* desc0, float* desc1) {
        std::vector<int> vn(kpts0.size(), -1);
        return run(kpts0, kpts1, desc0, desc1, rows_, cols_, vn);
    }
    // SPmatcher.cc:374-410
    int MatchingPoints_onnx(std::vector<cv::Point2f> kpts0, std::vector<cv::Point2f> kpts1, cv::Mat desc0, cv::Mat desc1, std::vector<int>& vnMatches12) {
        vnMatches12.resize(kpts0.size(), -1);
        std::vector<float> d0 = pack(desc0), d1 = pack(desc1);
        return run(kpts0, kpts1, d0.data(), d1.data(), rows_, cols_, vnMatches12);
    }
    // SPmatcher.cc:412-454
    int MatchingPoints_onnx(std::vector<cv::KeyPoint> kpts0, const std::vector<cv::KeyPoint> kpts1, cv::Mat desc0, const cv::Mat desc1, std::vector<int>& vnMatches12) {
        vnMatches12.resize(kpts0.size(), -1);
        std::vector<cv::Point2f> p0, p1;
        for (const cv::KeyPoint& k : kpts0) p0.emplace_back(k.pt);
        for (const cv::KeyPoint& k : kpts1) p1.emplace_back(k.pt);
        std::vector<float> d0 = pack(desc0), d1 = pack(desc1);
        return run(p0, p1, d0.data(), d1.data(), rows_, cols_, vnMatches12);
    }
    // SPmatcher.cc:457-542 ; FrameT needs mvKeys, mDescriptors, imgLeft (ORB_SLAM3::Frame has them)
    template <class FrameT>
    int MatchingPoints_onnx(FrameT& f1, FrameT& f2, std::vector<int>& vnMatches12) {
        vnMatches12.resize(f1.mvKeys.size(), -1);
        std::vector<cv::Point2f> p0, p1;
        for (const cv::KeyPoint& k : f1.mvKeys) p0.emplace_back(k.pt);
        for (const cv::KeyPoint& k : f2.mvKeys) p1.emplace_back(k.pt);
        std::vector<float> d0 = pack(f1.mDescriptors), d1 = pack(f2.mDescriptors);
        return run(p0, p1, d0.data(), d1.data(), f2.imgLeft.rows, f2.imgLeft.cols, vnMatches12);
    }

    static const float TH_LOW;
    static const float TH_HIGH;
    static const int HISTO_LENGTH;
    LightGlueDecoupleOnnxRunner* featureMatcher;

private:
    int rows_ = 300, cols_ = 400;   // "需要修改" in the reference; see header comment
    static std::vector<float> pack(const cv::Mat& d) {
        std::vector<float> v((size_t)d.rows * d.cols);
        for (int r = 0; r < d.rows; ++r) { const float* s = d.ptr<float>(r); std::copy(s, s + d.cols, v.begin() + (size_t)r * d.cols); }
        return v;
    }
    int run(const std::vector<cv::Point2f>& k0, const std::vector<cv::Point2f>& k1, float* d0, float* d1, int rows, int cols, std::vector<int>& vn) {
        auto n0 = featureMatcher->Matcher_PreProcess(k0, rows, cols);
        auto n1 = featureMatcher->Matcher_PreProcess(k1, rows, cols);
        std::vector<rfe::Tensor> output = featureMatcher->Matcher_Inference(n0, n1, d0, d1);
        return featureMatcher->Matcher_PostProcess_fused(output, k0, k1, vn);
    }
};
inline const float SPmatcher::TH_HIGH = 1.4f;   // SPmatcher.cc:13-15
inline const float SPmatcher::TH_LOW = 1.2f;
inline const int SPmatcher::HISTO_LENGTH = 30;

}  // namespace ORB_SLAM3
#endif
