// SPmatcher.h -- ORB_SLAM3::SPmatcher over librover_fe.so (reference include/Matchers/SPmatcher.h:44-140,
// src/Matchers/SPmatcher.cc).  C++14 (the reference builds with -std=c++14, CMakeLists.txt:12).  Two ways to use it:
//
// (1) Inside the Rover-SLAM tree, -DRFE_WITH_ROVER_SLAM (what tools/apply_integration.py sets up): this header stands in for
//     the reference's and DECLARES the complete member list of the reference class; every body -- the constructor (:17-27),
//     the four MatchingPoints_onnx overloads (:359-542), the classic projection / BoW / fuse searches (:49-357, :695-2189) and
//     the three static constants (:13-15) -- stays where it is, in the reference's own src/Matchers/SPmatcher.cc, which
//     compiles unchanged: its `Ort::Value` is rfe::Tensor (include/rfe/ort_compat/onnxruntime_cxx_api.h) and its
//     featureMatcher is the LightGlueDecoupleOnnxRunner of this repo (lightglue_onnx.h).
//
// (2) Stand-alone (default; tests, tools, other hosts): a header-only class with the constructor, DescriptorDistance_sp
//     and the four MatchingPoints_onnx overloads, same signatures and semantics; the Frame overload is a template over any
//     type with mvKeys / mDescriptors / imgLeft.  The classic searches need Frame / KeyFrame / MapPoint and are not here.
//
// Quirk kept on purpose: three of the four overloads normalise keypoints with a hard-coded 300 x 400 image size
// (SPmatcher.cc:360-361, 376-377, 414-415); only the Frame overload uses the real size (:463-464).  Stand-alone mode:
// SetImageSize() overrides it.
#ifndef ORBMATCHER_H   // same guard as the reference header: whichever comes first on the include path wins
#define ORBMATCHER_H

#ifdef RFE_WITH_ROVER_SLAM
// ------------------------------------------------------------------------------------------------------------------
#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "sophus/sim3.hpp"
#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"
#include "Matchers/lightglue_onnx.h"

namespace ORB_SLAM3 {

struct SuperGlueConfig {            // SPmatcher.h:33-41 (unused by the live code)
    int image_width;
    int image_height;
    int dla_core;
    std::vector<std::string> input_tensor_names;
    std::vector<std::string> output_tensor_names;
    std::string onnx_file;
    std::string engine_file;
};

class SPmatcher {
public:
    typedef Eigen::Matrix<double, 259, Eigen::Dynamic> Features259;

    SPmatcher(float thre);

    // ---- learned matching: LightGlue through featureMatcher (bodies: SPmatcher.cc:359-542, 1050-1080)
    int MatchingPoints_onnx(Frame& f1, Frame& f2, std::vector<int>& vnMatches12);
    int MatchingPoints_onnx(std::vector<cv::KeyPoint> kpts0, const std::vector<cv::KeyPoint> kpts1, cv::Mat desc0, const cv::Mat desc1,
                            std::vector<int>& vnMatches12);
    int MatchingPoints_onnx(std::vector<cv::Point2f> kpts0, std::vector<cv::Point2f> kpts1, float* desc0, float* desc1);
    int MatchingPoints_onnx(std::vector<cv::Point2f> kpts0, std::vector<cv::Point2f> kpts1, cv::Mat desc0, cv::Mat desc1,
                            std::vector<int>& vnMatches12);
    int SearchBySP(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchBySP(Frame& F, const std::vector<MapPoint*>& vpMapPoints);
    int SearchBySP(Frame& CurrentFrame, Frame& LastFrame);
    int Fuse_onnx(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bRight = false);

    // ---- SuperGlue-era leftovers (declared in the reference, no live definition)
    int MatchingPoints(const Features259& features0, const Features259& features1, std::vector<cv::DMatch>& matches,
                       bool outlier_rejection = false);
    int MatchingPoints(Frame& f1, Frame& f2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                       bool outlier_rejection = false);
    Features259 NormalizeKeypoints(const Features259& features, int width, int height);
    Features259 ConvertToEigenMatrix(const std::vector<cv::KeyPoint>& keypoints, const cv::Mat& descriptors);
    void ConvertMatchesToVector(const std::vector<cv::DMatch>& matches, std::vector<int>& vnMatches12);
    void plotspmatch(cv::Mat frame1, cv::Mat frame2, std::vector<cv::KeyPoint> kpts1, std::vector<cv::KeyPoint> kpts2,
                     std::vector<int> vmatches12);

    // ---- descriptor distances (rfe_l2_distance_matrix / rfe_search_candidates compute the first one in bulk)
    static float DescriptorDistance_sp(const cv::Mat& a, const cv::Mat& b);   // 256-d L2, SPmatcher.cc:2184-2189
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);        // ORB Hamming leftover

    // ---- classic CPU searches, bodies unchanged in SPmatcher.cc
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                           const float thFarPoints = 50.0f);
    int SearchByProjection1(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints,
                            const float thFarPoints);
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const int th);
    int SearchByProjection(Frame& CurrentFrame, Frame& LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                           const int ORBdist);
    int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                           std::vector<MapPoint*>& vpMatched, int th, float ratioHamming = 1.0);
    int SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints,
                           const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched, int th, float ratioHamming);
    int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                           const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched,
                           std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming = 1.0);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, std::vector<cv::DMatch>& vmatches);
    int SearchByBoWSP(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoWSP(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, std::vector<cv::DMatch>& vmatches);
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo, const bool bCoarse = false);
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false);
    int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, float th,
             std::vector<MapPoint*>& vpReplacePoint);

public:
    static const float TH_LOW;      // defined in SPmatcher.cc:13-15
    static const float TH_HIGH;
    static const int HISTO_LENGTH;
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

    LightGlueDecoupleOnnxRunner* featureMatcher;

protected:
    float RadiusByViewingCos(const float& viewCos);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);
    float mfNNratio;
    bool mbCheckOrientation;
};

typedef std::shared_ptr<SPmatcher> SPmatcherPtr;

}  // namespace ORB_SLAM3

#else  // ---------------------------------------------------------------------------------------------- stand-alone
#include <cmath>
#include <memory>
#include <vector>
#include "lightglue_onnx.h"

namespace ORB_SLAM3 {

namespace rfe_detail {
// C++14 has no inline variables: static data members of a class TEMPLATE may be defined in a header, so the three
// constants live in a templated base (values: SPmatcher.cc:13-15)
template <class Tag>
struct SPmatcherConstants {
    static const float TH_LOW;
    static const float TH_HIGH;
    static const int HISTO_LENGTH;
};
template <class Tag> const float SPmatcherConstants<Tag>::TH_HIGH = 1.4f;
template <class Tag> const float SPmatcherConstants<Tag>::TH_LOW = 1.2f;
template <class Tag> const int SPmatcherConstants<Tag>::HISTO_LENGTH = 30;
}  // namespace rfe_detail

class SPmatcher : public rfe_detail::SPmatcherConstants<void> {
public:
    explicit SPmatcher(float thre = 0.0f) : featureMatcher(nullptr), rows_(300), cols_(400) {
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

    // SPmatcher.cc:359-371 (its vnMatches12 is a local there too: only the count comes back)
    int MatchingPoints_onnx(std::vector<cv::Point2f> kpts0, std::vector<cv::Point2f> kpts1, float* desc0, float* desc1) {
        std::vector<int> vn(kpts0.size(), -1);
        return run(kpts0, kpts1, desc0, desc1, rows_, cols_, vn);
    }
    // SPmatcher.cc:374-410
    int MatchingPoints_onnx(std::vector<cv::Point2f> kpts0, std::vector<cv::Point2f> kpts1, cv::Mat desc0, cv::Mat desc1,
                            std::vector<int>& vnMatches12) {
        vnMatches12.resize(kpts0.size(), -1);
        std::vector<float> d0 = pack(desc0), d1 = pack(desc1);
        return run(kpts0, kpts1, d0.data(), d1.data(), rows_, cols_, vnMatches12);
    }
    // SPmatcher.cc:412-454
    int MatchingPoints_onnx(std::vector<cv::KeyPoint> kpts0, const std::vector<cv::KeyPoint> kpts1, cv::Mat desc0, const cv::Mat desc1,
                            std::vector<int>& vnMatches12) {
        vnMatches12.resize(kpts0.size(), -1);
        std::vector<cv::Point2f> p0, p1;
        for (size_t i = 0; i < kpts0.size(); ++i) p0.push_back(kpts0[i].pt);
        for (size_t i = 0; i < kpts1.size(); ++i) p1.push_back(kpts1[i].pt);
        std::vector<float> d0 = pack(desc0), d1 = pack(desc1);
        return run(p0, p1, d0.data(), d1.data(), rows_, cols_, vnMatches12);
    }
    // SPmatcher.cc:457-542 ; FrameT needs mvKeys, mDescriptors, imgLeft (ORB_SLAM3::Frame has them)
    template <class FrameT>
    int MatchingPoints_onnx(FrameT& f1, FrameT& f2, std::vector<int>& vnMatches12) {
        vnMatches12.resize(f1.mvKeys.size(), -1);
        std::vector<cv::Point2f> p0, p1;
        for (size_t i = 0; i < f1.mvKeys.size(); ++i) p0.push_back(f1.mvKeys[i].pt);
        for (size_t i = 0; i < f2.mvKeys.size(); ++i) p1.push_back(f2.mvKeys[i].pt);
        std::vector<float> d0 = pack(f1.mDescriptors), d1 = pack(f2.mDescriptors);
        return run(p0, p1, d0.data(), d1.data(), f2.imgLeft.rows, f2.imgLeft.cols, vnMatches12);
    }

    LightGlueDecoupleOnnxRunner* featureMatcher;

private:
    int rows_, cols_;   // 300 x 400: "needs changing" says the reference; see header comment
    static std::vector<float> pack(const cv::Mat& d) {
        std::vector<float> v((size_t)d.rows * d.cols);
        for (int r = 0; r < d.rows; ++r) { const float* s = d.ptr<float>(r); std::copy(s, s + d.cols, v.begin() + (size_t)r * d.cols); }
        return v;
    }
    int run(const std::vector<cv::Point2f>& k0, const std::vector<cv::Point2f>& k1, float* d0, float* d1, int rows, int cols,
            std::vector<int>& vn) {
        std::vector<cv::Point2f> n0 = featureMatcher->Matcher_PreProcess(k0, rows, cols);
        std::vector<cv::Point2f> n1 = featureMatcher->Matcher_PreProcess(k1, rows, cols);
        std::vector<rfe::Tensor> output = featureMatcher->Matcher_Inference(n0, n1, d0, d1);
        return featureMatcher->Matcher_PostProcess_fused(output, k0, k1, vn);
    }
};

typedef std::shared_ptr<SPmatcher> SPmatcherPtr;

}  // namespace ORB_SLAM3
#endif  // RFE_WITH_ROVER_SLAM
#endif  // ORBMATCHER_H
