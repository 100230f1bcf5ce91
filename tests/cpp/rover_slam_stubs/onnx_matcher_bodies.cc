// In-tree mode check (see README.md): a translation unit shaped like the part of the reference's src/Matchers/SPmatcher.cc
// that touches the runner -- out-of-line definitions of the three constants, the constructor and the four
// MatchingPoints_onnx overloads written against Ort::Value -- compiled against include/Matchers/SPmatcher.h with
// -DRFE_WITH_ROVER_SLAM -std=c++14.  Written for this test; it exercises names and types, the arithmetic is the shim's.
#include <cmath>
#include <onnxruntime_cxx_api.h>
#include "Matchers/Configuration.h"
#include "Matchers/SPmatcher.h"
using namespace std;

namespace ORB_SLAM3 {
const float SPmatcher::TH_HIGH = 1.4;
const float SPmatcher::TH_LOW = 1.2;
const int SPmatcher::HISTO_LENGTH = 30;

SPmatcher::SPmatcher(float thre) {
    Configuration cfg;
    cfg.device = "cuda";
    cfg.extractorPath = "";
    cfg.extractorType = "";
    featureMatcher = new LightGlueDecoupleOnnxRunner();
    featureMatcher->InitOrtEnv(cfg);
    featureMatcher->SetMatchThresh(thre);
}

static float* flat(const cv::Mat& d) {
    float* p = new float[(size_t)d.rows * d.cols];
    for (int i = 0; i < d.rows; i++) for (int j = 0; j < d.cols; j++) p[i * d.cols + j] = d.ptr<float>(i)[j];
    return p;
}

int SPmatcher::MatchingPoints_onnx(vector<cv::Point2f> kpts0, vector<cv::Point2f> kpts1, float* desc0, float* desc1) {
    auto n0 = featureMatcher->Matcher_PreProcess(kpts0, 300, 400);
    auto n1 = featureMatcher->Matcher_PreProcess(kpts1, 300, 400);
    vector<Ort::Value> output = featureMatcher->Matcher_Inference(n0, n1, desc0, desc1);
    vector<int> vnMatches12(n0.size(), -1);
    return featureMatcher->Matcher_PostProcess_fused(output, kpts0, kpts1, vnMatches12);
}

int SPmatcher::MatchingPoints_onnx(vector<cv::Point2f> kpts0, vector<cv::Point2f> kpts1, cv::Mat desc0, cv::Mat desc1, vector<int>& vnMatches12) {
    vnMatches12.resize(kpts0.size(), -1);
    auto n0 = featureMatcher->Matcher_PreProcess(kpts0, 300, 400);
    auto n1 = featureMatcher->Matcher_PreProcess(kpts1, 300, 400);
    float *d0 = flat(desc0), *d1 = flat(desc1);
    vector<Ort::Value> output = featureMatcher->Matcher_Inference(n0, n1, d0, d1);
    delete[] d0; delete[] d1;
    return featureMatcher->Matcher_PostProcess_fused(output, kpts0, kpts1, vnMatches12);
}

int SPmatcher::MatchingPoints_onnx(vector<cv::KeyPoint> kpts0, const vector<cv::KeyPoint> kpts1, cv::Mat desc0, const cv::Mat desc1, vector<int>& vnMatches12) {
    vnMatches12.resize(kpts0.size(), -1);
    vector<cv::Point2f> p0, p1;
    for (const cv::KeyPoint& k : kpts0) p0.emplace_back(k.pt);
    for (const cv::KeyPoint& k : kpts1) p1.emplace_back(k.pt);
    auto n0 = featureMatcher->Matcher_PreProcess(kpts0, 300, 400);    // the KeyPoint overload of the runner
    auto n1 = featureMatcher->Matcher_PreProcess(kpts1, 300, 400);
    float *d0 = flat(desc0), *d1 = flat(desc1);
    vector<Ort::Value> output = featureMatcher->Matcher_Inference(n0, n1, d0, d1);
    delete[] d0; delete[] d1;
    return featureMatcher->Matcher_PostProcess_fused(output, p0, p1, vnMatches12);
}

int SPmatcher::MatchingPoints_onnx(Frame& f1, Frame& f2, vector<int>& vnMatches12) {
    vnMatches12.resize(f1.mvKeys.size(), -1);
    vector<cv::Point2f> p0, p1;
    for (const cv::KeyPoint& k : f1.mvKeys) p0.emplace_back(k.pt);
    for (const cv::KeyPoint& k : f2.mvKeys) p1.emplace_back(k.pt);
    auto n0 = featureMatcher->Matcher_PreProcess(p0, f2.imgLeft.rows, f2.imgLeft.cols);
    auto n1 = featureMatcher->Matcher_PreProcess(p1, f2.imgLeft.rows, f2.imgLeft.cols);
    float *d0 = flat(f1.mDescriptors), *d1 = flat(f2.mDescriptors);
    vector<Ort::Value> output = featureMatcher->Matcher_Inference(n0, n1, d0, d1);
    delete[] d0; delete[] d1;
    return featureMatcher->Matcher_PostProcess_fused(output, p0, p1, vnMatches12);
}

float SPmatcher::DescriptorDistance_sp(const cv::Mat& a, const cv::Mat& b) {
    double s = 0;
    for (int i = 0; i < a.cols; ++i) { const double d = (double)a.ptr<float>(0)[i] - b.ptr<float>(0)[i]; s += d * d; }
    return (float)sqrt(s);
}
}  // namespace ORB_SLAM3

// a caller in another translation unit style: constants, the shared_ptr typedef, a classic-search declaration
int rfe_in_tree_probe() {
    ORB_SLAM3::SPmatcherPtr m = std::make_shared<ORB_SLAM3::SPmatcher>(0.0f);
    int (ORB_SLAM3::SPmatcher::*fuse)(ORB_SLAM3::KeyFrame*, const std::vector<ORB_SLAM3::MapPoint*>&, const float, const bool) = &ORB_SLAM3::SPmatcher::Fuse;
    (void)fuse;
    return ORB_SLAM3::SPmatcher::HISTO_LENGTH + (ORB_SLAM3::SPmatcher::TH_HIGH > ORB_SLAM3::SPmatcher::TH_LOW);
}
