// shim_driver.cpp -- exercises the drop-in headers the way the reference's callers do
// (src/Frame.cc:544-559 extractor call, src/Matchers/SPmatcher.cc:457-542 Frame overload,
// src/testDbow.cpp:160-201 direct runner use) with a mock Frame, and dumps the results for the
// Python test to compare with the C-ABI path and the oracle.
// usage: shim_driver <frames.u8> H W <out.bin>     (weights via $RFE_SP_WEIGHTS / $RFE_LG_WEIGHTS)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "Extractors/SPextractor.h"
#include "Matchers/SPmatcher.h"
#include "SuperPoint.h"
#include "super_glue.h"
#include "rfe/stereo_match.h"

struct MockFrame {                      // the three members SPmatcher::MatchingPoints_onnx(Frame&,Frame&) reads
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors;
    cv::Mat imgLeft;
};
struct MockStereoFrame {                // members Frame::ComputeStereoMatches uses (src/Frame.cc:1159-1446)
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight, imgLeft, imgRight;
    float mb = 0.11f, mbf = 0.11f * 435.0f;
    std::vector<float> mvuRight, mvDepth;
};

// Eigen-shaped stand-ins (column-major 259 x K matrix, dynamic vectors): SuperGlue::infer is a template over them, so the
// reference's Eigen::Matrix<double,259,Dynamic> / VectorXi / VectorXd call (include/super_glue.h:27-32) compiles as it stands
struct Mat259 {
    int k; std::vector<double> v;
    explicit Mat259(int k_) : k(k_), v((size_t)259 * k_) {}
    int cols() const { return k; } int rows() const { return 259; }
    double& operator()(int r, int c) { return v[(size_t)c * 259 + r]; }
    double operator()(int r, int c) const { return v[(size_t)c * 259 + r]; }
};
template <class T> struct VecX {
    std::vector<T> v;
    void resize(int n) { v.resize(n); }
    T& operator()(int i) { return v[i]; }
};

static void put(FILE* f, const void* p, size_t n) { fwrite(p, 1, n, f); }

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage\n"); return 2; }
    const int H = atoi(argv[2]), W = atoi(argv[3]);
    std::vector<unsigned char> raw((size_t)2 * H * W);
    FILE* fi = fopen(argv[1], "rb");
    if (!fi || fread(raw.data(), 1, raw.size(), fi) != raw.size()) { fprintf(stderr, "cannot read frames\n"); return 2; }
    fclose(fi);

    ORB_SLAM3::SPextractor ext(1000, 1.2f, 1, 20, 7);      // as in src/testDbow.cpp:36
    if (ext.GetLevels() != 1 || ext.GetScaleFactors().size() != 1 || ext.GetInverseScaleSigmaSquares()[0] != 1.0f) return 3;
    MockFrame f[2];
    for (int i = 0; i < 2; ++i) {
        f[i].imgLeft = cv::Mat(H, W, CV_8UC1, raw.data() + (size_t)i * H * W);
        const int n = ext(f[i].imgLeft, f[i].mvKeys, f[i].mDescriptors);
        if (n != (int)f[i].mvKeys.size() || f[i].mDescriptors.rows != n || f[i].mDescriptors.cols != 256) return 4;
    }
    ORB_SLAM3::SPmatcher matcher(0.0f);                    // Tracking: mspmatcher(0.0), src/Tracking.cc:70
    std::vector<int> vnFrame, vnQuirk;
    const int sFrame = matcher.MatchingPoints_onnx(f[0], f[1], vnFrame);                                    // true image size
    const int sQuirk = matcher.MatchingPoints_onnx(f[0].mvKeys, f[1].mvKeys, f[0].mDescriptors, f[1].mDescriptors, vnQuirk);  // 300x400
    // the two Point2f overloads (SPmatcher.cc:359-371 raw float*, :374-410 cv::Mat), also 300x400
    std::vector<cv::Point2f> p0, p1;
    for (const auto& k : f[0].mvKeys) p0.push_back(k.pt);
    for (const auto& k : f[1].mvKeys) p1.push_back(k.pt);
    std::vector<int> vnP2fMat;
    const int sP2fMat = matcher.MatchingPoints_onnx(p0, p1, f[0].mDescriptors, f[1].mDescriptors, vnP2fMat);
    std::vector<float> flat0((size_t)f[0].mDescriptors.rows * 256), flat1((size_t)f[1].mDescriptors.rows * 256);
    for (int r = 0; r < f[0].mDescriptors.rows; ++r) std::copy(f[0].mDescriptors.ptr<float>(r), f[0].mDescriptors.ptr<float>(r) + 256, flat0.begin() + (size_t)r * 256);
    for (int r = 0; r < f[1].mDescriptors.rows; ++r) std::copy(f[1].mDescriptors.ptr<float>(r), f[1].mDescriptors.ptr<float>(r) + 256, flat1.begin() + (size_t)r * 256);
    const int sP2fPtr = matcher.MatchingPoints_onnx(p0, p1, flat0.data(), flat1.data());
    if (ORB_SLAM3::SPmatcher::TH_HIGH != 1.4f || ORB_SLAM3::SPmatcher::TH_LOW != 1.2f || ORB_SLAM3::SPmatcher::HISTO_LENGTH != 30) return 6;

    // the SuperGlue veneer with Eigen-shaped arguments: keypoints normalised by the caller with the true image size -> same
    // assignment as the Frame overload; SPdetect() -> the same keypoints / descriptors as the extractor
    {
        SuperGlue sg;
        if (!sg.build()) return 8;
        Mat259 F0((int)p0.size()), F1((int)p1.size());
        const std::vector<cv::Point2f> n0 = NormalizeKeypoints(p0, H, W), n1 = NormalizeKeypoints(p1, H, W);
        for (int c = 0; c < F0.cols(); ++c) { F0(0, c) = f[0].mvKeys[c].response; F0(1, c) = n0[c].x; F0(2, c) = n0[c].y; for (int r = 0; r < 256; ++r) F0(3 + r, c) = flat0[(size_t)c * 256 + r]; }
        for (int c = 0; c < F1.cols(); ++c) { F1(0, c) = f[1].mvKeys[c].response; F1(1, c) = n1[c].x; F1(2, c) = n1[c].y; for (int r = 0; r < 256; ++r) F1(3 + r, c) = flat1[(size_t)c * 256 + r]; }
        VecX<int> i0, i1; VecX<double> s0, s1;
        if (!sg.infer(F0, F1, i0, i1, s0, s1)) return 8;
        for (size_t i = 0; i < vnFrame.size(); ++i) {
            if (i0(i) != vnFrame[i]) return 8;
            if (i0(i) >= 0 && (i1(i0(i)) != (int)i || s0(i) != s1(i0(i)) || !(s0(i) > 0.1))) return 8;
        }
        std::shared_ptr<ORB_SLAM3::SuperPoint> model = std::make_shared<ORB_SLAM3::SuperPoint>();
        Configuration cfg;
        if (model->InitOrtEnv(cfg) != EXIT_SUCCESS) return 9;
        std::vector<cv::KeyPoint> kd;
        cv::Mat dd = ORB_SLAM3::SPdetect(model, f[0].imgLeft, kd, 0.0, true, true);
        if (kd.size() != f[0].mvKeys.size() || dd.rows != (int)kd.size()) return 9;
        for (size_t i = 0; i < kd.size(); ++i)
            if (kd[i].pt.x != f[0].mvKeys[i].pt.x || kd[i].pt.y != f[0].mvKeys[i].pt.y || dd.ptr<float>((int)i)[7] != f[0].mDescriptors.ptr<float>((int)i)[7]) return 9;
    }

    // transform.cpp:3-17 / :85-90 on a colour image: NormalizeImage swaps BGR -> RGB and scales, RGB2Grayscale weights the channels
    {
        unsigned char bgr[2 * 3] = {10, 20, 30, 255, 0, 128};
        cv::Mat col(1, 2, CV_8UC3, bgr);
        cv::Mat nrm = NormalizeImage(col);
        if (nrm.channels() != 3 || nrm.ptr<float>(0)[0] != 30.0f * (float)(1.0 / 255.0) || nrm.ptr<float>(0)[2] != 10.0f * (float)(1.0 / 255.0)) return 7;
        cv::Mat gray = RGB2Grayscale(col);
        if (gray.channels() != 1 || gray.ptr<unsigned char>(0)[0] != (unsigned char)((10 * 4899 + 20 * 9617 + 30 * 1868 + 8192) >> 14)) return 7;
        cv::Mat g1(H, W, CV_8UC1, raw.data());
        cv::Mat n1 = NormalizeImage(g1);
        if (n1.channels() != 1 || n1.ptr<float>(0)[0] != (float)raw[0] * (float)(1.0 / 255.0)) return 7;
    }

    // the runner's float entry exactly as the reference drives it (superpoint_onnx.cc:88-162 on NormalizeImage's CV_32F output, then
    // Extractor_PostProcess): the same keypoints, responses and descriptors as operator() on the u8 frame, bit for bit
    {
        Configuration cfg;
        cv::Mat g0(H, W, CV_8UC1, raw.data());
        cv::Mat nf = NormalizeImage(g0);
        SuperPointOnnxRunner* run = ext.featureExtractor;
        if (run->Extractor_Inference(cfg, nf) != EXIT_SUCCESS || run->extractor_outputtensors.empty()) return 10;
        std::vector<cv::KeyPoint> kf;
        cv::Mat df;
        run->Extractor_PostProcess(cfg, std::move(run->extractor_outputtensors[0]), kf, df);
        if (kf.size() != f[0].mvKeys.size() || df.rows != (int)kf.size()) return 10;
        for (size_t i = 0; i < kf.size(); ++i) {
            if (kf[i].pt.x != f[0].mvKeys[i].pt.x || kf[i].pt.y != f[0].mvKeys[i].pt.y || kf[i].response != f[0].mvKeys[i].response) return 10;
            for (int c = 0; c < 256; c += 17)
                if (df.ptr<float>((int)i)[c] != f[0].mDescriptors.ptr<float>((int)i)[c]) return 10;
        }
    }

    // the two frames again as the left / right views of one stereo frame
    MockStereoFrame sf;
    sf.mvKeys = f[0].mvKeys; sf.mvKeysRight = f[1].mvKeys; sf.mDescriptors = f[0].mDescriptors; sf.mDescriptorsRight = f[1].mDescriptors;
    sf.imgLeft = f[0].imgLeft; sf.imgRight = f[1].imgLeft;
    if (ORB_SLAM3::ComputeStereoMatches_rfe(ext.featureExtractor->ExtractorSession, sf) != 0) return 5;

    FILE* fo = fopen(argv[4], "wb");
    for (int i = 0; i < 2; ++i) {
        const int32_t n = (int32_t)f[i].mvKeys.size();
        put(fo, &n, 4);
        for (const auto& k : f[i].mvKeys) { float v[5] = {k.pt.x, k.pt.y, k.response, k.size, (float)k.octave}; put(fo, v, sizeof(v)); }
        for (int r = 0; r < n; ++r) put(fo, f[i].mDescriptors.ptr<float>(r), 256 * 4);
    }
    const int32_t s[4] = {sFrame, sQuirk, sP2fMat, sP2fPtr}, m = (int32_t)vnFrame.size();
    put(fo, s, 16); put(fo, &m, 4);
    put(fo, vnFrame.data(), (size_t)m * 4); put(fo, vnQuirk.data(), (size_t)m * 4); put(fo, vnP2fMat.data(), (size_t)m * 4);
    put(fo, sf.mvuRight.data(), sf.mvuRight.size() * 4); put(fo, sf.mvDepth.data(), sf.mvDepth.size() * 4);
    fclose(fo);
    printf("shim_driver: %d / %d keypoints, %d matches (frame overload), %d (300x400 overload)\n",
           (int)f[0].mvKeys.size(), (int)f[1].mvKeys.size(), sFrame, sQuirk);
    return 0;
}
