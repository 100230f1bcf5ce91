// shim_driver.cpp -- exercises the drop-in headers the way the reference's callers do
// (src/Frame.cc:544-559 extractor call, src/Matchers/SPmatcher.cc:457-542 Frame overload,
// src/testDbow.cpp:160-201 direct runner use) with a mock Frame, and dumps the results for the
// Python test to compare with the C-ABI path and the oracle.
// usage: shim_driver <frames.u8> H W <out.bin>     (weights via $RFE_SP_WEIGHTS / $RFE_LG_WEIGHTS)
#include <cstdio>
#include <cstdlib>
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
    const int32_t s[2] = {sFrame, sQuirk}, m = (int32_t)vnFrame.size();
    put(fo, s, 8); put(fo, &m, 4);
    put(fo, vnFrame.data(), (size_t)m * 4); put(fo, vnQuirk.data(), (size_t)m * 4);
    put(fo, sf.mvuRight.data(), sf.mvuRight.size() * 4); put(fo, sf.mvDepth.data(), sf.mvDepth.size() * 4);
    fclose(fo);
    printf("shim_driver: %d / %d keypoints, %d matches (frame overload), %d (300x400 overload)\n",
           (int)f[0].mvKeys.size(), (int)f[1].mvKeys.size(), sFrame, sQuirk);
    return 0;
}
