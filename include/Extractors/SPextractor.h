// SPextractor.h -- drop-in for ORB_SLAM3::SPextractor (reference include/Extractors/SPextractor.h:55-136,
// src/Extractors/SPextractor.cc:84-146, :516-617): same constructor, operator(), getters and public
// members, so Frame / Tracking (src/Frame.cc:118-134,544-559, src/Tracking.cc:645-651) compile unchanged.
// The SuperPoint arithmetic runs on the MI355X through librover_fe.so.
#ifndef SPEXTRACTOR_H
#define SPEXTRACTOR_H
#include <cassert>
#include <cmath>
#include <string>
#include <vector>
#include "superpoint_onnx.h"

namespace ORB_SLAM3 {

class SPextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    SPextractor(int _nfeatures, float _scaleFactor, int _nlevels, float _iniThFAST, float _minThFAST)
        : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
        Configuration cfg;
        cfg.device = "cuda";                       // kept for symmetry with SPextractor.cc:92; ignored
        cfg.extractorPath = "onnxmodel/superpoint.onnx";   // src/Extractors/SPextractor.cc:93, unedited
        cfg.extractorType = "superpoint";
        featureExtractor = new SuperPointOnnxRunner();
        featureExtractor->InitOrtEnv(cfg);         // return value ignored, as in SPextractor.cc:96
        // scale tables exactly as SPextractor.cc:109-145
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mvImagePyramid.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        const float factor = 1.0f / (float)scaleFactor;
        float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = (int)std::lrint(nDesired);
            sum += mnFeaturesPerLevel[level];
            nDesired *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    }
    ~SPextractor() { delete featureExtractor; }
    SPextractor(const SPextractor&) = delete;
    SPextractor& operator=(const SPextractor&) = delete;

    // Compute the SuperPoint keypoints and descriptors of an 8-bit grayscale image; returns their number.
    int operator()(cv::InputArray _image, std::vector<cv::KeyPoint>& _keypoints, cv::Mat& _descriptors) {
        if (_image.empty()) return 0;
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);           // SPextractor.cc:525
        if (nlevels == 1) return ExtractSingleLayer(image, _keypoints, _descriptors);
        // nlevels > 1: the reference's ExtractMultiLayers never calls the model (SPextractor.cc:629,634 are
        // commented out) and returns an empty result; reproduced.
        _keypoints.clear();
        return 0;
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<cv::Mat> mvImagePyramid;
    SuperPointOnnxRunner* featureExtractor;
    std::string mModelstr = "onnx";
    float lastmatchnum = 0;                        // written by LocalMapping (src/LocalMapping.cc:951-952)

protected:
    int ExtractSingleLayer(const cv::Mat& image, std::vector<cv::KeyPoint>& vKeyPoints, cv::Mat& Descriptors) {
        Configuration cfg;
        featureExtractor->lastmatch = lastmatchnum;
        // NormalizeImage (transform.cpp:11) is fused into the first kernel: hand the u8 rows over directly.  Inference + post-processing in one
        // step (the library writes the descriptors into the cv::Mat this call returns; same results as Extractor_Inference_u8 followed by
        // Extractor_PostProcess, which stay available: tests/cpp/shim_driver.cpp drives both and compares)
        (void)cfg;
        featureExtractor->Extract_u8_direct(image.ptr<unsigned char>(0), image.rows, image.cols, (int)image.step, vKeyPoints, Descriptors);
        return (int)vKeyPoints.size();
    }

    int nfeatures;
    double scaleFactor;
    int nlevels;
    float iniThFAST;
    float minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM3
#endif
