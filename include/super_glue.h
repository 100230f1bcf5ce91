// super_glue.h -- compatibility veneer for the name BASELINE.json cites.  The reference's
// include/super_glue.h:20-73 + src/super_glue.cpp (TensorRT SuperGlue, AirVO lineage) is dead code: it
// includes Thirdparty/TensorRTBuffer and read_configs.h, neither of which exists in the tree, and is not
// compiled.  Same outward shape here -- constructor from a SuperGlueConfig, build(), infer() filling
// indices0 / indices1 / mscores0 / mscores1 as the CPU decode of src/super_glue.cpp:341-369 did -- with
// LightGlue on librover_fe.so inside.
//
// Source compatibility with include/super_glue.h:27-32 without an Eigen dependency: infer() is a template over
// the matrix / vector types, so a caller's
//     Eigen::Matrix<double, 259, Eigen::Dynamic> features0, features1;  Eigen::VectorXi i0, i1;  Eigen::VectorXd s0, s1;
//     superglue.infer(features0, features1, i0, i1, s0, s1);
// compiles as it stands (uses .cols(), (row, col), .resize(n), (i)).  Feature layout as in src/super_glue.cpp:201-246:
// row 0 = keypoint score (unused by LightGlue), rows 1-2 = x, y ALREADY normalised by the caller
// (SPmatcher::NormalizeKeypoints(features, width, height), include/Matchers/SPmatcher.h:64-66), rows 3..258 = descriptor.
#pragma once
#include <memory>
#include <vector>
#include "Matchers/lightglue_onnx.h"

class SuperGlue {
public:
    SuperGlue() {}
    template <class ConfigT>   // the reference's SuperGlueConfig (image_width / image_height are read by infer_xy only)
    explicit SuperGlue(const ConfigT& cfg) : width_(cfg.image_width), height_(cfg.image_height) {}

    bool build() {
        Configuration cfg;
        return runner_.InitOrtEnv(cfg) == EXIT_SUCCESS;
    }
    void save_engine() {}                          // TensorRT engine cache of the reference: nothing to cache here
    bool deserialize_engine() { return false; }    // false -> the reference's build() path is taken

    template <class FeatMat, class VecI, class VecD>
    bool infer(const FeatMat& features0, const FeatMat& features1, VecI& indices0, VecI& indices1, VecD& mscores0, VecD& mscores1) {
        const int M = (int)features0.cols(), N = (int)features1.cols();
        std::vector<cv::Point2f> k0(M), k1(N);
        std::vector<float> d0((size_t)M * 256), d1((size_t)N * 256);
        for (int c = 0; c < M; ++c) {
            k0[c] = cv::Point2f((float)features0(1, c), (float)features0(2, c));
            for (int r = 0; r < 256; ++r) d0[(size_t)c * 256 + r] = (float)features0(3 + r, c);
        }
        for (int c = 0; c < N; ++c) {
            k1[c] = cv::Point2f((float)features1(1, c), (float)features1(2, c));
            for (int r = 0; r < 256; ++r) d1[(size_t)c * 256 + r] = (float)features1(3 + r, c);
        }
        std::vector<int> i0, i1;
        std::vector<double> s0, s1;
        if (!decode(runner_.Matcher_Inference(k0, k1, d0.data(), d1.data()), M, N, i0, i1, s0, s1)) return false;
        indices0.resize(M); mscores0.resize(M); indices1.resize(N); mscores1.resize(N);
        for (int i = 0; i < M; ++i) { indices0(i) = i0[i]; mscores0(i) = s0[i]; }
        for (int j = 0; j < N; ++j) { indices1(j) = i1[j]; mscores1(j) = s1[j]; }
        return true;
    }

    // plain-array form: PIXEL keypoints [K,2] (normalised here with the given image size), descriptors [K,256]
    bool infer_xy(const std::vector<cv::Point2f>& kpts0, const std::vector<cv::Point2f>& kpts1, float* desc0, float* desc1,
                  int rows, int cols, std::vector<int>& indices0, std::vector<int>& indices1,
                  std::vector<double>& mscores0, std::vector<double>& mscores1) {
        if (rows <= 0) rows = height_;
        if (cols <= 0) cols = width_;
        return decode(runner_.Matcher_Inference(runner_.Matcher_PreProcess(kpts0, rows, cols), runner_.Matcher_PreProcess(kpts1, rows, cols), desc0, desc1),
                      (int)kpts0.size(), (int)kpts1.size(), indices0, indices1, mscores0, mscores1);
    }

private:
    // matches0 [S,2] / mscores0 [S] -> per-keypoint indices (-1 = unmatched) and scores (0 = unmatched), both directions
    static bool decode(std::vector<rfe::Tensor> out, int M, int N, std::vector<int>& indices0, std::vector<int>& indices1,
                       std::vector<double>& mscores0, std::vector<double>& mscores1) {
        indices0.assign(M, -1); indices1.assign(N, -1);
        mscores0.assign(M, 0.0); mscores1.assign(N, 0.0);
        if (out.size() < 2) return false;
        const int64_t S = out[0].GetTensorTypeAndShapeInfo().GetShape()[0];
        const int64_t* m = out[0].GetTensorMutableData<int64_t>();
        const float* s = out[1].GetTensorMutableData<float>();
        for (int64_t i = 0; i < S; ++i) {
            indices0[m[2 * i]] = (int)m[2 * i + 1]; indices1[m[2 * i + 1]] = (int)m[2 * i];
            mscores0[m[2 * i]] = s[i]; mscores1[m[2 * i + 1]] = s[i];
        }
        return true;
    }
    int width_ = 400, height_ = 300;
    LightGlueDecoupleOnnxRunner runner_;
};

typedef std::shared_ptr<SuperGlue> SuperGluePtr;
