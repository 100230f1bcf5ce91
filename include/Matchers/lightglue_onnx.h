// lightglue_onnx.h -- drop-in for the reference's LightGlueDecoupleOnnxRunner
// (include/Matchers/lightglue_onnx.h:10-67, src/Matchers/lightglue_onnx.cpp): same class / method names;
// the session is an rfe_ctx of librover_fe.so, tensors are rfe::Tensor instead of Ort::Value.
//  * InitOrtEnv: EXIT_SUCCESS / EXIT_FAILURE + std::cerr, like lightglue_onnx.cpp:4-98; the model path is
//    cfg.lightgluePath, else the reference's hard-coded onnxmodel/lightglue_sim.onnx (lightglue_onnx.cpp:38) -- read as an ONNX
//    graph by the library itself ($RFE_LG_WEIGHTS overrides the location; an RFEW container is accepted too).
//  * Matcher_Inference returns {matches0 i64 [S,2], mscores0 f32 [S]} or an EMPTY vector on failure
//    (lightglue_onnx.cpp:232-237).  Unlike the reference it does not leak its keypoint copies (:176-177).
//  * Matcher_PostProcess_fused: lightglue_onnx.cpp:396-482 (an empty `output` returns 0 instead of
//    the reference's unchecked index).
#pragma once
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>
#include <utility>
#include <vector>
#include "../rover_fe.h"
#include "../rfe/cv_compat.h"
#include "../rfe/tensor.h"
#include "Configuration.h"
#include "transform.h"

class LightGlueDecoupleOnnxRunner {
public:
    const unsigned int num_threads;
    rfe_ctx* MatcherSession = nullptr;             // reference: Ort::Session*
    std::vector<std::vector<int64_t>> MatcherInputNodeShapes = {{1, -1, 2}, {1, -1, 2}, {1, -1, 256}, {1, -1, 256}};
    float matchThresh = 0.0f;
    float filter_threshold = 0.1f;                 // in-graph filter of the fused LightGlue export: a constant of lightglue_sim.onnx, read from
                                                   // the weight file's RFEW v2 header at InitOrtEnv (rfe_get_hparams); 0.1 = published default
    long long extractor_timer = 0;
    long long matcher_timer = 0;
    std::vector<float> scales = {1.0f, 1.0f};
    std::vector<rfe::Tensor> matcher_outputtensors;
    std::pair<std::vector<cv::Point2f>, std::vector<cv::Point2f>> keypoints_result;

    explicit LightGlueDecoupleOnnxRunner(unsigned int threads = 1) : num_threads(threads) {}
    ~LightGlueDecoupleOnnxRunner() { if (MatcherSession) rfe_destroy(MatcherSession); }
    LightGlueDecoupleOnnxRunner(const LightGlueDecoupleOnnxRunner&) = delete;
    LightGlueDecoupleOnnxRunner& operator=(const LightGlueDecoupleOnnxRunner&) = delete;

    int InitOrtEnv(Configuration cfg) {
        std::string path = cfg.lightgluePath;
        // as given: the reference's hard-coded "onnxmodel/lightglue_sim.onnx" (src/Matchers/lightglue_onnx.cpp:38) is read by the library itself;
        // an RFEW container works as well; $RFE_LG_WEIGHTS only overrides where the file is
        if (const char* e = std::getenv("RFE_LG_WEIGHTS")) path = e;
        if (path.empty()) path = "onnxmodel/lightglue_sim.onnx";
        int dev = 0;
        if (const char* e = std::getenv("RFE_DEVICE")) dev = std::atoi(e);
        int rc = rfe_init(dev, &MatcherSession);
        if (rc != RFE_OK) {
            std::cerr << "[ERROR] rover_fe environment created failed : " << rfe_last_error(nullptr) << '\n';
            MatcherSession = nullptr;
            return EXIT_FAILURE;
        }
        rc = rfe_load_weights(MatcherSession, nullptr, path.c_str());
        if (rc != RFE_OK) {
            std::cerr << "[ERROR] rover_fe environment created failed : " << rfe_last_error(MatcherSession) << '\n';
            return EXIT_FAILURE;
        }
        rfe_hparams hp;
        if (rfe_get_hparams(MatcherSession, &hp) == RFE_OK) filter_threshold = hp.lg_filter_threshold;
        if (const char* e = std::getenv("RFE_HOST_GRAPH")) rfe_set_option(MatcherSession, RFE_OPT_HOST_GRAPH, std::atoi(e));   // deployment switch, see rover_fe.h
        return EXIT_SUCCESS;
    }

    // lightglue_onnx.cpp:140-159
    std::vector<cv::Point2f> Matcher_PreProcess(std::vector<cv::Point2f> kpts, int h, int w) { return NormalizeKeypoints(kpts, h, w); }
    std::vector<cv::Point2f> Matcher_PreProcess(std::vector<cv::KeyPoint> kpts, int h, int w) {
        std::vector<cv::Point2f> p;
        p.reserve(kpts.size());
        for (const cv::KeyPoint& k : kpts) p.push_back(k.pt);
        return NormalizeKeypoints(p, h, w);
    }

    // lightglue_onnx.cpp:162-240 ; kpts are already normalised
    std::vector<rfe::Tensor> Matcher_Inference(std::vector<cv::Point2f> kpts0, std::vector<cv::Point2f> kpts1, float* desc0, float* desc1) {
        std::vector<rfe::Tensor> out;
        if (!MatcherSession) { std::cerr << "[ERROR] Matcher inference failed : no session" << std::endl; return out; }
        const int M = (int)kpts0.size(), N = (int)kpts1.size();
        int32_t S = 0;
        const int cap = std::min(M, N);
        std::vector<int32_t> pairs((size_t)std::max(cap, 1) * 2);
        std::vector<float> ms(std::max(cap, 1));
        if (M > 0 && N > 0) {
            std::vector<float> k0((size_t)M * 2), k1((size_t)N * 2);
            for (int i = 0; i < M; ++i) { k0[2 * i] = kpts0[i].x; k0[2 * i + 1] = kpts0[i].y; }
            for (int i = 0; i < N; ++i) { k1[2 * i] = kpts1[i].x; k1[2 * i + 1] = kpts1[i].y; }
            int32_t m = M, n = N;
            auto t0 = std::chrono::high_resolution_clock::now();
            int rc = rfe_match(MatcherSession, k0.data(), k1.data(), desc0, desc1, &m, &n, 1, M, N, filter_threshold, &S, pairs.data(), ms.data());
            matcher_timer += std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
            if (rc != RFE_OK) {
                std::cerr << "[ERROR] LightGlueDecoupleOnnxRunner Matcher inference failed : " << rfe_last_error(MatcherSession) << std::endl;
                return out;   // empty, like lightglue_onnx.cpp:235-236
            }
        }
        out.emplace_back(std::vector<int64_t>{S, 2}, sizeof(int64_t));
        out.emplace_back(std::vector<int64_t>{S}, sizeof(float));
        int64_t* m64 = out[0].GetTensorMutableData<int64_t>();
        for (int i = 0; i < 2 * S; ++i) m64[i] = pairs[i];
        std::copy(ms.begin(), ms.begin() + S, out[1].GetTensorMutableData<float>());
        return out;
    }
    std::vector<rfe::Tensor> Matcher_Inference(std::vector<cv::KeyPoint> kpts0, std::vector<cv::KeyPoint> kpts1, float* desc0, float* desc1) {
        std::vector<cv::Point2f> p0, p1;
        for (const auto& k : kpts0) p0.push_back(k.pt);
        for (const auto& k : kpts1) p1.push_back(k.pt);
        return Matcher_Inference(p0, p1, desc0, desc1);
    }

    // lightglue_onnx.cpp:396-482
    int Matcher_PostProcess_fused(std::vector<rfe::Tensor>& output, std::vector<cv::Point2f>, std::vector<cv::Point2f>, std::vector<int>& vnMatches12) {
        int size = 0;
        if (output.size() < 2) { std::cerr << "[ERROR] PostProcess failed : empty inference output" << std::endl; return size; }
        const std::vector<int64_t> shape = output[0].GetTensorTypeAndShapeInfo().GetShape();
        const int64_t* matches = output[0].GetTensorMutableData<int64_t>();
        const float* mscores = output[1].GetTensorMutableData<float>();
        for (int64_t i = 0; i < shape[0]; i++)
            if (mscores[i] > this->matchThresh) { size++; vnMatches12[matches[i * 2]] = (int)matches[i * 2 + 1]; }
        return size;
    }

    float GetMatchThresh() { return matchThresh; }
    void SetMatchThresh(float thresh) { matchThresh = thresh; }
    double GetTimer(std::string name) { return name == "extractor" ? (double)extractor_timer : (double)matcher_timer; }
    std::pair<std::vector<cv::Point2f>, std::vector<cv::Point2f>> GetKeypointsResult() { return keypoints_result; }
};
