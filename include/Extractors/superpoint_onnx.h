// superpoint_onnx.h -- drop-in for the reference's SuperPointOnnxRunner
// (include/Extractors/superpoint_onnx.h:10-68, src/Extractors/superpoint_onnx.cc): same class name,
// public members and method names, but the "session" is an rfe_ctx of librover_fe.so (hand-written
// HIP kernels for gfx950) instead of an Ort::Session, and tensors are rfe::Tensor instead of Ort::Value.
//
// Behavioural notes w.r.t. the reference:
//  * InitOrtEnv returns EXIT_SUCCESS / EXIT_FAILURE and prints to std::cerr on failure, as
//    superpoint_onnx.cc:59-65 does.  cfg.device is ignored (the only backend is HIP); cfg.extractorPath
//    names the model file exactly as in the reference (an .onnx graph; an RFEW container is accepted too, see Matchers/Configuration.h).
//  * Extractor_Inference takes the NormalizeImage()d CV_32F image like superpoint_onnx.cc:88 and
//    stores {keypoints i64 [1,K,2], scores f32 [1,K], descriptors f32 [1,K,256]} in
//    extractor_outputtensors, K = number of detected keypoints (<= max_keypoints).
//  * Extractor_PostProcess fills response = scores[idx]; the reference indexes scores[2*idx]
//    (superpoint_onnx.cc:227, out of bounds for idx >= K/2) -- the intended value is used here.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>
#include <utility>
#include <mutex>
#include <vector>
#include "../rover_fe.h"
#include "../rfe/cv_compat.h"
#include "../rfe/tensor.h"
#include "../Matchers/Configuration.h"
#include "../Matchers/transform.h"

class SuperPointOnnxRunner {
public:
    const unsigned int num_threads;
    rfe_ctx* ExtractorSession = nullptr;            // reference: Ort::Session*
    std::vector<std::vector<int64_t>> ExtractorInputNodeShapes = {{1, 1, -1, -1}};
    float matchThresh = 0.0f;
    long long extractor_timer = 0;
    long long matcher_timer = 0;
    float lastmatch = 0;
    std::vector<float> scales = {1.0f, 1.0f};
    // The reference takes K from the shape of the graph's `keypoints` output (superpoint_onnx.cc:169-181): max_num_keypoints and the
    // detection threshold are constants of superpoint.onnx.  Here they come from the weight file's RFEW v2 header (written by
    // rover-slam_amd/onnx_weights.py from the graph it converts) through rfe_get_hparams, as do the NMS radius and the border inside
    // the library; a version-1 file gives the published defaults below.
    int max_keypoints = 1024;
    float detection_threshold = 0.0005f;
    std::vector<std::vector<rfe::Tensor>> extractor_outputtensors;
    std::pair<std::vector<cv::Point2f>, std::vector<cv::Point2f>> keypoints_result;

    explicit SuperPointOnnxRunner(unsigned int threads = 1) : num_threads(threads) {}
    ~SuperPointOnnxRunner() { if (ExtractorSession) rfe_destroy(ExtractorSession); }
    SuperPointOnnxRunner(const SuperPointOnnxRunner&) = delete;
    SuperPointOnnxRunner& operator=(const SuperPointOnnxRunner&) = delete;

    int InitOrtEnv(Configuration cfg) {
        std::string path = cfg.extractorPath;
        // the path is used AS GIVEN: the reference's own "onnxmodel/superpoint.onnx" (src/Extractors/SPextractor.cc:93) is read by the library
        // itself (initializers + graph hyper-parameters, rover-slam_amd/csrc/onnx_load.hip), an RFEW container works as well; $RFE_SP_WEIGHTS
        // only overrides where the file is
        if (const char* e = std::getenv("RFE_SP_WEIGHTS")) path = e;
        if (path.empty()) path = "onnxmodel/superpoint.onnx";
        int dev = 0;
        if (const char* e = std::getenv("RFE_DEVICE")) dev = std::atoi(e);
        int rc = rfe_init(dev, &ExtractorSession);
        if (rc != RFE_OK) {
            std::cerr << "[ERROR] rover_fe environment created failed : " << rfe_last_error(nullptr) << '\n';
            ExtractorSession = nullptr;
            return EXIT_FAILURE;
        }
        rc = rfe_load_weights(ExtractorSession, path.c_str(), nullptr);
        if (rc != RFE_OK) {
            std::cerr << "[ERROR] rover_fe environment created failed : " << rfe_last_error(ExtractorSession) << '\n';
            return EXIT_FAILURE;
        }
        rfe_hparams hp;
        if (rfe_get_hparams(ExtractorSession, &hp) == RFE_OK) { max_keypoints = hp.sp_max_keypoints; detection_threshold = hp.sp_detection_threshold; }
        if (const char* e = std::getenv("RFE_HOST_GRAPH")) rfe_set_option(ExtractorSession, RFE_OPT_HOST_GRAPH, std::atoi(e));   // deployment switch, see rover_fe.h
        return EXIT_SUCCESS;
    }

    // reference superpoint_onnx.cc:68-86 (unused there as well)
    cv::Mat Extractor_PreProcess(Configuration, const cv::Mat& srcImage, float&) {
        cv::Mat t = srcImage.clone();
        return NormalizeImage(t);
    }

    // u8 fast path used by SPextractor::ExtractSingleLayer (NormalizeImage is fused into the first kernel)
    int Extractor_Inference_u8(const unsigned char* img, int H, int W, int stride) { return run_(img, false, H, W, stride); }

    // Extractor_Inference_u8 + Extractor_PostProcess in one step for SPextractor::operator(): the library writes the K x 256 descriptors straight into
    // a FRESH cv::Mat (a new allocation per frame like the reference's `Descriptors = mat1`, superpoint_onnx.cc:231-244: headers that share the
    // previous frame's buffer stay valid), the detected n <= K rows are declared afterwards (rowRange: a header, no copy) -- one host copy of the
    // descriptors per frame (the library's staging block -> the Mat) instead of two (-> tensor -> Mat).  Same field values as
    // Extractor_PostProcess: pt = (x, y), response = scores[idx], size 10, octave 0; its threshold is the reference's constant 0 (:190), which no
    // score of the graph's output is below.
    int Extract_u8_direct(const unsigned char* img, int H, int W, int stride, std::vector<cv::KeyPoint>& vKeyPoints, cv::Mat& Descriptors) {
        extractor_outputtensors.clear();
        if (!ExtractorSession) { std::cerr << "[ERROR] Extractor inference failed : no session" << std::endl; return EXIT_FAILURE; }
        const int K = max_keypoints;
        if (stage_kxy_.size() < (size_t)K * 2) stage_kxy_.resize((size_t)K * 2);
        if (stage_score_.size() < (size_t)K) stage_score_.resize((size_t)K);
        cv::Mat d(K, 256, CV_32F);
        int32_t n = 0;
        auto t0 = std::chrono::high_resolution_clock::now();
        const int rc = rfe_extract_u8(ExtractorSession, img, H, W, stride, 1, K, detection_threshold, &n, stage_kxy_.data(), stage_score_.data(), d.ptr<float>(0));
        extractor_timer += std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        if (rc != RFE_OK) { std::cerr << "[ERROR] Extractor inference failed : " << rfe_last_error(ExtractorSession) << std::endl; return EXIT_FAILURE; }
        vKeyPoints.reserve(vKeyPoints.size() + (size_t)n);
        for (int i = 0; i < n; ++i) {
            cv::KeyPoint kp;
            kp.pt = cv::Point2f((float)stage_kxy_[2 * i], (float)stage_kxy_[2 * i + 1]);
            kp.response = stage_score_[i];
            kp.size = 10;
            kp.octave = 0;
            vKeyPoints.emplace_back(kp);
        }
        Descriptors = n == K ? d : d.rowRange(0, n);
        return EXIT_SUCCESS;
    }

    // reference superpoint_onnx.cc:88-162: image is a CV_32F single-channel image, normally NormalizeImage's output; like the graph,
    // the kernels take the values as they are (no assumption that they are multiples of 1/255 or inside [0, 1])
    int Extractor_Inference(Configuration, const cv::Mat& image) {
        if (image.step % sizeof(float) != 0) { std::cerr << "[ERROR] Extractor inference failed : row step is not a multiple of 4" << std::endl; return EXIT_FAILURE; }
        return run_(image.ptr<float>(0), true, image.rows, image.cols, (int)(image.step / sizeof(float)));
    }

    // reference superpoint_onnx.cc:165-255
    void Extractor_PostProcess(Configuration, std::vector<rfe::Tensor> tensor, std::vector<cv::KeyPoint>& vKeyPoints,
                               cv::Mat& Descriptors) {
        if (tensor.size() < 3) { std::cerr << "[ERROR] Extractor postprocess failed : missing tensors" << std::endl; return; }
        const std::vector<int64_t> kshape = tensor[0].GetTensorTypeAndShapeInfo().GetShape();
        const int64_t* kpts = tensor[0].GetTensorMutableData<int64_t>();
        const float* scores = tensor[1].GetTensorMutableData<float>();
        const std::vector<int64_t> dshape = tensor[2].GetTensorTypeAndShapeInfo().GetShape();
        const float* desc = tensor[2].GetTensorMutableData<float>();
        const float threshold = 0;  // adaptive branch disabled in the reference (superpoint_onnx.cc:190-193)
        int keep = 0;
        for (int64_t i = 0; i < kshape[1]; ++i) if (!(scores[i] < threshold)) ++keep;
        cv::Mat mat1(keep, (int)dshape[2], CV_32F);
        int row = 0;
        for (int64_t i = 0; i < kshape[1]; ++i) {
            if (scores[i] < threshold) continue;
            cv::KeyPoint kp;
            kp.pt = cv::Point2f((float)kpts[2 * i], (float)kpts[2 * i + 1]);
            kp.response = scores[i];
            kp.size = 10;
            kp.octave = 0;
            vKeyPoints.emplace_back(kp);
            std::copy(desc + i * dshape[2], desc + (i + 1) * dshape[2], mat1.ptr<float>(row));
            ++row;
        }
        Descriptors = mat1;
    }

    float GetMatchThresh() { return matchThresh; }
    void SetMatchThresh(float thresh) { matchThresh = thresh; }
    double GetTimer(std::string name) { return name == "extractor" ? (double)extractor_timer : (double)matcher_timer; }
    std::pair<std::vector<cv::Point2f>, std::vector<cv::Point2f>> GetKeypointsResult() { return keypoints_result; }

private:

    int run_(const void* img, bool f32, int H, int W, int stride) {
        extractor_outputtensors.clear();
        if (!ExtractorSession) { std::cerr << "[ERROR] Extractor inference failed : no session" << std::endl; return EXIT_FAILURE; }
        const int K = max_keypoints;
        std::vector<int32_t>& kxy = stage_kxy_;        // grow-only staging owned by the runner (int32 keypoints; the tensors below are int64 / float)
        if (kxy.size() < (size_t)K * 2) kxy.resize((size_t)K * 2);
        // the library writes scores and descriptors straight into K-row tensors; the n <= K detected rows are declared afterwards
        std::vector<rfe::Tensor> out;
        out.emplace_back(std::vector<int64_t>{1, K, 2}, sizeof(int64_t));
        out.emplace_back(std::vector<int64_t>{1, K}, sizeof(float));
        // the K x 256 descriptor tensor lives in a pooled PINNED block (rfe_host_malloc): rfe_extract_* DMAs the descriptors straight into it, and
        // the block returns to the pool when the caller drops the tensor -- no 1 MB allocation and no 1 MB host copy per frame
        out.emplace_back(std::vector<int64_t>{1, K, 256}, pool_->take((size_t)K * 256 * sizeof(float)));
        float* sc = out[1].GetTensorMutableData<float>();
        float* desc = out[2].GetTensorMutableData<float>();
        int32_t n = 0;
        auto t0 = std::chrono::high_resolution_clock::now();
        int rc = f32 ? rfe_extract_f32(ExtractorSession, (const float*)img, H, W, stride, 1, K, detection_threshold, &n, kxy.data(), sc, desc)
                     : rfe_extract_u8(ExtractorSession, (const unsigned char*)img, H, W, stride, 1, K, detection_threshold, &n, kxy.data(), sc, desc);
        extractor_timer += std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        if (rc != RFE_OK) {
            std::cerr << "[ERROR] Extractor inference failed : " << rfe_last_error(ExtractorSession) << std::endl;
            return EXIT_FAILURE;
        }
        int64_t* k64 = out[0].GetTensorMutableData<int64_t>();
        for (int i = 0; i < 2 * n; ++i) k64[i] = kxy[i];
        out[0].ShrinkTo({1, n, 2}); out[1].ShrinkTo({1, n}); out[2].ShrinkTo({1, n, 256});
        extractor_outputtensors.emplace_back(std::move(out));
        return EXIT_SUCCESS;
    }
    std::vector<int32_t> stage_kxy_;
    std::vector<float> stage_score_;

    // blocks of one size, handed out as shared_ptr whose deleter puts them back; the pool outlives the runner while tensors are in flight
    struct PinnedPool : std::enable_shared_from_this<PinnedPool> {
        std::mutex mu;
        std::vector<std::pair<unsigned char*, size_t>> free_;
        std::shared_ptr<unsigned char> take(size_t bytes) {
            unsigned char* p = nullptr;
            {
                std::lock_guard<std::mutex> lk(mu);
                for (size_t i = 0; i < free_.size(); ++i)
                    if (free_[i].second == bytes) { p = free_[i].first; free_.erase(free_.begin() + i); break; }
            }
            if (!p) {
                void* q = nullptr;
                if (rfe_host_malloc(bytes, &q) != RFE_OK)        // no pinned memory left: an ordinary block (the library then stages and copies)
                    return std::shared_ptr<unsigned char>(new unsigned char[bytes], std::default_delete<unsigned char[]>());
                p = (unsigned char*)q;
            }
            std::shared_ptr<PinnedPool> self = shared_from_this();
            return std::shared_ptr<unsigned char>(p, [self, bytes](unsigned char* b) {
                std::lock_guard<std::mutex> lk(self->mu);
                if (self->free_.size() < 8) self->free_.push_back({b, bytes}); else rfe_host_free(b);
            });
        }
        ~PinnedPool() { for (auto& b : free_) rfe_host_free(b.first); }
    };
    std::shared_ptr<PinnedPool> pool_ = std::make_shared<PinnedPool>();
};
