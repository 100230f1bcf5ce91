// SuperPoint.h -- compatibility veneer for the name BASELINE.json cites.  The reference's
// include/SuperPoint.h:18-60 (struct SuperPoint : torch::nn::Module, class SPDetector) is dead code:
// it needs libtorch, is not in CMakeLists.txt:75-148 and nothing includes it.  This header offers the
// same entry points on top of librover_fe.so: SPdetect(), and SPDetector::detect() then getKeyPoints() / computeDescriptors()
// (include/SuperPoint.h:45-57).  `SuperPoint` names the runner class, so `std::shared_ptr<SuperPoint> model` keeps compiling;
// the torch::Tensor overload of detect() takes the 8-bit grayscale cv::Mat instead.
#pragma once
#include <memory>
#include "Extractors/SPextractor.h"

namespace ORB_SLAM3 {

typedef SuperPointOnnxRunner SuperPoint;   // reference: struct SuperPoint : torch::nn::Module (include/SuperPoint.h:18-43)

class SPDetector {
public:
    explicit SPDetector(std::shared_ptr<SuperPointOnnxRunner> model) : model_(std::move(model)) {}
    // runs the network on an 8-bit grayscale image (the reference also took a `cuda` flag; ignored)
    void detect(cv::Mat& image, bool /*cuda*/ = true) {
        kpts_.clear();
        Configuration cfg;
        if (model_->Extractor_Inference_u8(image.ptr<unsigned char>(0), image.rows, image.cols, (int)image.step) == EXIT_SUCCESS)
            model_->Extractor_PostProcess(cfg, std::move(model_->extractor_outputtensors[0]), kpts_, desc_);
    }
    // keypoints inside [iniX,maxX) x [iniY,maxY) with response >= threshold (NMS already applied in-graph)
    void getKeyPoints(float threshold, int iniX, int maxX, int iniY, int maxY, std::vector<cv::KeyPoint>& keypoints, bool /*nms*/ = true) {
        keypoints.clear(); sel_.clear();
        for (size_t i = 0; i < kpts_.size(); ++i) {
            const cv::KeyPoint& k = kpts_[i];
            if (k.response >= threshold && k.pt.x >= iniX && k.pt.x < maxX && k.pt.y >= iniY && k.pt.y < maxY) { keypoints.push_back(k); sel_.push_back((int)i); }
        }
    }
    void computeDescriptors(const std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors) {
        descriptors = cv::Mat((int)keypoints.size(), 256, CV_32F);
        for (size_t r = 0; r < keypoints.size() && r < sel_.size(); ++r)
            std::copy(desc_.ptr<float>(sel_[r]), desc_.ptr<float>(sel_[r]) + 256, descriptors.ptr<float>((int)r));
    }
private:
    std::shared_ptr<SuperPointOnnxRunner> model_;
    std::vector<cv::KeyPoint> kpts_;
    std::vector<int> sel_;
    cv::Mat desc_;
};

// include/SuperPoint.h:45: one call -- keypoints with response >= threshold, returns their descriptors [K,256] CV_32F
inline cv::Mat SPdetect(std::shared_ptr<SuperPoint> model, cv::Mat img, std::vector<cv::KeyPoint>& keypoints, double threshold,
                        bool nms = true, bool cuda = true) {
    SPDetector det(model);
    det.detect(img, cuda);
    det.getKeyPoints((float)threshold, 0, img.cols, 0, img.rows, keypoints, nms);
    cv::Mat desc;
    det.computeDescriptors(keypoints, desc);
    return desc;
}

}  // namespace ORB_SLAM3
