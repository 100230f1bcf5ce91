// super_glue.h -- compatibility veneer for the name BASELINE.json cites.  The reference's
// include/super_glue.h:20-73 + src/super_glue.cpp (TensorRT SuperGlue, AirVO lineage) is dead code: it
// includes Thirdparty/TensorRTBuffer and read_configs.h, neither of which exists in the tree, and is not
// compiled.  Same outward shape here -- build(), infer() returning indices0 / indices1 / mscores0 /
// mscores1 as the CPU decode of src/super_glue.cpp:341-369 did -- with LightGlue on librover_fe.so inside.
#pragma once
#include <vector>
#include "Matchers/lightglue_onnx.h"

class SuperGlue {
public:
    SuperGlue() = default;
    bool build() {
        Configuration cfg;
        return runner_.InitOrtEnv(cfg) == EXIT_SUCCESS;
    }
    // features: [3 + 256] x K column-major in the reference (score, x, y, descriptor); here the
    // caller passes plain arrays: pixel keypoints [K,2], descriptors [K,256], image size.
    bool infer(const std::vector<cv::Point2f>& kpts0, const std::vector<cv::Point2f>& kpts1, float* desc0, float* desc1,
               int rows, int cols, std::vector<int>& indices0, std::vector<int>& indices1,
               std::vector<double>& mscores0, std::vector<double>& mscores1) {
        indices0.assign(kpts0.size(), -1); indices1.assign(kpts1.size(), -1);
        mscores0.assign(kpts0.size(), 0.0); mscores1.assign(kpts1.size(), 0.0);
        auto out = runner_.Matcher_Inference(runner_.Matcher_PreProcess(kpts0, rows, cols), runner_.Matcher_PreProcess(kpts1, rows, cols), desc0, desc1);
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
private:
    LightGlueDecoupleOnnxRunner runner_;
};
