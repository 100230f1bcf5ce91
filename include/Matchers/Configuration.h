// Configuration.h -- option bag handed to the runner classes; member names and defaults follow the
// reference's include/Matchers/Configuration.h:6-20 so that call sites assign the same fields.
//
// Differences: the two paths name RFEW weight containers (rover-slam_amd/weights.py), not .onnx files --
// when they do not end in ".rfew" the shims fall back to $RFE_SP_WEIGHTS / $RFE_LG_WEIGHTS and then to
// onnxmodel/superpoint.rfew / onnxmodel/lightglue_sim.rfew; `device` is accepted and ignored (the only
// backend is HIP on gfx950; the reference passes "cuda", SPextractor.cc:92).  The keypoint budget, detection threshold, NMS radius,
// border and match filter are NOT configuration here, just as they are not in the reference: they are constants of the model files and
// travel in the RFEW v2 header (rfe_hparams, include/rover_fe.h).
#pragma once
#include <string>

struct Configuration {
    // model files
    std::string extractorPath, lightgluePath;
    std::string extractorType;                 // "superpoint"
    std::string device;                        // ignored
    // behaviour flags kept for source compatibility (unused by the reference's live path as well)
    bool isEndtoEnd = true, grayScale = false, viz = false;
    unsigned int image_size = 512;
    float threshold = 0.0f;
};
