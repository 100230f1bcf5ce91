// Configuration.h -- option bag handed to the runner classes; member names and defaults follow the
// reference's include/Matchers/Configuration.h:6-20 so that call sites assign the same fields.
//
// Differences: none in the paths -- they name the reference's own .onnx graph files and are used as given (the library reads initializers and
// graph hyper-parameters itself, rover-slam_amd/csrc/onnx_load.hip; an RFEW container from rover-slam_amd/weights.py / onnx_weights.py is
// accepted too, the file's magic decides; $RFE_SP_WEIGHTS / $RFE_LG_WEIGHTS override the location); `device` is accepted and ignored (the only
// backend is HIP on gfx950; the reference passes "cuda", SPextractor.cc:92).  The keypoint budget, detection threshold, NMS radius,
// border and match filter are NOT configuration here, just as they are not in the reference: they are constants of the model files and
// are read from the graph (or travel in an RFEW v2 header): rfe_hparams, include/rover_fe.h.
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
