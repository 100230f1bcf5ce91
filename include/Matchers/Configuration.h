// Configuration.h -- same fields as the reference's include/Matchers/Configuration.h:6-20.
// Extension: lightgluePath / extractorPath name RFEW weight containers (rover-slam_amd/weights.py)
// instead of .onnx files; when empty the shims use $RFE_LG_WEIGHTS / $RFE_SP_WEIGHTS or the
// reference's directory with the new extension (onnxmodel/lightglue_sim.rfew, onnxmodel/superpoint.rfew).
#ifndef CONFIGURATION_H
#define CONFIGURATION_H
#include <string>
struct Configuration {
    std::string lightgluePath;
    std::string extractorPath;
    std::string extractorType;
    bool isEndtoEnd = true;
    bool grayScale = false;
    unsigned int image_size = 512;
    float threshold = 0.0f;
    std::string device;   // "cuda" in the reference (SPextractor.cc:92); ignored: the only backend is HIP/gfx950
    bool viz = false;
};
#endif
