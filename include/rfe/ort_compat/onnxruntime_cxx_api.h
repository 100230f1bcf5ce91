// onnxruntime_cxx_api.h (rfe/ort_compat) -- NOT ONNXRuntime.  Put this directory on the include path INSTEAD of
// onnxruntime-linux-x64-gpu-1.16.3/include (reference CMakeLists.txt:63) and the reference sources that only mention
// Ort::Value in passing keep compiling unchanged against the librover_fe.so runner classes:
//   src/Matchers/SPmatcher.cc:3,367,401,446,528   `#include <onnxruntime_cxx_api.h>`, `std::vector<Ort::Value> output = featureMatcher->Matcher_Inference(...)`
//   src/Extractors/SPextractor.cc:600              `std::move(featureExtractor->extractor_outputtensors[0])`
// The runners of this repo (include/Matchers/lightglue_onnx.h, include/Extractors/superpoint_onnx.h) hand out rfe::Tensor,
// which carries the slice of the Ort::Value surface those callers touch (GetTensorTypeAndShapeInfo().GetShape(),
// GetTensorMutableData<T>()).  Nothing else of the ONNXRuntime API exists here: code that creates sessions or tensors
// itself (the reference's own runner .cc files) is replaced, not recompiled.
#pragma once
#include "../tensor.h"
namespace Ort {
typedef rfe::Tensor Value;
}
