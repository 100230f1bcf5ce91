// rfe/tensor.h -- stand-in for the Ort::Value tensors that leak through the reference's runner
// interface (include/Extractors/superpoint_onnx.h:47, include/Matchers/lightglue_onnx.h:48-51):
// just enough of the Ort::Value surface (GetTensorTypeAndShapeInfo().GetShape(),
// GetTensorMutableData<T>(), IsTensor(), HasValue()) for SPextractor.cc / SPmatcher.cc style callers.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace rfe {
struct TensorShapeInfo {
    std::vector<int64_t> shape;
    std::vector<int64_t> GetShape() const { return shape; }
};
class Tensor {
public:
    Tensor() = default;
    Tensor(std::vector<int64_t> shape, size_t elem_size) : shape_(std::move(shape)) {
        size_t n = elem_size;
        for (auto d : shape_) n *= (size_t)d;
        buf_ = std::shared_ptr<unsigned char>(new unsigned char[n ? n : 1], std::default_delete<unsigned char[]>());
    }
    // a tensor over a caller-provided buffer (the runner's pooled pinned blocks); `buf` keeps it alive
    Tensor(std::vector<int64_t> shape, std::shared_ptr<unsigned char> buf) : shape_(std::move(shape)), buf_(std::move(buf)) {}
    TensorShapeInfo GetTensorTypeAndShapeInfo() const { return TensorShapeInfo{shape_}; }
    template <typename T> T* GetTensorMutableData() { return (T*)buf_.get(); }
    template <typename T> const T* GetTensorData() const { return (const T*)buf_.get(); }
    // keep the buffer, declare fewer elements (the drop-in runner lets the library write straight into a K-row tensor and then
    // declares the n <= K rows that were detected, instead of copying out of a staging vector)
    void ShrinkTo(std::vector<int64_t> shape) { shape_ = std::move(shape); }
    bool IsTensor() const { return true; }
    bool HasValue() const { return (bool)buf_; }
private:
    std::vector<int64_t> shape_;
    std::shared_ptr<unsigned char> buf_;
};
}  // namespace rfe
