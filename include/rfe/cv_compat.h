// rfe/cv_compat.h -- the handful of OpenCV types the extractor/matcher interface of Rover-SLAM
// exposes (cv::Point2f, cv::KeyPoint, cv::Mat, cv::InputArray).  With OpenCV installed the real
// headers are used; without it (this build image has none) a minimal POD mirror with the same
// member names lets the shim headers and their tests compile.
#pragma once
#if defined(RFE_USE_OPENCV) || (!defined(RFE_NO_OPENCV) && __has_include(<opencv2/core.hpp>))
#include <opencv2/core.hpp>
#define RFE_HAVE_OPENCV 1
#else
#define RFE_HAVE_OPENCV 0
#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5
#define CV_8UC3 16      // depth + ((channels - 1) << 3), as in OpenCV
#define CV_32FC3 21
namespace cv {
struct Point2f {
    float x = 0, y = 0;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
    Point2f operator-(const Point2f& o) const { return {x - o.x, y - o.y}; }
    Point2f operator/(float s) const { return {x / s, y / s}; }
};
struct Size { int width = 0, height = 0; Size() = default; Size(int w, int h) : width(w), height(h) {} };
struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
// row-major, reference-counted, 1 or 3 interleaved channels, CV_8U or CV_32F
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;
    size_t step = 0;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), data((unsigned char*)ext), type_(type) {
        step = step_ ? step_ : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type; step = (size_t)c * elemSize();
        buf_ = std::shared_ptr<unsigned char>(new unsigned char[step * (size_t)(r > 0 ? r : 1)], std::default_delete<unsigned char[]>());
        data = buf_.get();
    }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    int depth() const { return type_ & 7; }
    size_t elemSize() const { return (size_t)(depth() == CV_32F ? 4 : 1) * channels(); }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    Size size() const { return Size(cols, rows); }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + step * (size_t)r); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + step * (size_t)r); }
    template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) memcpy(m.ptr<unsigned char>(r), ptr<unsigned char>(r), (size_t)cols * elemSize());
        return m;
    }
    void release() { buf_.reset(); data = nullptr; rows = cols = 0; }
    Mat rowRange(int r0, int r1) const { Mat m(*this); m.data = data + step * (size_t)r0; m.rows = r1 - r0; return m; }   // header over the same buffer, like cv::Mat
    Mat getMat() const { return *this; }   // so that Mat doubles as InputArray
private:
    int type_ = CV_8U;
    std::shared_ptr<unsigned char> buf_;
};
typedef const Mat& InputArray;
}  // namespace cv
#endif
