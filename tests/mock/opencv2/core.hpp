// MOCK of the few OpenCV 3.x declarations include/plf.hpp's PLF_WITH_OPENCV adapters touch -- test infrastructure only (this image has no OpenCV):
// lets tests/test_abi.py COMPILE the exact-signature adapters and lets tests/test_gpu_cpp_mirror.py drive them on the GPU box.
// Layouts that matter are real: cv::KeyPoint is the 28-byte POD, cv::Mat exposes data / rows / cols / step like the real header.
#pragma once
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_Assert(expr) assert(expr)
namespace cv {
struct Point2f { float x, y; };
class KeyPoint {
public:
    Point2f pt; float size, angle, response; int octave, class_id;
};
class Mat {
public:
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t stp = 0) : data((unsigned char *)ext), rows(r), cols(c), step(stp ? stp : (size_t)c * esz(type)), type_(type) {}
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        store_ = std::shared_ptr<unsigned char>(new unsigned char[step * (size_t)(r > 0 ? r : 1)], std::default_delete<unsigned char[]>());
        data = store_.get();
    }
    void release() { store_.reset(); data = nullptr; rows = cols = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    template <class T> T &at(int r, int c) { return *(T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <class T> const T &at(int r, int c) const { return *(const T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <class T> const T &at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
    unsigned char *data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_ = 0;
    std::shared_ptr<unsigned char> store_;
};
// the proxy classes of the real API, reduced to "wraps a cv::Mat"
class _InputArray {
public:
    _InputArray(const Mat &m) : m_(&m) {}
    bool empty() const { return m_->empty(); }
    Mat getMat() const { return *m_; }
private:
    const Mat *m_;
};
class _OutputArray {
public:
    _OutputArray(Mat &m) : m_(&m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
    Mat &getMat() const { return *m_; }
private:
    Mat *m_;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
}  // namespace cv
