// tests/cpp/opencv_stub — NOT OpenCV.  Declarations of exactly the OpenCV 3.4.2 types include/imagestitch_cv.hpp touches (cv::Mat as a
// plain matrix header, the InputArray / OutputArray proxies, Point / Size / Rect, CV_Assert, the type codes), so that the adapter header
// can be compiled and exercised in an image that has no OpenCV.  It builds no reference code and implements no OpenCV algorithm; in the
// reference tree the real <opencv2/core.hpp> is found first and this directory is never on the include path.
#ifndef ISX_TEST_OPENCV_STUB_CORE_HPP
#define ISX_TEST_OPENCV_STUB_CORE_HPP
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_16S 3
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16SC3 CV_MAKETYPE(CV_16S, 3)
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)

namespace cv {
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<int> Point;
typedef Point_<float> Point2f;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect() : x(0), y(0), width(0), height(0) {} Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };

class Mat {
public:
    unsigned char* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
    Mat() {}
    Mat(int r, int c, int t) { create(r, c, t); }
    Mat(int r, int c, int t, void* d, size_t s) : data((unsigned char*)d), rows(r), cols(c), step(s), type_(t) {}
    int type() const { return type_; }
    bool empty() const { return data == nullptr; }
    Size size() const { return Size(cols, rows); }
    static size_t elemSize(int t) { static const int d[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return (size_t)d[t & 7] * ((t >> 3) + 1); }
    void create(int r, int c, int t) {
        if (data && r == rows && c == cols && t == type_) return;
        own_.reset(new unsigned char[(size_t)r * c * elemSize(t)], std::default_delete<unsigned char[]>());
        data = own_.get(); rows = r; cols = c; type_ = t; step = (size_t)c * elemSize(t);
    }
    template <class T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <class T> T* ptr(int y) { return (T*)(data + (size_t)y * step); }
    template <class T> const T* ptr(int y) const { return (const T*)(data + (size_t)y * step); }
private:
    int type_ = 0;
    std::shared_ptr<unsigned char> own_;
};

// cv::UMat as the reference holds its warped tiles (W:206-207: vector<UMat> masks_warped, images_warped; W:148: UMat xmap, ymap): an array
// that is NOT a Mat and is reached through getMat(access) - a Mat header over the UMat's storage, as cv::UMat::getMat maps it on a host
// (non-OpenCL) build.  maps() counts the mappings so that a test can tell the adapter went through one.
enum { ACCESS_READ = 1 << 24, ACCESS_WRITE = 1 << 25, ACCESS_RW = 3 << 24 };
class UMat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    UMat() {}
    int type() const { return type_; }
    bool empty() const { return !buf_; }
    Size size() const { return Size(cols, rows); }
    void create(int r, int c, int t) {
        if (buf_ && r == rows && c == cols && t == type_) return;
        buf_.reset(new unsigned char[(size_t)r * c * Mat::elemSize(t)], std::default_delete<unsigned char[]>());
        rows = r; cols = c; type_ = t; step = (size_t)c * Mat::elemSize(t);
    }
    Mat getMat(int /*access*/) const { ++maps_; return buf_ ? Mat(rows, cols, type_, buf_.get(), step) : Mat(); }
    int maps() const { return maps_; }
private:
    int type_ = 0;
    mutable int maps_ = 0;
    std::shared_ptr<unsigned char> buf_;
};

// cv::Ptr of 3.4.2 is a reference-counted pointer with an explicit constructor from a raw pointer and conversions between related types
template <class T> using Ptr = std::shared_ptr<T>;

// the array proxies: a reference to a Mat or to a UMat is all the adapter needs
class _InputArray {
public:
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)), u_(nullptr) {}
    _InputArray(const UMat& u) : m_(nullptr), u_(const_cast<UMat*>(&u)) {}
    Mat getMat() const { return m_ ? *m_ : u_->getMat(ACCESS_RW); }
    bool isUMat() const { return u_ != nullptr; }
protected:
    Mat* m_;
    UMat* u_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) : _InputArray(m) {}
    _OutputArray(UMat& u) : _InputArray(u) {}
    void create(int r, int c, int t) const { if (m_) m_->create(r, c, t); else u_->create(r, c, t); }
};
typedef _OutputArray _InputOutputArray;
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _InputOutputArray& InputOutputArray;
}  // namespace cv
#endif
