// imagestitch.hpp — header-only C++ host-side mirror of the reference's call surface over the C-ABI.
//
// The reference is C++ on OpenCV (`cv::detail::RotationWarper`, `cv::detail::Blender`); this header gives
// the same class / method names, argument meaning and error behaviour (exceptions) on top of
// include/imagestitch_hip.h, so the pipeline code of W:217-233 / W:271-313 ports line by line:
//
//     isx::CylindricalWarper creator;                                   // W:219
//     auto warper = creator.create(focal);                              // W:222
//     isx::Point corner = warper->warp(img, K, R, isx::INTER_LINEAR, isx::BORDER_REFLECT, warped);   // W:229
//     warper->warp(mask, K, R, isx::INTER_NEAREST, isx::BORDER_CONSTANT, mask_warped);               // W:232
//     auto blender = isx::Blender::createDefault(isx::Blender::MULTI_BAND, false);                   // W:271
//     static_cast<isx::MultiBandBlender*>(blender.get())->setNumBands(4);                            // W:273
//     blender->prepare(corners, sizes);  blender->feed(img_s, mask, corner);  blender->blend(result, result_mask);
//
// With OpenCV available, define ISX_HAVE_OPENCV before including: isx::Mat then converts from / to cv::Mat
// without copying (same data / rows / cols / type() / step).
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <cstdio>
#include <vector>

#include "imagestitch_hip.h"
#ifdef ISX_HAVE_OPENCV
#include <opencv2/core.hpp>
#endif

namespace isx {

enum { INTER_NEAREST = ISX_INTER_NEAREST, INTER_LINEAR = ISX_INTER_LINEAR };
enum { BORDER_CONSTANT = ISX_BORDER_CONSTANT, BORDER_REPLICATE = ISX_BORDER_REPLICATE, BORDER_REFLECT = ISX_BORDER_REFLECT,
       BORDER_REFLECT_101 = ISX_BORDER_REFLECT_101 };

struct Point { int x = 0, y = 0; Point() = default; Point(int x_, int y_) : x(x_), y(y_) {} };
struct Size { int width = 0, height = 0; Size() = default; Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x = 0, y = 0, width = 0, height = 0; };

// cv::Exception analogue: thrown for every non-zero status of the C-ABI
class Exception : public std::runtime_error {
public:
    Exception(int code_, const std::string& msg) : std::runtime_error(msg), code(code_) {}
    int code;
};
inline void check(int rc) { if (rc != ISX_OK) throw Exception(rc, isx_last_error()); }

// cv::Mat-shaped matrix: owns a host buffer (create) or aliases foreign memory (host or HIP device).
class Mat {
public:
    Mat() { std::memset(&m_, 0, sizeof(m_)); m_.device = -1; }
    Mat(int rows, int cols, int type) : Mat() { create(rows, cols, type); }
    Mat(int rows, int cols, int type, void* data, size_t step, int device = -1) : Mat() {
        m_.data = data; m_.rows = rows; m_.cols = cols; m_.type = type; m_.step = step; m_.device = device;
    }
#ifdef ISX_HAVE_OPENCV
    Mat(const cv::Mat& c) : Mat(c.rows, c.cols, c.type(), c.data, c.step, -1) {}   // zero-copy view
    cv::Mat toCv() const { return cv::Mat(m_.rows, m_.cols, m_.type, m_.data, m_.step); }
#endif
    void create(int rows, int cols, int type) {   // OutputArray::create (W:128-129,150)
        // cv::Mat::create: a mat that already has this geometry keeps its buffer - whether it owns it or aliases the caller's
        // (host or device) memory
        if (m_.data != nullptr && m_.rows == rows && m_.cols == cols && m_.type == type) return;
        size_t es = elemSize(type);
        own_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)rows * cols * es], std::default_delete<unsigned char[]>());
        m_.data = own_.get(); m_.rows = rows; m_.cols = cols; m_.type = type; m_.step = (size_t)cols * es; m_.device = -1;
    }
    static size_t elemSize(int type) {
        static const int d[8] = {1, 1, 2, 2, 4, 4, 8, 2};
        return (size_t)d[type & 7] * ((type >> 3) + 1);
    }
    bool empty() const { return m_.data == nullptr; }
    int rows() const { return m_.rows; }
    int cols() const { return m_.cols; }
    int type() const { return m_.type; }
    Size size() const { return Size(m_.cols, m_.rows); }
    template <class T> T* ptr(int y) { return (T*)((unsigned char*)m_.data + (size_t)y * m_.step); }
    template <class T> const T* ptr(int y) const { return (const T*)((const unsigned char*)m_.data + (size_t)y * m_.step); }
    void setTo(unsigned char v) { for (int y = 0; y < m_.rows; ++y) std::memset(ptr<unsigned char>(y), v, (size_t)m_.cols * elemSize(m_.type)); }
    isx_mat* c() { return &m_; }
    const isx_mat* c() const { return &m_; }
private:
    isx_mat m_;
    std::shared_ptr<unsigned char> own_;
};

// cv::detail::RotationWarper (W:122-161; stock call sites B:105,109)
class RotationWarper {
public:
    RotationWarper(int kind, float scale, int device = 0) { check(isx_warper_create(kind, scale, device, &h_)); }
    virtual ~RotationWarper() { isx_warper_destroy(h_); }
    RotationWarper(const RotationWarper&) = delete;
    RotationWarper& operator=(const RotationWarper&) = delete;
    void setStream(void* hip_stream) { check(isx_warper_set_stream(h_, hip_stream)); }
    // Rect buildMaps(Size src_size, K, R, xmap, ymap)  W:122
    Rect buildMaps(Size src_size, const float K[9], const float R[9], Mat& xmap, Mat& ymap) {
        int roi[4];
        check(isx_warper_roi(h_, src_size.width, src_size.height, K, R, roi, nullptr));
        xmap.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, ISX_32FC1);   // W:128
        ymap.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, ISX_32FC1);   // W:129
        check(isx_warper_build_maps_roi(h_, K, R, roi, xmap.c(), ymap.c()));   // the fill W:133-141 over the ROI just computed: one scan per call
        Rect r; r.x = roi[0]; r.y = roi[1]; r.width = roi[2] - roi[0]; r.height = roi[3] - roi[1];   // Rect(tl, br)  W:143
        return r;
    }
    // Point warp(src, K, R, interp_mode, border_mode, dst)  W:145
    Point warp(const Mat& src, const float K[9], const float R[9], int interp_mode, int border_mode, Mat& dst) {
        int roi[4];
        check(isx_warper_roi(h_, src.cols(), src.rows(), K, R, roi, nullptr));   // buildMaps -> detectResultRoi, ONCE per call  W:126,149
        dst.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, src.type());   // dst.create(roi.height + 1, roi.width + 1)  W:150
        check(isx_warper_warp_roi(h_, src.c(), K, R, interp_mode, border_mode, roi, dst.c()));   // the remap with that ROI  W:157
        return Point(roi[0], roi[1]);   // dst_roi.tl()  W:160
    }
    Rect warpRoi(Size src_size, const float K[9], const float R[9]) {
        int roi[4];
        check(isx_warper_roi(h_, src_size.width, src_size.height, K, R, roi, nullptr));
        Rect r; r.x = roi[0]; r.y = roi[1]; r.width = roi[2] - roi[0] + 1; r.height = roi[3] - roi[1] + 1;
        return r;
    }
    // Not in the reference: the image and its all-255 mask in one pass (W:229 + W:232; dst_img CV_8UC3 or CV_16SC3 = + W:294), and the gain of
    // GainCompensator::apply (W:241-244) folded into that pass's store (isx_warper_set_gain; 1.0 = off)
    Point warpWithMask(const Mat& src, const float K[9], const float R[9], Mat& dst_img, Mat& dst_mask, int img_type = ISX_8UC3) {
        int roi[4];
        check(isx_warper_roi(h_, src.cols(), src.rows(), K, R, roi, nullptr));
        dst_img.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, img_type);
        dst_mask.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, ISX_8UC1);
        check(isx_warper_warp_with_mask_roi(h_, src.c(), nullptr, K, R, roi, dst_img.c(), dst_mask.c()));
        return Point(roi[0], roi[1]);
    }
    void setGain(double gain) { check(isx_warper_set_gain(h_, gain)); }
    // Not in the reference: the fused tile warps issued between the two calls (planned or _roi forms on device mats) leave as ONE launch
    // (isx_warper_begin_batch / isx_warper_end_batch); their outputs exist once endBatch() has returned and its launch has run
    void beginBatch() { check(isx_warper_begin_batch(h_)); }
    void endBatch() { check(isx_warper_end_batch(h_)); }
    isx_warper* handle() { return h_; }
private:
    isx_warper* h_ = nullptr;
};

struct WarperCreator { virtual ~WarperCreator() {} virtual std::shared_ptr<RotationWarper> create(float scale) const = 0; };
struct CylindricalWarper : WarperCreator {   // cv::CylindricalWarper  W:219
    std::shared_ptr<RotationWarper> create(float scale) const override { return std::make_shared<RotationWarper>(ISX_WARP_CYLINDRICAL, scale); }
};
struct SphericalWarper : WarperCreator {     // cv::SphericalWarper  B:93 (commented out in the reference)
    std::shared_ptr<RotationWarper> create(float scale) const override { return std::make_shared<RotationWarper>(ISX_WARP_SPHERICAL, scale); }
};

// cv::detail::Blender / MultiBandBlender (W:271-281,302,313)
class Blender {
public:
    enum { NO = ISX_BLEND_NO, FEATHER = ISX_BLEND_FEATHER, MULTI_BAND = ISX_BLEND_MULTI_BAND };
    virtual ~Blender() { isx_blender_destroy(h_); }
    Blender(const Blender&) = delete;              // owns the handle: a copy would destroy it twice
    Blender& operator=(const Blender&) = delete;
    static std::shared_ptr<Blender> createDefault(int type, bool try_gpu = false, int precision = ISX_PREC_I16);
    void prepare(const std::vector<Point>& corners, const std::vector<Size>& sizes) {   // W:281
        if (corners.size() != sizes.size()) throw Exception(ISX_ERR_INVALID, "prepare: corners.size() != sizes.size()");
        std::vector<int> c, s;
        for (size_t i = 0; i < corners.size(); ++i) { c.push_back(corners[i].x); c.push_back(corners[i].y); s.push_back(sizes[i].width); s.push_back(sizes[i].height); }
        check(isx_blender_prepare(h_, (int)corners.size(), c.data(), s.data()));
    }
    void prepare(Rect dst_roi) { check(isx_blender_prepare_roi(h_, dst_roi.x, dst_roi.y, dst_roi.width, dst_roi.height)); }
    void feed(const Mat& img, const Mat& mask, Point tl) { check(isx_blender_feed(h_, img.c(), mask.c(), tl.x, tl.y)); }   // W:302
    // Not in the reference: W:286-302 in one call - feed(img, dilate(seam_mask, MORPH_RECT kw x kh) & warped_mask, tl), the mask prepared
    // straight into the blender's own buffer (isx_blender_feed_dilated)
    void feedDilated(const Mat& img, const Mat& seam_mask, const Mat& warped_mask, int kw, int kh, Point tl) {
        check(isx_blender_feed_dilated(h_, img.c(), seam_mask.c(), warped_mask.c(), kw, kh, tl.x, tl.y));
    }
    void blend(Mat& dst, Mat& dst_mask) {   // W:313
        int w, h;
        check(isx_blender_result_size(h_, &w, &h));
        if (win_x1_ > win_x0_) w = win_x1_ - win_x0_;   // setWindow: the mats hold the window's columns only
        if (dst.empty() || dst.rows() != h || dst.cols() != w) dst.create(h, w, ISX_16SC3);
        dst_mask.create(h, w, ISX_8UC1);
        check(isx_blender_blend(h_, dst.c(), dst_mask.c()));
    }
    // Not in the reference: blend() of several prepared + fed blenders of one rig in ONE chain of launches (isx_blender_blend_batch);
    // dsts / dst_masks are allocated like blend()'s, every result is bit-identical to the blender's own blend().
    static void blendBatch(const std::vector<Blender*>& bs, std::vector<Mat>& dsts, std::vector<Mat>& dst_masks) {
        std::vector<isx_blender*> hs;
        std::vector<isx_mat> d, m;
        dsts.resize(bs.size()); dst_masks.resize(bs.size());
        for (size_t i = 0; i < bs.size(); ++i) {
            int w, h;
            check(isx_blender_result_size(bs[i]->h_, &w, &h));
            if (bs[i]->win_x1_ > bs[i]->win_x0_) w = bs[i]->win_x1_ - bs[i]->win_x0_;
            if (dsts[i].empty() || dsts[i].rows() != h || dsts[i].cols() != w) dsts[i].create(h, w, ISX_16SC3);
            dst_masks[i].create(h, w, ISX_8UC1);
            hs.push_back(bs[i]->h_); d.push_back(*dsts[i].c()); m.push_back(*dst_masks[i].c());
        }
        if (!bs.empty()) check(isx_blender_blend_batch(hs.data(), (int)bs.size(), d.data(), m.data()));
    }
    void setStream(void* hip_stream) { check(isx_blender_set_stream(h_, hip_stream)); }
    // Not in the reference (see imagestitch_hip.h): the deferred cycle (1: fed device mats stay valid until blend(); 2: feed() copies
    // them, OpenCV's contract) and the column window of a panorama that is cut into strips across GPUs.
    void setDeferredLevel0(int mode) { check(isx_blender_set_deferred_level0(h_, mode)); }
    void setWindow(int x0, int x1) { check(isx_blender_set_window(h_, x0, x1)); win_x0_ = x0; win_x1_ = x1; }
    isx_blender* handle() { return h_; }
protected:
    Blender() = default;
    isx_blender* h_ = nullptr;
    int win_x0_ = 0, win_x1_ = 0;
};

class MultiBandBlender : public Blender {
public:
    MultiBandBlender(int try_gpu = false, int num_bands = 5, int precision = ISX_PREC_I16, int device = 0) {
        (void)try_gpu;   // the reference passes false everywhere (W:276,278); this library IS the accelerator path
        check(isx_blender_create(ISX_BLEND_MULTI_BAND, num_bands, precision, device, &h_));
    }
    int numBands() { int n; check(isx_blender_num_bands(h_, &n)); return n; }
    void setNumBands(int val) { check(isx_blender_set_num_bands(h_, val)); }   // W:273
};

// cv::detail::FeatherBlender — the blender every reference demo actually runs (W:278-280)
class FeatherBlender : public Blender {
public:
    FeatherBlender(float sharpness = 0.02f, int device = 0) {
        check(isx_blender_create(ISX_BLEND_FEATHER, 0, ISX_PREC_I16, device, &h_));
        setSharpness(sharpness);
    }
    void setSharpness(float val) { check(isx_blender_set_sharpness(h_, val)); }   // fb->setSharpness(0.1)  W:280
};

// cv::detail::Blender itself - what Blender::createDefault(Blender::NO, false) returns (W:276): feed() copies the tile under its mask,
// blend() zeroes what no mask covered
class NoBlender : public Blender {
public:
    explicit NoBlender(int device = 0) { check(isx_blender_create(ISX_BLEND_NO, 0, ISX_PREC_I16, device, &h_)); }
};

inline std::shared_ptr<Blender> Blender::createDefault(int type, bool try_gpu, int precision) {
    if (type == NO) return std::make_shared<NoBlender>();
    if (type == FEATHER) return std::make_shared<FeatherBlender>();
    if (type != MULTI_BAND) throw Exception(ISX_ERR_INVALID, "Blender::createDefault: NO, FEATHER or MULTI_BAND");
    return std::make_shared<MultiBandBlender>(try_gpu, 5, precision);
}

// src.convertTo(dst, type) with alpha = 1, beta = 0 between the CV_8U / CV_16S / CV_32F depths (W:261, W:294, W:315's input)
inline void convertTo(const Mat& src, Mat& dst, int type, int device = 0) {
    if (dst.empty() || dst.rows() != src.rows() || dst.cols() != src.cols() || dst.type() != type) dst.create(src.rows(), src.cols(), type);
    check(isx_convert_to(src.c(), dst.c(), device, nullptr));
}

// dilate(mask, getStructuringElement(MORPH_RECT, Size(kw, kh))) [& other]  (W:286-301)
inline void dilateAnd(const Mat& mask, int kw, int kh, const Mat* other, Mat& out, int device = 0) {
    out.create(mask.rows(), mask.cols(), ISX_8UC1);
    check(isx_mask_dilate_and(mask.c(), other ? other->c() : nullptr, kw, kh, out.c(), device, nullptr));
}

// cv::detail::DpSeamFinder as the reference restates it in-tree (S:60-1093): seam_finder->find(images_warped_f, corners, masks_seam)
class DpSeamFinder {
public:
    explicit DpSeamFinder(int device = 0) : device_(device) {}
    void find(const std::vector<Mat>& src, const std::vector<Point>& corners, std::vector<Mat>& masks) {   // S:87, called at S:1192
        if (src.size() != corners.size() || src.size() != masks.size()) throw Exception(ISX_ERR_INVALID, "find: src, corners and masks differ in length");
        std::vector<isx_mat> im(src.size()), mk(src.size());
        std::vector<int> c;
        for (size_t i = 0; i < src.size(); ++i) { im[i] = *src[i].c(); mk[i] = *masks[i].c(); c.push_back(corners[i].x); c.push_back(corners[i].y); }
        check(isx_dp_seam_find((int)src.size(), im.data(), c.data(), mk.data(), device_, nullptr));
    }
private:
    int device_;
};

// cv::imread(path) (W:166): the decoder follows the file's signature, as OpenCV's does - "BM" bitmaps and JFIF / Exif JPEGs;
// cv::imwrite(path, img) for .bmp and .jpg (W:155-156,315; "pano.jpg" S:1282)
inline Mat imread(const char* path) {
    unsigned char sig[2] = {0, 0};
    if (FILE* f = std::fopen(path, "rb")) { if (std::fread(sig, 1, 2, f) != 2) sig[0] = sig[1] = 0; std::fclose(f); }
    const bool jpeg = sig[0] == 0xFF && sig[1] == 0xD8;
    int rows = 0, cols = 0;
    check(jpeg ? isx_jpeg_size(path, &rows, &cols) : isx_bmp_size(path, &rows, &cols));
    Mat m;
    m.create(rows, cols, ISX_8UC3);
    check(jpeg ? isx_jpeg_read(path, m.c()) : isx_bmp_read(path, m.c()));
    return m;
}
// the format follows the extension, as in OpenCV: .jpg / .jpeg -> baseline JFIF at cv::IMWRITE_JPEG_QUALITY (default 95), else .bmp
inline void imwrite(const char* path, const Mat& img, int jpeg_quality = 95) {
    const char* dot = nullptr;
    for (const char* q = path; *q; ++q) if (*q == '.') dot = q;
    auto ieq = [](const char* a, const char* b) { for (; *a && *b; ++a, ++b) if ((*a | 32) != *b) return false; return *a == *b; };
    if (dot && (ieq(dot, ".jpg") || ieq(dot, ".jpeg") || ieq(dot, ".jpe"))) check(isx_jpeg_write(path, img.c(), jpeg_quality));
    else check(isx_bmp_write(path, img.c()));
}

// GainCompensator::apply: multiply(image, gain, image)  (W:241-244)
inline void gainApply(Mat& image, double gain, int device = 0) { check(isx_gain_apply(image.c(), gain, device, nullptr)); }

}  // namespace isx
