// imagestitch_cv.hpp — the adapter a maintainer of the reference adds: subclasses of OpenCV 3.4.2's own plug-in interfaces
// cv::detail::RotationWarper and cv::detail::Blender over the C-ABI library (include/imagestitch_hip.h), so that the demos'
// Ptr<RotationWarper> / Ptr<Blender> variables (W:217-222, W:269-280; S:1236-1251; B:99-110) take the MI355X path with one changed
// line each.  cv::Mat goes in, cv::Mat comes out; host mats are staged by the library (PCIe-inclusive), results
// are those of OpenCV's CPU code path.
//
// Compiled inside the reference tree, where OpenCV 3.4.2 is installed.  This image has no OpenCV: tests/cpp/cv_adapter_demo.cpp
// compiles this header against tests/cpp/opencv_stub (declarations of exactly the OpenCV types used below, nothing more) and runs
// it on the GPU box, so the header is a compiled, tested file rather than documentation.
#ifndef IMAGESTITCH_CV_HPP
#define IMAGESTITCH_CV_HPP

#define ISX_HAVE_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/stitching/detail/blenders.hpp>
#include <opencv2/stitching/detail/warpers.hpp>

#include "imagestitch.hpp"

namespace isx_cv {

inline void k9(cv::InputArray a, float o[9]) {            // K, R are 3x3 CV_32F (asserted at W:94-95)
    cv::Mat m = a.getMat();
    CV_Assert(m.rows == 3 && m.cols == 3 && m.type() == CV_32F);
    for (int i = 0; i < 9; ++i) o[i] = m.at<float>(i / 3, i % 3);
}
inline isx::Rect to_isx(const cv::Rect& r) { isx::Rect q; q.x = r.x; q.y = r.y; q.width = r.width; q.height = r.height; return q; }

// cv::detail::RotationWarper over isx_warper.  StockWarper = cv::detail::CylindricalWarper / SphericalWarper: the two members of the
// interface that are not on the hot path (warpPoint, warpBackward: never called by the reference) are forwarded to it.
template <int KIND, class StockWarper>
class HipRotationWarper : public cv::detail::RotationWarper {
public:
    // fixed_rig: a rig that warps frame after frame with the same K, R may remember detectResultRoi (isx_warper_set_roi_cache)
    explicit HipRotationWarper(float scale, bool fixed_rig = false, int device = 0) : w_(KIND, scale, device), stock_(scale), scale_(scale) {
        if (fixed_rig) isx::check(isx_warper_set_roi_cache(w_.handle(), 1));
    }
    // Point warp(src, K, R, interp_mode, border_mode, dst)  W:145-161: ONE detectResultRoi (W:126), dst.create(h + 1, w + 1) (W:150), remap (W:157)
    // dst may be a cv::Mat or - as the reference passes it, W:206-207, 229, 232 - a cv::UMat: create() makes the array, getMat() maps it as a Mat
    // header that lives to the end of this call (the unmap of an OpenCL-backed UMat happens in its destructor, before the caller sees dst).
    cv::Point warp(cv::InputArray src, cv::InputArray K, cv::InputArray R, int interp_mode, int border_mode, cv::OutputArray dst) override {
        float k[9], r[9]; k9(K, k); k9(R, r);
        cv::Mat s = src.getMat();
        int roi[4];
        isx::check(isx_warper_roi(w_.handle(), s.cols, s.rows, k, r, roi, nullptr));
        dst.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, s.type());
        cv::Mat d = dst.getMat();
        isx::Mat is(s), id(d);
        isx::check(isx_warper_warp_roi(w_.handle(), is.c(), k, r, interp_mode, border_mode, roi, id.c()));
        return cv::Point(roi[0], roi[1]);                     // dst_roi.tl()  W:160
    }
    // Rect buildMaps(src_size, K, R, xmap, ymap)  W:122-144
    cv::Rect buildMaps(cv::Size src_size, cv::InputArray K, cv::InputArray R, cv::OutputArray xmap, cv::OutputArray ymap) override {
        float k[9], r[9]; k9(K, k); k9(R, r);
        int roi[4];
        isx::check(isx_warper_roi(w_.handle(), src_size.width, src_size.height, k, r, roi, nullptr));
        xmap.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, CV_32F);   // W:128
        ymap.create(roi[3] - roi[1] + 1, roi[2] - roi[0] + 1, CV_32F);   // W:129
        cv::Mat mx = xmap.getMat(), my = ymap.getMat();
        isx::Mat ix(mx), iy(my);
        isx::check(isx_warper_build_maps_roi(w_.handle(), k, r, roi, ix.c(), iy.c()));
        return cv::Rect(roi[0], roi[1], roi[2] - roi[0], roi[3] - roi[1]);   // Rect(dst_tl, dst_br)  W:143
    }
    // Rect warpRoi(src_size, K, R): the stock class returns Rect(tl, Point(br.x + 1, br.y + 1))
    cv::Rect warpRoi(cv::Size src_size, cv::InputArray K, cv::InputArray R) override {
        float k[9], r[9]; k9(K, k); k9(R, r);
        int roi[4];
        isx::check(isx_warper_roi(w_.handle(), src_size.width, src_size.height, k, r, roi, nullptr));
        return cv::Rect(roi[0], roi[1], roi[2] - roi[0] + 1, roi[3] - roi[1] + 1);
    }
    // warpPoint / warpBackward are NOT on the path this library replaces: the reference never calls them (W:229, 232 and B:105, 109 call warp();
    // W:122 buildMaps) and the C-ABI has no entry for either.  They forward to the stock OpenCV warper this adapter is instantiated over (host
    // code, OpenCV's own arithmetic) so that the class stays a complete cv::detail::RotationWarper - a pipeline that does call them gets stock
    // behaviour, not an accelerated one, and needs the stock class to exist (the declarations-only test stub has just enough of it).
    cv::Point2f warpPoint(const cv::Point2f& pt, cv::InputArray K, cv::InputArray R) override { return stock_.warpPoint(pt, K, R); }
    void warpBackward(cv::InputArray src, cv::InputArray K, cv::InputArray R, int interp_mode, int border_mode, cv::Size dst_size,
                      cv::OutputArray dst) override {
        stock_.warpBackward(src, K, R, interp_mode, border_mode, dst_size, dst);
    }
    float getScale() const override { return scale_; }
    isx_warper* handle() { return w_.handle(); }
private:
    isx::RotationWarper w_;
    StockWarper stock_;
    float scale_;
};
typedef HipRotationWarper<ISX_WARP_CYLINDRICAL, cv::detail::CylindricalWarper> HipCylindricalWarper;   // W:219  new cv::CylindricalWarper()
typedef HipRotationWarper<ISX_WARP_SPHERICAL, cv::detail::SphericalWarper> HipSphericalWarper;         // B:93   (commented out there)

// cv::detail::Blender over isx_blender.  The base class's prepare(corners, sizes) (W:281) is not virtual: it computes resultRoi and calls
// prepare(Rect), which is overridden here, as in OpenCV's own MultiBandBlender / FeatherBlender.
class HipBlenderBase : public cv::detail::Blender {
public:
    void prepare(cv::Rect dst_roi) override { blender().prepare(to_isx(dst_roi)); }
    // feed(img CV_16SC3, mask CV_8U, tl)  W:302.  Host mats are staged in the blender's own buffers: the caller may release them at
    // once (W:305-308), whichever cycle the blender runs.
    void feed(cv::InputArray img, cv::InputArray mask, cv::Point tl) override {
        cv::Mat i = img.getMat(), m = mask.getMat();
        blender().feed(isx::Mat(i), isx::Mat(m), isx::Point(tl.x, tl.y));
    }
    // blend(dst, dst_mask)  W:313: CV_16SC3 result + CV_8U mask of dst_roi's size
    void blend(cv::InputOutputArray dst, cv::InputOutputArray dst_mask) override {
        int w = 0, h = 0;
        isx::check(isx_blender_result_size(blender().handle(), &w, &h));
        if (win_x1_ > win_x0_) w = win_x1_ - win_x0_;   // HipMultiBandBlender::setWindow: the strip's columns only
        dst.create(h, w, CV_16SC3);
        dst_mask.create(h, w, CV_8U);
        cv::Mat d = dst.getMat(), m = dst_mask.getMat();
        isx::Mat id(d), im(m);
        isx::check(isx_blender_blend(blender().handle(), id.c(), im.c()));
    }
protected:
    virtual isx::Blender& blender() = 0;
    int win_x0_ = 0, win_x1_ = 0;
};

class HipMultiBandBlender : public HipBlenderBase {             // Blender::createDefault(Blender::MULTI_BAND, false)  W:271
public:
    explicit HipMultiBandBlender(int num_bands = 5, int precision = ISX_PREC_I16, int device = 0) : b_(true, num_bands, precision, device) {
        // blend() does all the work and the destination pyramid never exists; 2 = feed() takes a private copy of any DEVICE mat it
        // records (host cv::Mats are staged anyway), so OpenCV's "feed consumes its inputs" contract holds
        isx::check(isx_blender_set_deferred_level0(b_.handle(), 2));
    }
    int numBands() { return b_.numBands(); }
    void setNumBands(int val) { b_.setNumBands(val); }          // mb->setNumBands(4)  W:273
    // not in OpenCV: blend() produces the result's columns [x0, x1) only (x0 a multiple of ISX_WINDOW_GRANULE) - one strip of a
    // panorama that is cut across GPUs; feed only the tiles near the strip (isx_blender_set_window in imagestitch_hip.h)
    void setWindow(int x0, int x1) { b_.setWindow(x0, x1); win_x0_ = x0; win_x1_ = x1; }
protected:
    isx::Blender& blender() override { return b_; }
private:
    isx::MultiBandBlender b_;
};

class HipFeatherBlender : public HipBlenderBase {               // Blender::createDefault(Blender::FEATHER, false): what every demo runs, W:278-280
public:
    explicit HipFeatherBlender(float sharpness = 0.02f, int device = 0) : b_(sharpness, device) {}
    void setSharpness(float val) { b_.setSharpness(val); }      // fb->setSharpness(0.1)  W:280
protected:
    isx::Blender& blender() override { return b_; }
private:
    isx::FeatherBlender b_;
};

class HipNoBlender : public HipBlenderBase {                    // Blender::createDefault(Blender::NO, false)  W:276: cv::detail::Blender itself
public:
    explicit HipNoBlender(int device = 0) : b_(device) {}
protected:
    isx::Blender& blender() override { return b_; }
private:
    isx::NoBlender b_;
};

// Blender::createDefault(type, try_gpu) (W:271, 276, 278) for the three types the reference names, returning what it returns - a
// Ptr<Blender> - so that `blender = isx_cv::createDefaultBlender(Blender::MULTI_BAND);` replaces W:271 token for token
inline cv::Ptr<cv::detail::Blender> createDefaultBlender(int type, bool /*try_gpu*/ = false) {
    if (type == cv::detail::Blender::NO) return cv::Ptr<cv::detail::Blender>(new HipNoBlender());
    if (type == cv::detail::Blender::FEATHER) return cv::Ptr<cv::detail::Blender>(new HipFeatherBlender());
    return cv::Ptr<cv::detail::Blender>(new HipMultiBandBlender());
}

}  // namespace isx_cv

#endif  // IMAGESTITCH_CV_HPP
