// tests/cpp/opencv_stub — NOT OpenCV: the interface of cv::detail::RotationWarper as OpenCV 3.4.2 declares it (the members the adapter
// overrides), and stock warper classes reduced to what the adapter forwards to (never called by the reference's hot path).
// Written from (knowledge of) OpenCV 3.4.2 modules/stitching/include/opencv2/stitching/detail/warpers.hpp, class CV_EXPORTS RotationWarper:
//   virtual ~RotationWarper() {}
//   virtual Point2f warpPoint(const Point2f &pt, InputArray K, InputArray R) = 0;
//   virtual Rect buildMaps(Size src_size, InputArray K, InputArray R, OutputArray xmap, OutputArray ymap) = 0;
//   virtual Point warp(InputArray src, InputArray K, InputArray R, int interp_mode, int border_mode, OutputArray dst) = 0;
//   virtual void warpBackward(...) = 0;  virtual Rect warpRoi(Size src_size, InputArray K, InputArray R) = 0;
//   virtual float getScale() const { return 1.f; }   virtual void setScale(float) {}
// Only the members the reference calls (W:122, W:145-161; B:105,109) are declared below; same -Werror flags as blenders.hpp.
#ifndef ISX_TEST_OPENCV_STUB_WARPERS_HPP
#define ISX_TEST_OPENCV_STUB_WARPERS_HPP
#include <opencv2/core.hpp>
namespace cv { namespace detail {
class RotationWarper {
public:
    virtual ~RotationWarper() {}
    virtual Point2f warpPoint(const Point2f& pt, InputArray K, InputArray R) = 0;
    virtual Rect buildMaps(Size src_size, InputArray K, InputArray R, OutputArray xmap, OutputArray ymap) = 0;
    virtual Point warp(InputArray src, InputArray K, InputArray R, int interp_mode, int border_mode, OutputArray dst) = 0;
    virtual void warpBackward(InputArray src, InputArray K, InputArray R, int interp_mode, int border_mode, Size dst_size, OutputArray dst) = 0;
    virtual Rect warpRoi(Size src_size, InputArray K, InputArray R) = 0;
    virtual float getScale() const { return 1.f; }
    virtual void setScale(float) {}
};
struct StockWarperStub {
    explicit StockWarperStub(float) {}
    Point2f warpPoint(const Point2f&, InputArray, InputArray) { throw std::runtime_error("opencv_stub: warpPoint is not implemented"); }
    void warpBackward(InputArray, InputArray, InputArray, int, int, Size, OutputArray) { throw std::runtime_error("opencv_stub: warpBackward is not implemented"); }
};
struct CylindricalWarper : StockWarperStub { explicit CylindricalWarper(float s) : StockWarperStub(s) {} };
struct SphericalWarper : StockWarperStub { explicit SphericalWarper(float s) : StockWarperStub(s) {} };
}}  // namespace cv::detail
#endif
