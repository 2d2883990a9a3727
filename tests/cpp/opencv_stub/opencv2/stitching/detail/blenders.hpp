// tests/cpp/opencv_stub — NOT OpenCV: the interface of cv::detail::Blender as OpenCV 3.4.2 declares it.  prepare(corners, sizes) is the
// base class's own (non-virtual) member: resultRoi(corners, sizes) -> prepare(Rect).
// Written from (knowledge of) OpenCV 3.4.2 modules/stitching/include/opencv2/stitching/detail/blenders.hpp, class CV_EXPORTS Blender:
//   virtual ~Blender() {}                                                        same
//   enum { NO, FEATHER, MULTI_BAND };                                             same
//   static Ptr<Blender> createDefault(int type, bool try_gpu = false);            not declared here (the adapter is constructed directly)
//   void prepare(const std::vector<Point> &corners, const std::vector<Size> &sizes);   same signature; body = blenders.cpp Blender::prepare
//   virtual void prepare(Rect dst_roi);                                           pure here (3.4.2 has a body the adapter overrides)
//   virtual void feed(InputArray img, InputArray mask, Point tl);                 same, pure here
//   virtual void blend(InputOutputArray dst, InputOutputArray dst_mask);          same, pure here
// Compiled with -Werror=suggest-override -Werror=overloaded-virtual (tests/test_gpu_cpp_mirror.py): a drifted signature in
// include/imagestitch_cv.hpp fails the build instead of silently declaring a new virtual.
#ifndef ISX_TEST_OPENCV_STUB_BLENDERS_HPP
#define ISX_TEST_OPENCV_STUB_BLENDERS_HPP
#include <opencv2/core.hpp>
#include <algorithm>
#include <climits>
namespace cv { namespace detail {
class Blender {
public:
    virtual ~Blender() {}
    enum { NO, FEATHER, MULTI_BAND };
    void prepare(const std::vector<Point>& corners, const std::vector<Size>& sizes) {
        Point tl(INT_MAX, INT_MAX), br(INT_MIN, INT_MIN);
        for (size_t i = 0; i < corners.size(); ++i) {
            tl.x = std::min(tl.x, corners[i].x); tl.y = std::min(tl.y, corners[i].y);
            br.x = std::max(br.x, corners[i].x + sizes[i].width); br.y = std::max(br.y, corners[i].y + sizes[i].height);
        }
        prepare(Rect(tl.x, tl.y, br.x - tl.x, br.y - tl.y));
    }
    virtual void prepare(Rect dst_roi) = 0;
    virtual void feed(InputArray img, InputArray mask, Point tl) = 0;
    virtual void blend(InputOutputArray dst, InputOutputArray dst_mask) = 0;
};
}}  // namespace cv::detail
#endif
