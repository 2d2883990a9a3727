/*
 * oracle.h — CPU restatement of the reference's warp + multi-band blend hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under imagestitch_amd/ may include, link, load or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker / the CPU timing baseline.
 *
 * Reference aliases (raw-file line numbers, sources are GB18030 encoded):
 *   W = /root/reference/圆柱面投影变换/圆柱面投影变换/圆柱面投影.cpp
 *   B = /root/reference/图像融合/图像融合/图像融合.cpp
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_map_forward / orc_map_backward (cylindrical): PINNED bit-exactly against W:30-63
 *     compiled verbatim into oracle/_ref/libref_warp.so (oracle/build.sh) and against the
 *     committed vectors tests/golden/cyl_maps.npz generated from it; ROI width 1086 pinned
 *     against the reference's committed images_warped_f[0].bmp.
 *   - warp geometry + remap LINEAR/REFLECT + gain apply: PINNED UP TO THE TIE RULE by the reference's committed
 *     images_warped_f[0].bmp = gain * warp(src2.bmp) (tests/golden/ref_warp_artifact.npz, tests/test_ref_artifact.py):
 *     the artefact came from OpenCV's OpenCL remap (float blend, round-half-even); orc_remap_u8 is the CPU fixed-point
 *     path = the same weighted sum rounded half-up, and differs from the artefact exactly at the ties.
 *   - orc_seam_costs / orc_seam_estimate (+ oracle/dpseam_np.py, the whole finder S:87-1093): PINNED by the reference's
 *     committed mask_seam[0,1].bmp, which the restatement reproduces exactly from the reconstructed inputs
 *     (tests/golden/ref_dpseam_artifact.npz, ref_seam_artifact.npz).
 *   - remap (other modes), pyrDown/pyrUp, MultiBandBlender, SphericalProjector: the arithmetic lives in
 *     OpenCV 3.4.2 (opencv_world342, README.md:23-24), which is neither vendored in
 *     /root/reference nor installed here.  These functions restate OpenCV 3.4.2's published
 *     algorithm (modules/imgproc/src/imgwarp.cpp, pyramids.cpp, modules/stitching/src/
 *     blenders.cpp, include/opencv2/stitching/detail/warpers_inl.hpp) as specified in
 *     SURVEY.md §8(a) A8-A12.  PARITY UNPINNED by reference artefacts: checked by
 *     known-answer tests and by an independent NumPy restatement (oracle/oracle_np.py).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_CYL = 0, ORC_SPH = 1 };
enum { ORC_NEAREST = 0, ORC_LINEAR = 1 };
enum { ORC_BORDER_CONSTANT = 0, ORC_BORDER_REPLICATE = 1, ORC_BORDER_REFLECT = 2,
       ORC_BORDER_WRAP = 3, ORC_BORDER_REFLECT_101 = 4 };
enum { ORC_I16 = 0, ORC_F32 = 1, ORC_F16ACC32 = 2 };

/* scalar helpers exposed for known-answer tests */
int   orc_cvround(float v);                       /* cvRound = cvtss2si (round-half-even)     */
int   orc_f2i_trunc(float v);                     /* static_cast<int>(float) on x86           */
int   orc_border_interpolate(int p, int len, int border);
float orc_f16_round(float v);                     /* f32 -> f16 (RNE) -> f32                  */

/* A2  setCameraParams  W:90-120 */
void orc_camera(const float K[9], const float R[9], float k[9], float rinv[9],
                float r_kinv[9], float k_rinv[9]);
/* A3  mapForward  W:36-45 (cylindrical) / SphericalProjector::mapForward */
void orc_map_forward(int kind, float scale, const float r_kinv[9], float x, float y,
                     float* u, float* v);
/* A4  mapBackward W:46-63 (cylindrical) / SphericalProjector::mapBackward */
void orc_map_backward(int kind, float scale, const float k_rinv[9], float u, float v,
                      float* x, float* y);
/* A5  detectResultRoi W:64-88 (cylindrical: full scan, as the in-tree warper does);
 *     spherical: detectResultRoiByBorder + pole tests (OpenCV SphericalWarper).
 *     minmax (may be NULL) = {min u, min v, max u, max v} before the int casts.              */
void orc_detect_roi(int kind, float scale, const float k[9], const float rinv[9],
                    const float r_kinv[9], int src_w, int src_h, int roi[4], float minmax[4]);
/* A6  buildMaps W:122-144, maps are (roi[3]-roi[1]+1) x (roi[2]-roi[0]+1) */
void orc_build_maps(int kind, float scale, const float k_rinv[9], const int roi[4],
                    float* xmap, float* ymap);
/* A8  cv::remap (W:157) with CV_32FC1 maps.  cn = 1 or 3, steps in BYTES. */
void orc_remap_u8(const uint8_t* src, int sh, int sw, int cn, size_t sstep,
                  uint8_t* dst, int dh, int dw, size_t dstep,
                  const float* xmap, const float* ymap, int interp, int border);
void orc_remap_f32(const float* src, int sh, int sw, int cn, size_t sstep,
                   float* dst, int dh, int dw, size_t dstep,
                   const float* xmap, const float* ymap, int interp, int border);
/* A7  warp W:145-161 on a u8 image: returns corner and fills dst ((roi.h+1) x (roi.w+1)).
 *     Two-step use: call with dst == NULL to get roi, then with a buffer.                    */
void orc_warp_u8(int kind, float scale, const float K[9], const float R[9],
                 const uint8_t* src, int sh, int sw, int cn, int interp, int border,
                 int roi[4], uint8_t* dst);

/* A10 pyrDown / pyrUp (OpenCV pyramids.cpp), contiguous HWC buffers */
void orc_pyr_down_s16(const int16_t* src, int sh, int sw, int cn, int16_t* dst);
void orc_pyr_down_f32(const float* src, int sh, int sw, int cn, float* dst);
void orc_pyr_up_s16(const int16_t* src, int sh, int sw, int cn, int16_t* dst); /* dst 2sh x 2sw */
void orc_pyr_up_f32(const float* src, int sh, int sw, int cn, float* dst);

/* A9, A11, A12 MultiBandBlender */
typedef struct orc_mb orc_mb;
orc_mb* orc_mb_create(int num_bands, int precision);
void    orc_mb_destroy(orc_mb* b);
void    orc_mb_prepare(orc_mb* b, int n, const int* corners_xy, const int* sizes_wh);
int     orc_mb_num_bands(const orc_mb* b);
void    orc_mb_result_size(const orc_mb* b, int* w, int* h);
/* img: int16 HWC (3 ch) when img_is_f32 == 0, float HWC otherwise; mask u8 */
void    orc_mb_feed(orc_mb* b, const void* img, int img_is_f32, const uint8_t* mask,
                    int rows, int cols, int tl_x, int tl_y);
/* level access before blend (parity of the accumulated pyramids) */
void    orc_mb_level(const orc_mb* b, int level, void* lap, float* weight, int* rows, int* cols);
/* dst: int16 HWC when dst_is_f32 == 0 (I16 exact; F32 via saturate_cast), else float HWC     */
void    orc_mb_blend(orc_mb* b, void* dst, int dst_is_f32, uint8_t* dst_mask);

/* A13 in-tree linear-ramp pair blend B:141-717.  Returns 0 on success, 1 when the tiles do
 * not overlap (B:182-183 `return 0`).  pano is panoHe_ x panoBr_ x 3 floats (zero-filled by
 * the callee).  seam_x (may be NULL) gets panoHe_ ints.                                      */
void orc_blend_pair_linear_size(int rows1, int cols1, int rows2, int cols2,
                                int tl1x, int tl1y, int tl2x, int tl2y, int* pr, int* pc);
int  orc_blend_pair_linear(const float* img1, int rows1, int cols1,
                           const float* img2, int rows2, int cols2,
                           int tl1x, int tl1y, int tl2x, int tl2y, float* pano, int* seam_x);

/* N3 mask preparation W:286-301 (cv::dilate MORPH_RECT) and N2 FeatherBlender W:278-281,302,313 */
void orc_dilate_rect_u8(const uint8_t* src, int h, int w, int kw, int kh, uint8_t* dst);
/* N3 GainCompensator::apply W:241-244: multiply(image, gain, image) on n bytes of a CV_8U image, in place */
void orc_gain_apply_u8(uint8_t* img, size_t n, double gain);
void orc_distance_transform_l1(const uint8_t* src, int h, int w, float* dst);
void orc_feather_weight_map(const uint8_t* mask, int h, int w, float sharpness, float* weight);
typedef struct orc_fb orc_fb;
orc_fb* orc_fb_create(float sharpness);
void    orc_fb_destroy(orc_fb* b);
void    orc_fb_prepare(orc_fb* b, int n, const int* corners_xy, const int* sizes_wh);
void    orc_fb_result_size(const orc_fb* b, int* w, int* h);
void    orc_fb_feed(orc_fb* b, const int16_t* img, const uint8_t* mask, int rows, int cols, int tl_x, int tl_y);
void    orc_fb_blend(orc_fb* b, int16_t* dst, uint8_t* dst_mask);

/* Blender::NO = cv::detail::Blender itself (W:276) */
typedef struct orc_nb orc_nb;
orc_nb* orc_nb_create(void);
void    orc_nb_destroy(orc_nb* b);
void    orc_nb_prepare(orc_nb* b, int n, const int* corners_xy, const int* sizes_wh);
void    orc_nb_result_size(const orc_nb* b, int* w, int* h);
void    orc_nb_feed(orc_nb* b, const int16_t* img, const uint8_t* mask, int rows, int cols, int tl_x, int tl_y);
void    orc_nb_blend(orc_nb* b, int16_t* dst, uint8_t* dst_mask);

/* Mat::convertTo where it narrows (W:294: CV_32F -> CV_16S; W:315's input: -> CV_8U) */
void orc_convert_f32_s16(const float* src, size_t n, int16_t* dst);
void orc_convert_f32_u8(const float* src, size_t n, uint8_t* dst);

/* N1, the data-parallel part of the in-tree DP seam finder (S = 动态规划法寻找最佳缝合线.cpp):
 * computeCosts S:733-803 (costFunc_ COLOR) and estimateSeam S:806-957.  Images are HWC 3-channel float (is_u8 == 0)
 * or uint8 (is_u8 == 1), contiguous; labels is the union-sized int32 label image (labels_), label = comp + 1,
 * (rx, ry, rw, rh) = Rect(tls_[comp], brs_[comp]).  A labels_ read outside the union counts as "not label" (the
 * reference reads past the row there).  costV: rh x (rw + 1), costH: (rh + 1) x rw.
 * orc_seam_estimate returns the seam length (0: p2 is not reachable from p1, the reference returns false) and
 * writes the seam points (union coordinates, p1 first) to seam_xy. */
void orc_seam_costs(const void* img1, int rows1, int cols1, const void* img2, int rows2, int cols2, int is_u8,
                    int tl1x, int tl1y, int tl2x, int tl2y, int utlx, int utly,
                    const int32_t* labels, int uh, int uw, int label, int rx, int ry, int rw, int rh,
                    float* costV, float* costH);
int  orc_seam_estimate(const void* img1, int rows1, int cols1, const void* img2, int rows2, int cols2, int is_u8,
                       int tl1x, int tl1y, int tl2x, int tl2y, int utlx, int utly,
                       const int32_t* labels, int uh, int uw, int label, int rx, int ry, int rw, int rh,
                       int p1x, int p1y, int p2x, int p2y, int* seam_xy, int cap, int* is_horizontal);

#ifdef __cplusplus
}
#endif
#endif
