/*
 * imagestitch_hip.h — C-ABI of the MI355X (gfx950) warp + multi-band blend hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one call the
 * reference makes on cv::detail::RotationWarper / cv::detail::Blender (or their in-tree
 * restatements).  Reference files, cited as alias:line (raw-file line numbers):
 *   W = /root/reference/圆柱面投影变换/圆柱面投影变换/圆柱面投影.cpp   (cylindrical warper demo)
 *   B = /root/reference/图像融合/图像融合/图像融合.cpp                 (blend demo)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions across the boundary.
 *   - every function returns an isx_status (0 = OK); isx_last_error() returns a thread-local
 *     human-readable message for the last failure (the analogue of cv::Exception::what()).
 *   - isx_mat mirrors cv::Mat{data,rows,cols,type(),step}; `type` uses OpenCV's numeric type
 *     codes so an adapter can pass mat.type() through unchanged.  `device` = -1 means `data`
 *     is a host pointer (the library stages it through HBM, PCIe-inclusive), >= 0 means
 *     `data` is a HIP device pointer on that device (zero-copy; the measured path).
 *   - output mats are caller-allocated (replaces OutputArray::create, W:128-129,150); the
 *     *_roi / *_result_size queries give the sizes.
 *   - all work is enqueued on the handle's HIP stream (isx_*_set_stream); calls that return
 *     host values (ROI, corner) synchronise that stream, the others are asynchronous when all
 *     mats are device mats.
 */
#ifndef IMAGESTITCH_HIP_H
#define IMAGESTITCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (CV_Assert / CV_Error analogue, W:94-96) ------------------------------ */
typedef enum {
    ISX_OK = 0,
    ISX_ERR_INVALID = 1,     /* bad argument (null pointer, bad enum, bad size)                 */
    ISX_ERR_TYPE = 2,        /* wrong isx_mat type for this call (CV_Assert(type()==...))       */
    ISX_ERR_STATE = 3,       /* call order violated (feed before prepare, blend twice, ...)     */
    ISX_ERR_HIP = 4,         /* a HIP runtime call failed; message carries hipGetErrorString    */
    ISX_ERR_NOMEM = 5,       /* hipMalloc failed                                                */
    ISX_ERR_UNSUPPORTED = 6, /* valid in OpenCV, not implemented on this path                   */
    ISX_ERR_SIZE = 7,        /* caller-allocated output has the wrong rows/cols                 */
    ISX_ERR_PLAN = 8,        /* a planned (sync-free) run saw geometry that differs from plan   */
    ISX_ERR_INTERNAL = 9     /* a C++ exception was stopped at this boundary (never propagated) */
} isx_status;

/* ---- cv::Mat type codes: depth + ((cn-1)<<3), same numbers as OpenCV -------------------- */
enum {
    ISX_8UC1 = 0,   /* CV_8UC1  : masks (W:213-214,232)                                       */
    ISX_8UC3 = 16,  /* CV_8UC3  : source and warped images (W:166-169,229)                    */
    ISX_16SC3 = 19, /* CV_16SC3 : Blender::feed input / blend output (W:294,302,313)          */
    ISX_32SC1 = 4,  /* CV_32SC1 : the seam finder's label image labels_ (S:83)                    */
    ISX_32FC1 = 5,  /* CV_32FC1 : xmap / ymap (W:128-129)                                     */
    ISX_32FC3 = 21  /* CV_32FC3 : float images (W:261; B:143-145)                             */
};

/* cv::InterpolationFlags / cv::BorderTypes values used by W:229,232 */
enum { ISX_INTER_NEAREST = 0, ISX_INTER_LINEAR = 1,
       /* not an OpenCV flag: OR it to ISX_INTER_LINEAR in isx_warper_warp / isx_remap to round the 8-bit bilinear sum half to
        * EVEN, as OpenCV's OpenCL (UMat / T-API) remap does — the arithmetic the reference's committed images_warped_f[0].bmp
        * was produced with.  Default (OpenCV's CPU remap, the reference CPU path): half up.                         */
       ISX_INTER_TIES_EVEN = 0x100 };
enum { ISX_BORDER_CONSTANT = 0, ISX_BORDER_REPLICATE = 1, ISX_BORDER_REFLECT = 2,
       ISX_BORDER_WRAP = 3, ISX_BORDER_REFLECT_101 = 4 };

/* warper kinds: cv::CylindricalWarper (W:219) / cv::SphericalWarper (B:93, commented) */
enum { ISX_WARP_CYLINDRICAL = 0, ISX_WARP_SPHERICAL = 1 };

/* cv::detail::Blender::{NO, FEATHER, MULTI_BAND} (W:271,276,278) */
enum { ISX_BLEND_NO = 0, ISX_BLEND_FEATHER = 1, ISX_BLEND_MULTI_BAND = 2 };

/* pyramid precision of the multi-band blender */
enum {
    ISX_PREC_I16 = 0,      /* OpenCV's arithmetic: int16 Laplacian pyramid, fp32 weights       */
    ISX_PREC_F32 = 1,      /* everything fp32, same order of operations, no short casts        */
    ISX_PREC_F16ACC32 = 2  /* Gaussian levels stored fp16, arithmetic + accumulators fp32      */
};

typedef struct isx_mat {
    void*  data;    /* first byte of row 0                                                     */
    int    rows;
    int    cols;
    int    type;    /* ISX_8UC1 ...                                                            */
    size_t step;    /* bytes between consecutive rows (cv::Mat::step)                          */
    int    device;  /* -1: host pointer; >=0: HIP device pointer on that device                */
} isx_mat;

typedef struct isx_warper  isx_warper;
typedef struct isx_blender isx_blender;

/* ---- library ---------------------------------------------------------------------------- */
const char* isx_last_error(void);            /* thread-local message of the last failure      */
const char* isx_version(void);
int         isx_device_count(int* count);    /* hipGetDeviceCount; ISX_ERR_HIP if no runtime  */

/* ---- warper: replaces warper_creator->create(scale) + RotationWarper (W:217-233) -------- */
/* replaces Ptr<RotationWarper> w = warper_creator->create(float scale)  (W:217-222; the
 * in-tree twin hard-codes `float scale = 2707.47f`, W:30).  `device` is the HIP device.      */
int isx_warper_create(int kind, float scale, int device, isx_warper** out);
int isx_warper_destroy(isx_warper* w);
int isx_warper_set_stream(isx_warper* w, void* hip_stream /* hipStream_t, NULL = default */);

/* setCameraParams (W:90-120): K, R are 3x3 row-major CV_32F.  Writes r_kinv = R*K^-1 (W:108)
 * and k_rinv = K*R^T (W:113) as the library computes them (for inspection / parity tests).   */
int isx_warper_camera(isx_warper* w, const float K[9], const float R[9],
                      float r_kinv[9], float k_rinv[9]);

/* detectResultRoi (W:64-88): full forward scan of every source pixel on the GPU.
 * roi = {tl.x, tl.y, br.x, br.y} (inclusive br, trunc-toward-zero casts, W:83-86).
 * minmax (optional, may be NULL) receives the four float extrema {min u, min v, max u, max v}. */
int isx_warper_roi(isx_warper* w, int src_w, int src_h, const float K[9], const float R[9],
                   int roi[4], float minmax[4]);
/* Opt-in: remember detectResultRoi's result per (K, R, scale, source size) - it is a pure function of them - so that a
 * fixed rig's repeated isx_warper_warp / _roi / _build_maps calls skip the GPU scan and its host round trip (the corner
 * must reach the host before the destination can be sized, W:148-150).  The spherical ROI (host code) is not cached.
 * Off by default: only the LAST result is then remembered - the reference asks for the same ROI twice in a row (warp(img, K, R), then
 * warp(mask, K, R), W:229,232) and the second call returns the first one's answer; any other call computes its ROI as the reference does. */
int isx_warper_set_roi_cache(isx_warper* w, int on);
/* compensator->apply(i, corners[i], images_warped[i], masks_warped[i]) (W:241-244) folded into the fused tile warps that
 * follow (isx_warper_warp_with_mask / _roi / _planned with src_mask == NULL): every byte of the warped IMAGE becomes
 * saturate_cast<uchar>(cvRound((double)byte * gain)) - exactly isx_gain_apply on the warped tile, one pass over it less (the gain acts on
 * the remapped byte, not on the source: remap of scaled pixels is a different number).  gain = 1.0 switches it off.                    */
int isx_warper_set_gain(isx_warper* w, double gain);

/* buildMaps (W:122-144): xmap,ymap are caller-allocated CV_32FC1 of
 * (roi[3]-roi[1]+1) rows x (roi[2]-roi[0]+1) cols (W:128-129).                               */
int isx_warper_build_maps(isx_warper* w, int src_w, int src_h, const float K[9], const float R[9],
                          isx_mat* xmap, isx_mat* ymap, int roi[4]);

/* The map fill of buildMaps (W:133-141) over a rectangle the caller already has - isx_warper_roi's result for this camera, or the
 * rectangle of maps it keeps - without the scan: xmap(v - roi[1], u - roi[0]) = mapBackward(u, v).x, same for ymap.               */
int isx_warper_build_maps_roi(isx_warper* w, const float K[9], const float R[9], const int roi[4], isx_mat* xmap, isx_mat* ymap);

/* Point warp(src,K,R,interp,border,dst) (W:145-161; stock call sites B:105,109): buildMaps +
 * cv::remap (W:157) fused into one gather kernel; the maps are never written to HBM.
 * src: CV_8UC3 / CV_8UC1 (fixed-point bilinear) or CV_32FC3 / CV_32FC1 (float bilinear).
 * dst: caller-allocated, same type, (roi.h+1) x (roi.w+1) (W:150).  corner = roi.tl() (W:160). */
int isx_warper_warp(isx_warper* w, const isx_mat* src, const float K[9], const float R[9],
                    int interp, int border, isx_mat* dst, int corner[2]);

/* The second half of the same call for adapters that must size `dst` themselves (OutputArray::create, W:150): `roi` is what
 * isx_warper_roi just returned for the same (source size, K, R) on this handle, so that ONE reference warp() = one
 * detectResultRoi (W:126) = isx_warper_roi + isx_warper_warp_roi, not two scans.  No scan, no host synchronisation (device
 * mats).  A `roi` that is not detectResultRoi's is not checked: the result is then the remap over that rectangle.          */
int isx_warper_warp_roi(isx_warper* w, const isx_mat* src, const float K[9], const float R[9],
                        int interp, int border, const int roi[4], isx_mat* dst);
/* ... and of the fused image + mask call below.                                                                        */
int isx_warper_warp_with_mask_roi(isx_warper* w, const isx_mat* src_img, const isx_mat* src_mask, const float K[9],
                                  const float R[9], const int roi[4], isx_mat* dst_img, isx_mat* dst_mask);

/* The two calls of W:229 + W:232 on one tile fused: image LINEAR/REFLECT and mask
 * NEAREST/CONSTANT in one pass over the destination.  src_mask may be NULL = all 255
 * (W:213-214).  dst_img may be CV_8UC3 or CV_16SC3 (= warp + convertTo(CV_16S), W:294).      */
int isx_warper_warp_with_mask(isx_warper* w, const isx_mat* src_img, const isx_mat* src_mask,
                              const float K[9], const float R[9],
                              isx_mat* dst_img, isx_mat* dst_mask, int corner[2]);

/* Sync-free variant for captured / batched runs: the caller supplies the ROI it planned with
 * isx_warper_roi; the ROI scan still runs on the GPU and is compared with `planned_roi` on
 * the device; a mismatch raises the sticky flag read by isx_warper_plan_status.               */
int isx_warper_warp_with_mask_planned(isx_warper* w, const isx_mat* src_img,
                                      const isx_mat* src_mask, const float K[9],
                                      const float R[9], const int planned_roi[4],
                                      isx_mat* dst_img, isx_mat* dst_mask);
/* Several tiles' warps in ONE launch.  Between begin and end the fused tile warps of this handle (isx_warper_warp_with_mask / _roi / _planned
 * with src_mask == NULL, device mats, no gain) are collected instead of launched, and isx_warper_end_batch sends them off as one launch of up
 * to 8 tiles per kernel variant (blockIdx.z = tile): the per-image loop of the reference (W:223-233) issues one warp after the other, and on
 * a GPU each launch waits for the last - longest-lived - waves of the one before and pays its own dispatch ramp (~3 us of a 175 us step).
 * Same kernel body, same bits.  The outputs are not written, nor even enqueued, before isx_warper_end_batch returns: use them (feed them)
 * after it.  isx_warper_verify / _join / _plan_status / _set_stream end the collection's pending launches as well; anything that cannot be
 * collected (a caller-supplied source mask, host mats, a gain) is launched at once, behind what was collected so far.                  */
int isx_warper_begin_batch(isx_warper* w);
int isx_warper_end_batch(isx_warper* w);
int isx_warper_plan_status(isx_warper* w, int* mismatches /* synchronises the stream */);
/* The verification scans of planned warps run on an internal side stream.  isx_warper_join makes the
 * handle's stream wait for them without blocking the host — required before hipStreamEndCapture when
 * the planned step is captured into a hipGraph (the forked work must re-join the capturing stream).   */
int isx_warper_join(isx_warper* w);
/* By default the verification scan of a planned warp is enqueued right behind its warp kernel.  With
 * deferred verification the scans are queued and isx_warper_verify enqueues them behind the stream's
 * position at THAT call — e.g. after the last warp of a step, so that the (VALU-bound) scans run under the
 * (memory-bound) pyramid kernels that follow.  join / plan_status flush the queue themselves.            */
int isx_warper_set_deferred_verify(isx_warper* w, int on);
/* Column range of the warped tile (nothing in the reference; the companion of isx_blender_set_window): the isx_warper_warp_with_mask*
 * calls that follow (without a source mask) compute only the columns [col0, col1) of dst_img / dst_mask - the left end rounded down to a
 * block of 64, the right end exactly col1 - and leave the rest of the mats as it is (of a host mat only the computed columns are copied
 * back from its staging buffer; the same holds for the padded last strip of a windowed blend).  For a rank that needs part of a neighbour's tile to blend its strip of a
 * panorama (imagestitch_amd/mosaic.py: tile_columns_for_window gives the range).  (0, 0) = the whole tile again.              */
int isx_warper_set_dst_columns(isx_warper* w, int col0, int col1);
int isx_warper_verify(isx_warper* w);
/* Verification beside a hipGraph instead of inside it (a fork of the verification stream inside a captured step costs the replay 12 us at 4K):
 * isx_warper_discard_pending drops the scans the planned warps of a CAPTURED step queued; after every replay the caller queues the same
 * verifications from the rig alone - isx_warper_queue_verify(source size, K, R, planned ROI): the scan reads no image - and starts them with
 * isx_warper_verify.  isx_warper_plan_status reports a stale plan as for any planned warp.                                           */
int isx_warper_discard_pending(isx_warper* w);
int isx_warper_queue_verify(isx_warper* w, int src_w, int src_h, const float K[9], const float R[9], const int planned_roi[4]);
/* Scheduling hint (nothing in the reference): is the verification of a planned warp of a src_cols x src_rows source under (K, R) a
 * border scan - one workgroup over the 2 (W + H) border pixels, which starts at once, with no event on the handle's stream - (1), or
 * the full detectResultRoi scan of every source pixel (0), which is worth placing under memory-bound work
 * (isx_blender_set_mark_event + isx_warper_verify_after)?  Spherical: always 1.  Cylindrical: 1 when the extrema provably lie on the
 * border (the image in front of the camera, no pole within two pixels of it).                                                          */
int isx_warper_verify_is_light(isx_warper* w, int src_cols, int src_rows, const float K[9], const float R[9], int* light);
/* Same, but the scans start once `hip_event` (a hipEvent_t already recorded, e.g. by
 * isx_blender_set_mark_event during blend()) has completed instead of at the stream's current position.  */
int isx_warper_verify_after(isx_warper* w, void* hip_event);

/* cv::remap(src, dst, xmap, ymap, interp_mode, border_mode) itself (W:157), for callers that keep the maps of a
 * fixed rig (isx_warper_build_maps once, isx_remap per frame).  CV_32FC1 maps; src / dst CV_8UC1, CV_8UC3, CV_32FC1 or
 * CV_32FC3; OpenCV's CPU arithmetic (coordinates quantised to 1/32 pixel, 15-bit fixed-point weights for 8-bit images). */
int isx_remap(const isx_mat* src, const isx_mat* xmap, const isx_mat* ymap, int interp_mode, int border_mode,
              isx_mat* dst, int device, void* hip_stream);

/* ---- blender: replaces Blender::createDefault + MultiBandBlender (W:271-281,302,313) ---- */
/* Blender::createDefault(type, try_gpu) (W:271,276,278) + setNumBands (W:273).
 * type: ISX_BLEND_MULTI_BAND (num_bands default in OpenCV: 5), ISX_BLEND_FEATHER (the blender every
 * reference demo actually runs, W:278-280; num_bands / precision are ignored, sharpness 0.02) or ISX_BLEND_NO
 * (W:276: the base class - feed() is a masked copy, blend() zeroes what no mask covered).            */
int isx_blender_create(int type, int num_bands, int precision, int device, isx_blender** out);
int isx_blender_destroy(isx_blender* b);
int isx_blender_set_stream(isx_blender* b, void* hip_stream);
int isx_blender_set_num_bands(isx_blender* b, int num_bands);   /* mb->setNumBands(n), W:273 */
int isx_blender_num_bands(isx_blender* b, int* num_bands);      /* after prepare: the clamped */
int isx_blender_set_sharpness(isx_blender* b, float sharpness); /* fb->setSharpness(0.1), W:280 */

/* blender->prepare(corners, sizes) (W:281): corners_xy = {x0,y0,x1,y1,...}, sizes_wh likewise */
int isx_blender_prepare(isx_blender* b, int n, const int* corners_xy, const int* sizes_wh);
/* MultiBandBlender::prepare(Rect dst_roi) */
int isx_blender_prepare_roi(isx_blender* b, int x, int y, int width, int height);

/* blender->feed(img [CV_16SC3], mask [CV_8U], tl) (W:302).  img may also be CV_32FC3 in
 * the F32 / F16ACC32 precisions, and CV_8UC3: OpenCV then takes createLaplacePyr's 8-bit branch,
 * whose numbers are those of the CV_16S branch on the converted image (bytes never saturate in
 * pyrDown / pyrUp) - the same path as isx_blender_feed_u8.                                        */
int isx_blender_feed(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y);
/* images_warped.convertTo(CV_16S) (W:261,294) fused into feed: img is CV_8UC3 and is widened
 * to int16 on load; results are identical to converting first and calling isx_blender_feed.   */
int isx_blender_feed_u8(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y);
/* the mask preparation of W:286-301 inside the feed - mask = dilate(seam_mask, getStructuringElement(MORPH_RECT, Size(kw, kh)))
 * & warped_mask is computed straight into the mask buffer the blender keeps for the tile (its private copy under
 * isx_blender_set_deferred_level0 = 2), then blender->feed(img, mask, tl) (W:302) as isx_blender_feed does: identical results, no
 * intermediate mask mat, one pass over the mask less.  Every blender type; elements up to 33 a side (ISX_ERR_UNSUPPORTED beyond:
 * isx_mask_dilate_and + isx_blender_feed); img CV_16SC3 or CV_8UC3.                                                                */
int isx_blender_feed_dilated(isx_blender* b, const isx_mat* img, const isx_mat* seam_mask, const isx_mat* warped_mask, int kw, int kh,
                             int tl_x, int tl_y);

/* Opt-in: deferred mode.  feed() then only RECORDS the tile and blend() does all the work: the
 * Gaussian chains of all tiles (one launch per level), then a collapse chain whose every step
 * gathers the tiles' Laplacians in registers.  The destination Laplacian / weight pyramid
 * (32 B/px of read-modify-write per feed, 16 B/px read back by blend) never touches HBM.
 * Results are identical to the eager path.  THE CONTRACT THAT CHANGES: every DEVICE mat passed to
 * feed()/feed_u8() must stay valid and unmodified until blend() returns (OpenCV's feed() consumes
 * its inputs immediately — the reference clears the fed images before blend(), W:305-308 — so this
 * is not the default).  Host mats are staged in per-tile device buffers owned by the blender:
 * nothing changes for them.  A launch's arguments hold 20 tiles: a cycle of more tiles (up to 4096, one type) runs the same chain with the
 * tiles' descriptors in a table in device memory (isx_blender_last_path: cycle 4; round 4's column strips of at most 20 tiles, cycle 3, remain
 * as ISX_TAB=0), bit-identical; when a tile of another type is fed, when isx_blender_debug_level is called, and for an int16 cycle of CV_8UC3
 * tiles stacked more than 128 deep over one place, the recorded tiles are replayed through the eager path.
 * on = 2 keeps OpenCV's contract: feed() takes a private copy of every DEVICE mat it records, so the caller may release or overwrite
 * the fed mats as soon as feed() returns - the drop-in mode for callers written against cv::detail::Blender (W:286-308).  For CV_8UC3
 * and CV_16SC3 tiles that copy is a by-product of the pass that builds level 1 of the tile's pyramid (one read of the caller's mats per
 * feed, isx_blender_feed_path); other mats take one device-to-device pass on the handle's stream.                                  */
int isx_blender_set_deferred_level0(isx_blender* b, int on);
/* In deferred mode: start each fed tile's Gaussian chain immediately on an internal side stream, so that
 * the (memory-bound) chain of tile t overlaps with whatever the caller enqueues next on the handle's
 * stream — typically the (VALU-bound) warp of tile t+1.  blend() joins the side streams.               */
int isx_blender_set_overlap(isx_blender* b, int on);
/* Scheduling hook, nothing in the reference: a deferred blend() records `hip_event` (hipEvent_t, NULL = off) on its
 * stream right after the pyrDown launch that produces level after_level + 1 of the tile pyramids.  From there to the
 * last collapse step the launches are small and leave most of the GPU idle: the place for unrelated VALU-bound work
 * such as the ROI verification scans (isx_warper_verify_after).                                           */
int isx_blender_set_mark_event(isx_blender* b, void* hip_event, int after_level);
/* Column window (one panorama of many tiles cut into column strips, a strip per GPU, no exchange before the final
 * gather).  Nothing in the reference corresponds to it.  With a window [x0, x1) set - columns of the result, 0 = the left edge of
 * dst_roi, x0 a multiple of ISX_WINDOW_GRANULE - blend() of a deferred cycle (MultiBandBlender or FeatherBlender) computes and writes those columns
 * only, into mats that are x1 - x0 wide (columns past the result's right edge, when x1 exceeds its width, are not touched); every
 * pixel equals the same pixel of the whole blend, bit for bit.  prepare() still gets EVERY tile's corner and size (dst_roi is the
 * whole panorama's); only the tiles that can reach the window need to be fed: those whose fed rectangle - the tile widened by the
 * gap 3 * 2^num_bands on either side, as MultiBandBlender::feed does - comes within 2^(num_bands + 1) columns of the window
 * (imagestitch_amd/mosaic.py: tiles_for_window; for the FeatherBlender simply the tiles that overlap the window).  The window stays
 * set until changed; (0, 0) removes it.                                                                                       */
#define ISX_WINDOW_GRANULE 128
int isx_blender_set_window(isx_blender* b, int x0, int x1);

/* size of the result of blend(): dst_roi_final_ (unpadded union of the fed tiles)             */
int isx_blender_result_size(isx_blender* b, int* width, int* height);
/* Which code path the last isx_blender_blend / _blend_batch of a multi-band blender took - the fast kernels have limits (tile type, tiles
 * per place, tile count: DESIGN.md §3) and nothing else says which side of them a blend ran on.  cycle: 0 eager (the destination pyramid),
 * 1 deferred, 2 deferred inside a batched chain, 3 deferred in column strips (ISX_TAB=0 and more than 20 recorded tiles: the library cuts the
 * result into strips that at most 20 tiles reach and runs the deferred chain per strip - bit-identical to the whole blend), 4 deferred with the
 * tiles in a device-resident table (more than 20 recorded tiles: one chain over all of them); last_step: the kernel of the last collapse step -
 * 0 none (a 0-band blend), 1 k_collapse, 2 k_collapse_gather, 3 k_collapse_roll.  Either pointer may be NULL.                    */
int isx_blender_last_path(isx_blender* b, int* cycle, int* last_step);
/* Layout of level 1 of the tiles' Gaussian pyramids in the last deferred isx_blender_blend (an internal buffer; reported because the step's
 * byte count depends on it): 0 = 16-byte records, 1 = planar (12-byte image records, int16: 6, + a weight plane), 2 = planar with the image
 * channels as three unsigned shorts (fp32 pyramids over CV_8UC3 tiles: level 1 is k / 256 exactly, DESIGN.md §3 "Round 5").               */
int isx_blender_level1_format(isx_blender* b, int* format);
/* Introspection of cycle 4: how many 3.5 KB pieces of tile tables this blender has uploaded so far.  They travel in kernel arguments, in
 * stream order, and only where they differ from what the device holds: a fixed rig uploads on its first blend() and never again.      */
int isx_blender_table_uploads(isx_blender* b, long long* pieces);
/* How the tiles of the last isx_blender_blend were FED in mode 2 (isx_blender_set_deferred_level0 = 2, OpenCV's contract W:302-308).
 * fused_tiles: tiles whose feed() was ONE pass over the caller's CV_8UC3 / CV_16SC3 device mats - level 1 of the tile's pyramid and the
 * private copy out of the same read (round 5; 0: host mats, other tile types, ISX_FEED_FUSE=0).  narrowed: 0 = no private copy was
 * narrowed; 1 = the CV_16SC3 tiles held only byte values (always the case after convertTo(CV_16S) of a warped CV_8UC3 image, W:294),
 * their private copies were kept as CV_8UC3 and the last collapse step ran its CV_8UC3 form - same bits, 3 fewer bytes per pixel written
 * and read; 2 = some tile held a value outside [0, 255]: the copies were widened to CV_16SC3 before the last step (same bits; this
 * blender keeps CV_16SC3 copies from then on).  The check is made on the device while the tile is read; blend() reads its one-word
 * answer from pinned memory after it has enqueued everything that does not depend on it - no stream synchronisation.  Either pointer
 * may be NULL.                                                                                                                    */
int isx_blender_feed_path(isx_blender* b, int* fused_tiles, int* narrowed);
/* Narrowed private copies (mode 2, CV_16SC3 device tiles) make blend() WAIT FOR THE GPU once: before it picks the last step's kernel the
 * calling thread polls the pinned word that the first launch of blend()'s own chain publishes, i.e. it waits until the GPU has worked off
 * whatever the stream held in front of that launch (no hipStreamSynchronize, but with a deep queue on the stream the host stalls for the
 * backlog, and a stream gated behind something this thread would only release AFTER blend() returns - a host callback, a host-signalled
 * event - would never reach the launch: the wait gives up into a stream synchronisation after 20 s).  on = 0 switches the narrowing off
 * for this blender (what the environment variable ISX_FEED_NARROW=0 does for the process): private copies of CV_16SC3 tiles stay
 * CV_16SC3, blend() enqueues and returns without looking at the device; same bits, 3 more bytes per pixel written and read (a 4K pair:
 * the last step 50 -> 60 us).  on = 1 (default) allows it again.  Call it before the first feed of a cycle (ISX_ERR_STATE otherwise).
 * hipGraph capture never narrows.                                                                                                    */
int isx_blender_set_narrow_copies(isx_blender* b, int on);
/* blender->blend(result, result_mask) (W:313).  dst: CV_16SC3 (I16: exact; F32: saturate_cast
 * round-half-even), CV_32FC3 (F32/F16ACC32 only) or CV_8UC3 (= blend to CV_16SC3 followed by
 * result.convertTo(CV_8U), what imwrite (W:315) does to the panorama); dst_mask: CV_8UC1.  Releases the pyramids:
 * prepare must be called again before the next feed (as in OpenCV).                           */
int isx_blender_blend(isx_blender* b, isx_mat* dst, isx_mat* dst_mask);
/* blend() of n blenders at once - the per-image loop of the reference's main() (W:223-233, 285-302, 313) run for a BATCH of independent
 * mosaics (BASELINE configs 3 and 4: 16 / 4 pairs per step).  Blenders that are in the deferred cycle (isx_blender_set_deferred_level0)
 * with the same precision, band count, tile type, device and stream share ONE chain of launches, up to 6 mosaics / 20 tiles per chain:
 * every level's Gaussian chain and collapse step is one launch for all of them (the mosaic is the grid's z), so the levels that
 * are launch-latency-bound for one pair run at P times the waves.  Every mosaic is bit for bit what isx_blender_blend gives; blenders
 * that do not qualify are blended one by one.  dsts / dst_masks: n mats (dst_masks may be NULL).                                   */
int isx_blender_blend_batch(isx_blender** bs, int n, isx_mat* dsts, isx_mat* dst_masks);

/* introspection for parity tests: copy destination pyramid level `level` (after feeds, before
 * blend) to host buffers.  lap: rows*cols*3 of int16 (I16) or float (F32/F16ACC32); weight:
 * rows*cols float.  Either may be NULL.  rows/cols are always written.                        */
int isx_blender_debug_level(isx_blender* b, int level, void* lap, float* weight,
                            int* rows, int* cols);

/* ---- mask preparation between seam finding and feed (W:286-301) ----------------------------- */
/* dilate(masks_seam[k], element) with element = getStructuringElement(MORPH_RECT, Size(kw, kh))
 * (W:286,295) followed by `& masks_warped[k]` (W:299; `other` may be NULL = dilate only).          */
int isx_mask_dilate_and(const isx_mat* mask, const isx_mat* other, int kw, int kh, isx_mat* out,
                        int device, void* hip_stream);

/* ---- exposure compensation, the per-pixel part (W:241-244) ------------------------------------ */
/* compensator->apply(i, corners[i], images_warped[i], masks_warped[i]) of the GainCompensator every demo creates
 * (W:238-239): multiply(image, gain, image) in place on a CV_8UC3 (or CV_8UC1) image, gain = gains_(i, 0) as
 * computed by compensator->feed (a small linear solve on the host, not part of this library).              */
int isx_gain_apply(isx_mat* image, double gain, int device, void* hip_stream);

/* ---- the glue conversions of the reference's main() ---------------------------------- */
/* src.convertTo(dst, dst.type()) with alpha = 1, beta = 0 between the CV_8U, CV_16S and CV_32F depths, same channel count:
 * images_warped[i].convertTo(images_warped_f[i], CV_32F) (W:261), images_warped_f[k].convertTo(images_warped_s[k], CV_16S) (W:294; CV_8U ->
 * CV_16S is the two composed, exact) and result.convertTo(CV_8U) (W:315's input).  Widening is exact; narrowing is OpenCV's
 * saturate_cast (float: cvRound = round-half-even, NaN / overflow -> INT_MIN, then the clamp).  dst is caller-allocated.      */
int isx_convert_to(const isx_mat* src, isx_mat* dst, int device, void* hip_stream);

/* ---- DP seam finder, its data-parallel part (S = 动态规划法寻找最佳缝合线.cpp) ------------------------- */
/* estimateSeam(image1, image2, tl1, tl2, comp, p1, p2, seam, isHorizontal) S:806-957 incl. computeCosts S:733-803
 * (costFunc_ COLOR, what `new DpSeamFinder(DpSeamFinder::COLOR)` W:253 / S:71-72 runs): the cost maps and the dynamic
 * programme run on the GPU, direction choice and backtracking on the host.  The component analysis around it
 * (findComponents, findEdges, resolveConflicts, getSeamTips, updateLabelsUsingSeam) stays with the caller and supplies
 * `labels` (labels_, CV_32SC1, union-sized), `label` = comp + 1, roi = {x, y, width, height} of Rect(tls_[comp],
 * brs_[comp]) and the tips p1, p2 (union coordinates).  Images: both CV_32FC3 (W:261) or both CV_8UC3.
 * seam_xy receives *seam_len points (x, y), p1 first; *seam_len = 0 when p2 is not reachable (`return false`).   */
int isx_seam_estimate(const isx_mat* image1, const isx_mat* image2, int tl1_x, int tl1_y, int tl2_x, int tl2_y,
                      int union_tl_x, int union_tl_y, const isx_mat* labels, int label, const int roi[4],
                      int p1_x, int p1_y, int p2_x, int p2_y, int* seam_xy, int cap, int* seam_len,
                      int* is_horizontal, int device, void* hip_stream);

/* seam_finder->find(images_warped_f, corners, masks_seam) of the in-tree DP seam finder as a whole (S:87-124 `find`, as
 * the S demo calls it at S:1192; W:253 is the stock `DpSeamFinder(DpSeamFinder::COLOR)` it restates): every pair of
 * images, last pair first; component / contour / graph logic on the host, the cost maps and the dynamic programme of every
 * estimateSeam on the GPU.  images: n mats, all CV_32FC3 (W:261) or all CV_8UC3, host or device; masks: n CV_8UC1 mats
 * of the images' sizes, edited in place.                                                                            */
int isx_dp_seam_find(int num_images, const isx_mat* images, const int* corners_xy, isx_mat* masks, int device,
                     void* hip_stream);
/* isx_dp_seam_find / isx_seam_estimate keep their work images (about 6-10 B per union pixel on the host, the staged images,
 * cost maps and DP records on the device) per calling thread between calls; this returns all of it (the analogue of the
 * DpSeamFinder going out of scope, S:1188-1192).  PER THREAD: the state is thread-local and is NOT freed when a thread ends (the
 * HIP runtime may already be gone then) - a thread-pool caller calls this on every worker thread before that thread exits, or
 * leaks one finder (tens of MB at 4K) per thread.                                                                    */
int isx_dp_seam_release(void);

/* ---- on-disk format either side of the path: .bmp (W:166 imread, W:155-156,315 imwrite) ----------- */
/* Reading: uncompressed Windows bitmaps only (the reference's inputs and committed artefacts are BMPs).
 * isx_bmp_read = cv::imread(path) with IMREAD_COLOR: `out` is a CV_8UC3 mat (host or device) of the size
 * isx_bmp_size reports; 8-bit paletted files are expanded through their palette.  isx_bmp_write = cv::imwrite
 * for CV_8UC3 (24-bit) and CV_8UC1 (8-bit, grey palette), host or device mats.                              */
int isx_bmp_size(const char* path, int* rows, int* cols);
int isx_bmp_read(const char* path, isx_mat* out);
int isx_bmp_write(const char* path, const isx_mat* img);
/* cv::imwrite("pano.jpg", result) (B:1132, S:1282): baseline sequential JFIF, 8-bit, the Annex K Huffman tables, 4:2:0
 * chroma for CV_8UC3 (BGR), one component for CV_8UC1; quality 1..100 scales the Annex K quantisation tables as libjpeg does
 * (OpenCV's default is 95).  Any JPEG decoder reads the file; it is not libjpeg's byte stream.  Host or device mats.   */
int isx_jpeg_write(const char* path, const isx_mat* img, int quality);
/* cv::imread(path) (IMREAD_COLOR) for .jpg: baseline / extended-sequential / progressive Huffman JPEG, 8-bit, grey or YCbCr (any integer
 * sampling; 4:4:4, 4:2:2 and 4:2:0 with libjpeg's "fancy" upsampling), restart intervals, interleaved or one scan per component.  The
 * arithmetic is libjpeg's (accurate integer IDCT, its upsampling and colour conversion): what cv::imread hands back.  Arithmetic-coded,
 * lossless, 12-bit and CMYK files, and progressive files whose scans stop short of full precision (libjpeg would smooth those):
 * ISX_ERR_UNSUPPORTED.  `out` is a CV_8UC3 mat (host or device) of the size isx_jpeg_size reports.                                     */
int isx_jpeg_size(const char* path, int* rows, int* cols);
int isx_jpeg_read(const char* path, isx_mat* out);

/* ---- the reference's in-tree single-band seam-ramp blend (B:141-717) --------------------- */
/* images1/images2: CV_32FC3 warped tiles (B:143-145), tl1/tl2 their corners (B:148-149),
 * pano: caller-allocated CV_32FC3 of isx_blend_pair_linear_size().  seam_x (optional, may be
 * NULL) receives the greedy seam's x per row (B:268-307), panoHe_ entries.                    */
int isx_blend_pair_linear_size(int rows1, int cols1, int rows2, int cols2,
                               int tl1_x, int tl1_y, int tl2_x, int tl2_y,
                               int* pano_rows, int* pano_cols);
int isx_blend_pair_linear(const isx_mat* images1, const isx_mat* images2,
                          int tl1_x, int tl1_y, int tl2_x, int tl2_y,
                          isx_mat* pano, int* seam_x, int device, void* hip_stream);
/* isx_blend_pair_linear keeps its work buffers (cost map, chunk maps, seam, weight maps: 41 MB for a 4K pair) per calling thread
 * between calls; this returns them.  PER THREAD, and not freed at thread exit: call it on every worker thread that made the call
 * before that thread ends (see isx_dp_seam_release). */
int isx_blend_pair_linear_release(void);

/* ---- assembling a batch of mosaics across the GPUs of a node (no counterpart in the reference; BASELINE config 4) ---------------- */
/* One process per GPU.  The independent pairs of a batch are partitioned across the ranks (no collective while blending); each rank
 * writes its blended mosaics into ONE packed send block, and every rank ends up with all blocks: an all-gather over RCCL (xGMI).
 * The library loads librccl.so.1 at the first of these calls (dlopen; inside a torch process that is torch's own copy).
 *   isx_gather_unique_id : rank 0 creates the 128-byte rendezvous id and hands it to the other ranks by any means it has (a file,
 *                          a socket, MPI_Bcast, torch.distributed.broadcast_object_list).
 *   isx_gather_create    : collective over all ranks; `device` is this rank's GPU.  id == NULL: no RCCL communicator is created (no
 *                          collective call needed) and only the direct schedule below is available.
 *   isx_gather_all       : ONE all-gather of `bytes` bytes from `send` into recv[rank * bytes ...) on `hip_stream`.
 *   isx_gather_chunk     : the same block gathered chunk by chunk (a chunk = one pair's mosaic, [offset, offset + bytes) of the send
 *                          block): enqueued on the handle's own communication stream behind `ready_event` (a hipEvent_t the caller
 *                          recorded after enqueueing that pair's blend; NULL = now), so the transfer of pair p runs under the blends
 *                          of the pairs after it.  Chunks land rank-major PER CHUNK: isx_gather_chunk_ptr gives the address of
 *                          (rank, chunk) inside recv_base, which must hold world * block_bytes bytes.  Every rank must post the
 *                          same chunks in the same order.
 *   isx_gather_wait      : makes `hip_stream` wait (without blocking the host) for every chunk enqueued so far;
 *   isx_gather_synchronize blocks the host until they are done.                                                             */
typedef struct isx_gather isx_gather;
int isx_gather_unique_id(unsigned char id[128]);
int isx_gather_create(int world, int rank, const unsigned char id[128], int device, isx_gather** out);
/* isx_gather_destroy frees the p2p receive buffer (isx_gather_p2p_alloc) and unmaps the peers': no other rank may still be copying into
 * this rank's buffer - barrier across the ranks first, as after any one-sided put - and no view of the buffer may be used afterwards. */
int isx_gather_destroy(isx_gather* g);
int isx_gather_info(const isx_gather* g, int* world, int* rank);
int isx_gather_all(isx_gather* g, const void* send, size_t bytes, void* recv, void* hip_stream);
int isx_gather_chunk(isx_gather* g, const void* send_base, size_t block_bytes, size_t offset, size_t bytes, void* recv_base,
                     void* ready_event);
int isx_gather_chunk_ptr(const isx_gather* g, void* recv_base, size_t offset, size_t bytes, int rank, void** ptr);
int isx_gather_wait(isx_gather* g, void* hip_stream);
int isx_gather_synchronize(isx_gather* g);
/* The direct schedule beside the collective (xGMI is point to point: a rank's block crosses each of its world - 1 links once whatever
 * the schedule; here the rank issues the world - 1 device-to-device copies itself, one stream per destination, instead of leaving rings
 * and channels to RCCL - bench.py --gather-backend p2p against torch / isx tells the schedule from the links):
 *   isx_gather_p2p_alloc : this rank's receive buffer (world x block bytes, its own hipMalloc) and its 64-byte HIP IPC handle, which
 *                          travels to the other ranks by whatever the caller has (as the unique id does);
 *   isx_gather_p2p_open  : maps every rank's buffer (handles: world x 64 bytes, in rank order), creates the per-destination streams;
 *                          a failure part of the way is rolled back (mappings closed, streams destroyed): the call can be repeated;
 *   isx_gather_p2p_chunk : bytes [offset, offset + bytes) of the send block copied into every rank's buffer (this rank's own
 *                          included) behind `ready_event`, same layout as isx_gather_chunk (isx_gather_chunk_ptr applies);
 *   isx_gather_p2p_wait  : makes `hip_stream` wait for this rank's copies enqueued so far (its send block may then be rewritten);
 *                          arrival on the destination ranks is the callers' to establish, as after any one-sided put.                 */
int isx_gather_p2p_alloc(isx_gather* g, size_t bytes, void** ptr, unsigned char handle[64]);
int isx_gather_p2p_open(isx_gather* g, const unsigned char* handles);
int isx_gather_p2p_chunk(isx_gather* g, const void* send_base, size_t block_bytes, size_t offset, size_t bytes, void* ready_event);
int isx_gather_p2p_wait(isx_gather* g, void* hip_stream);
int isx_gather_p2p_synchronize(isx_gather* g);

/* ---- self-test --------------------------------------------------------------------------------- */
/* The fused warp kernel divides x / z and y / z with one shared reciprocal and the hardware division's own recurrence
 * written out in packed FMAs (csrc/warp.hip, k_warp_tile).  This compares that recurrence with the compiler's IEEE division
 * on n pseudo-random operand pairs of the range the kernel admits to it; *mismatches must come back 0.              */
int isx_selftest_division(int device, int n, unsigned long long seed, int* mismatches);
/* detectResultRoi (W:64-88; SphericalWarper's border form) computed on the host alone - what isx_warper_roi returns for the cameras whose
 * extrema provably lie on the source's border (every spherical camera; a cylindrical one with the image in front of the camera and no pole
 * of the cylinder near it: every rig of the reference): the 2 (W + H) border pixels ranked on the caller's thread by two monotone stand-ins
 * (csrc/roihost.cpp, AVX2), the pixels within a tolerance of the four extrema evaluated with mapForward (W:36-45) and the host's libm.
 * Needs no device: the CPU test-suite compares it with the oracle's scan of every source pixel.  isa: 0 = the code isx_warper_roi runs
 * (AVX-512F where the CPU has it, else AVX2, else scalar), 1 = its scalar form, 2 = its AVX2 form.  ISX_ERR_UNSUPPORTED for a cylindrical camera outside that proof (isx_warper_roi scans every pixel on the GPU there). */
int isx_selftest_roi_host(int kind, float scale, const float K[9], const float R[9], int src_w, int src_h, int isa, int roi[4], float minmax[4]);
/* Every entry of this header is a function-try-block: a C++ exception raised underneath it (std::bad_alloc of a host container, a
 * std::length_error, anything a future change throws) is stopped there and comes back as a status - ISX_ERR_NOMEM for the two allocation
 * failures, ISX_ERR_INTERNAL otherwise, the text in isx_last_error() - never as an exception in the caller's frames (the
 * reference's own errors are cv::Exceptions, W:94-96).  This entry throws `kind` from inside such a block so that the barrier can be
 * tested without a GPU: 0 std::bad_alloc, 1 std::length_error (vector::reserve), 2 a failing 2^62-byte allocation, 3 std::runtime_error,
 * 4 a thrown int, 5 std::out_of_range; any other kind returns ISX_OK.                                                            */
int isx_selftest_exception_barrier(int kind);

/* ---- per-kernel HIP-event timing (feeds bench.py's roofline object) ----------------------- */
/* When enabled every kernel launch is bracketed by hipEvents on its own stream.               */
int isx_profile_enable(int on);
int isx_profile_reset(void);
/* restrict the bracketing to one kernel name (NULL / "" = all): lets bench.py time the dominant
 * kernel inside the timed region without perturbing the other launches.                          */
int isx_profile_filter(const char* kernel_name);
/* bracket only every `every`-th launch that passes the filter (1 = all): a bracketed launch (start / stop events on the
 * kernel's own dispatch) is serialised against its neighbours, which a benchmark's timed region should pay on a sample,
 * not on every step.  launches / total_ms / alg_bytes then count the bracketed launches only.                          */
int isx_profile_sample(int every);
/* number of distinct kernel names seen; then name / launches / total milliseconds by index.
 * isx_profile_collect() synchronises the device and folds pending events into the totals.     */
int isx_profile_collect(void);
int isx_profile_count(int* n);
int isx_profile_entry(int index, const char** name, long long* launches, double* total_ms,
                      double* alg_bytes /* algorithmic bytes summed over launches */);

#ifdef __cplusplus
}
#endif
#endif /* IMAGESTITCH_HIP_H */
