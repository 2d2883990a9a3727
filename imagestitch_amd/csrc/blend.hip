// blend.hip — multi-band Laplacian blender for MI355X (gfx950): Gaussian/Laplacian pyramid build,
// per-band weighted accumulation, normalise + collapse.  Replaces OpenCV 3.4.2
// cv::detail::MultiBandBlender as the reference calls it (W:271-273,281,302,313; spec of the
// arithmetic: SURVEY.md §8(a) A9-A12).  HBM-bound stencil work: no MFMA, 16-byte pixel records,
// coalesced row-major access, LDS tiles with halo, wavefront shuffles for the horizontal 5-tap.
//
// Data layout in HBM (per pyramid level, row-major, pitch == cols):
//   F32      : tile Gaussian levels and destination levels are float4 {b,g,r,weight}
//   F16ACC32 : tile Gaussian levels are 4 x f16 {b,g,r,weight} (8 B); destination levels float4
//   I16      : short4 {b,g,r,0} + a separate float plane for the weight (OpenCV's CV_16SC3 + CV_32F)
// Level 0 of a fed tile is never materialised: the level-0 kernels read the caller's image + mask
// through the copyMakeBorder index maps (BORDER_REFLECT image, BORDER_CONSTANT weight).
//
// Kernels (one launch per level):
//   k_pyr_down  : G_{k+1} = pyrDown(G_k)           (image + weight in one pass)
//   k_lap_acc   : dst_k[rc] += cast((G_k - pyrUp(G_{k+1})) * W_k), dstW_k[rc] += W_k
//   k_top_acc   : dst_L[rc] += cast(G_L * W_L), dstW_L[rc] += W_L
//   k_collapse  : out_{k-1} = sat(pyrUp(out_k) + norm(dst_{k-1})); last level writes the caller's mat
#include "isx_device.hpp"
#include "isx_internal.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <new>

using namespace isx;
using namespace isxd;

namespace {

enum { M_I16 = ISX_PREC_I16, M_F32 = ISX_PREC_F32, M_F16 = ISX_PREC_F16ACC32 };
enum { SK_LEVEL = -1, SK_U8 = 0, SK_S16 = 1, SK_F32 = 2 };

constexpr float WEIGHT_EPS = 1e-5f;

// ------------------------------------------------------------------------------------------------
// pixel record in registers: image channels in the work type (int for I16, float otherwise)
// ------------------------------------------------------------------------------------------------
template <int M> struct WorkT { using t = float; };
template <> struct WorkT<M_I16> { using t = int; };

template <int M>
struct Px {
    typename WorkT<M>::t c0, c1, c2;
    float w;
};

// one pyramid level in HBM
struct LevelBuf {
    void* img;   // float4* | ushort4*(f16 bits) | short4*
    float* wgt;  // I16 only
    int rows, cols;
};

// the caller's tile as level 0 of its pyramid (copyMakeBorder never materialised)
struct Src0 {
    const unsigned char* img;
    size_t img_step;
    const unsigned char* mask;
    size_t mask_step;
    int rows, cols;     // tile size
    int top, left;      // border offsets of copyMakeBorder
    int height, width;  // padded size = level-0 size of the tile pyramid
};

template <int M, bool DST>
__device__ __forceinline__ Px<M> load_px(const LevelBuf& L, int x, int y) {
    size_t i = (size_t)y * L.cols + x;
    Px<M> p;
    if constexpr (M == M_I16) {
        short4 v = ((const short4*)L.img)[i];
        p.c0 = v.x; p.c1 = v.y; p.c2 = v.z;
        p.w = L.wgt[i];
    } else if constexpr (M == M_F16 && !DST) {
        ushort4 v = ((const ushort4*)L.img)[i];
        p.c0 = h2f_bits(v.x); p.c1 = h2f_bits(v.y); p.c2 = h2f_bits(v.z); p.w = h2f_bits(v.w);
    } else {
        float4 v = ((const float4*)L.img)[i];
        p.c0 = v.x; p.c1 = v.y; p.c2 = v.z; p.w = v.w;
    }
    return p;
}

template <int M, bool DST>
__device__ __forceinline__ void store_px(const LevelBuf& L, int x, int y, const Px<M>& p) {
    size_t i = (size_t)y * L.cols + x;
    if constexpr (M == M_I16) {
        ((short4*)L.img)[i] = make_short4((short)p.c0, (short)p.c1, (short)p.c2, 0);
        L.wgt[i] = p.w;
    } else if constexpr (M == M_F16 && !DST) {
        ((ushort4*)L.img)[i] = make_ushort4(f2h_bits(p.c0), f2h_bits(p.c1), f2h_bits(p.c2), f2h_bits(p.w));
    } else {
        ((float4*)L.img)[i] = make_float4(p.c0, p.c1, p.c2, p.w);
    }
}

// level-0 pixel of the tile pyramid at padded coordinates (x, y) in [0,width) x [0,height)
template <int M, int SK>
__device__ __forceinline__ Px<M> load_src0(const Src0& s, int x, int y) {
    int yr = y - s.top, xr = x - s.left;
    int sy = reflect(yr, s.rows), sx = reflect(xr, s.cols);   // copyMakeBorder(BORDER_REFLECT)
    Px<M> p;
    float v0, v1, v2;
    if constexpr (SK == SK_U8) {
        const unsigned char* q = s.img + (size_t)sy * s.img_step + (size_t)sx * 3;
        v0 = q[0]; v1 = q[1]; v2 = q[2];
    } else if constexpr (SK == SK_S16) {
        const short* q = (const short*)(s.img + (size_t)sy * s.img_step) + (size_t)sx * 3;
        v0 = q[0]; v1 = q[1]; v2 = q[2];
    } else {
        const float* q = (const float*)(s.img + (size_t)sy * s.img_step) + (size_t)sx * 3;
        v0 = q[0]; v1 = q[1]; v2 = q[2];
    }
    if constexpr (M == M_I16) {
        if constexpr (SK == SK_F32) { p.c0 = sat_s16(cvround_x86(v0)); p.c1 = sat_s16(cvround_x86(v1)); p.c2 = sat_s16(cvround_x86(v2)); }
        else { p.c0 = (int)v0; p.c1 = (int)v1; p.c2 = (int)v2; }
    } else { p.c0 = v0; p.c1 = v1; p.c2 = v2; }
    // weight = mask * (float)(1./255.), copyMakeBorder(BORDER_CONSTANT 0)
    bool inside = (unsigned)yr < (unsigned)s.rows && (unsigned)xr < (unsigned)s.cols;
    p.w = inside ? (float)s.mask[(size_t)yr * s.mask_step + xr] * (float)(1. / 255.) : 0.f;
    return p;
}

template <int M, int SK>
__device__ __forceinline__ Px<M> load_any(const Src0& s0, const LevelBuf& L, int x, int y) {
    if constexpr (SK == SK_LEVEL) return load_px<M, false>(L, x, y);
    else return load_src0<M, SK>(s0, x, y);
}

template <int M>
__device__ __forceinline__ Px<M> shfl_up1(const Px<M>& p) {
    Px<M> r;
    r.c0 = __shfl_up(p.c0, 1); r.c1 = __shfl_up(p.c1, 1); r.c2 = __shfl_up(p.c2, 1); r.w = __shfl_up(p.w, 1);
    return r;
}
template <int M>
__device__ __forceinline__ Px<M> shfl_down1(const Px<M>& p) {
    Px<M> r;
    r.c0 = __shfl_down(p.c0, 1); r.c1 = __shfl_down(p.c1, 1); r.c2 = __shfl_down(p.c2, 1); r.w = __shfl_down(p.w, 1);
    return r;
}

// the [1 4 6 4 1] tap in the association pyramids.cpp uses: c*6 + (l1 + r1)*4 + l2 + r2
template <class T>
__device__ __forceinline__ T tap5(T c, T l1, T r1, T l2, T r2) { return c * 6 + (l1 + r1) * 4 + l2 + r2; }

// ------------------------------------------------------------------------------------------------
// k_pyr_down: one block = 64 x 16 outputs.  Each wave walks input rows; a lane loads the two
// input pixels (2x, 2x+1) of its output column, the other three taps come from the neighbour
// lanes by wavefront shuffle (lanes 0 / 63 fetch their missing neighbours themselves).  The
// row-filtered tile (35 x 64 records, 2-row halo each side) is staged in LDS, then the column
// filter runs out of LDS.
// ------------------------------------------------------------------------------------------------
constexpr int PD_TY = 16;
constexpr int PD_NR = 2 * PD_TY + 3;

template <int M, int SK>
__global__ __launch_bounds__(256) void k_pyr_down(Src0 s0, LevelBuf src, LevelBuf dst) {
    using WT = typename WorkT<M>::t;
    __shared__ Px<M> hb[PD_NR][WAVE];
    const int sw = (SK == SK_LEVEL) ? src.cols : s0.width;
    const int sh = (SK == SK_LEVEL) ? src.rows : s0.height;
    const int dw = dst.cols, dh = dst.rows;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ox = blockIdx.x * WAVE + lane, oy0 = blockIdx.y * PD_TY;
    const int oxc = min(ox, dw - 1);
    const int cA = reflect101(2 * oxc, sw), cB = reflect101(2 * oxc + 1, sw);
    const int cL2 = reflect101(2 * oxc - 2, sw), cL1 = reflect101(2 * oxc - 1, sw), cR = reflect101(2 * oxc + 2, sw);
    for (int r = wv; r < PD_NR; r += 4) {
        int iy = reflect101(2 * oy0 - 2 + r, sh);
        Px<M> A = load_any<M, SK>(s0, src, cA, iy);
        Px<M> B = load_any<M, SK>(s0, src, cB, iy);
        Px<M> Am = shfl_up1<M>(A), Bm = shfl_up1<M>(B), Ap = shfl_down1<M>(A);
        // a lane whose left/right neighbour is not the adjacent output column reloads those taps
        // (lane 0, lane 63, and the clamped lanes past the right image edge)
        if (lane == 0) { Am = load_any<M, SK>(s0, src, cL2, iy); Bm = load_any<M, SK>(s0, src, cL1, iy); }
        if (lane == 63 || ox >= dw - 1) Ap = load_any<M, SK>(s0, src, cR, iy);
        Px<M> h;
        h.c0 = tap5<WT>(A.c0, Bm.c0, B.c0, Am.c0, Ap.c0);
        h.c1 = tap5<WT>(A.c1, Bm.c1, B.c1, Am.c1, Ap.c1);
        h.c2 = tap5<WT>(A.c2, Bm.c2, B.c2, Am.c2, Ap.c2);
        h.w = tap5<float>(A.w, Bm.w, B.w, Am.w, Ap.w);
        hb[r][lane] = h;
    }
    __syncthreads();
    if (ox >= dw) return;
#pragma unroll
    for (int i = 0; i < PD_TY / 4; ++i) {
        int ty = wv + 4 * i, oy = oy0 + ty;
        if (oy >= dh) break;
        Px<M> r0 = hb[2 * ty][lane], r1 = hb[2 * ty + 1][lane], r2 = hb[2 * ty + 2][lane], r3 = hb[2 * ty + 3][lane], r4 = hb[2 * ty + 4][lane];
        Px<M> o;
        WT a0 = tap5<WT>(r2.c0, r1.c0, r3.c0, r0.c0, r4.c0);
        WT a1 = tap5<WT>(r2.c1, r1.c1, r3.c1, r0.c1, r4.c1);
        WT a2 = tap5<WT>(r2.c2, r1.c2, r3.c2, r0.c2, r4.c2);
        float aw = tap5<float>(r2.w, r1.w, r3.w, r0.w, r4.w);
        if constexpr (M == M_I16) { o.c0 = sat_s16((a0 + 128) >> 8); o.c1 = sat_s16((a1 + 128) >> 8); o.c2 = sat_s16((a2 + 128) >> 8); }
        else { o.c0 = a0 * (1.f / 256.f); o.c1 = a1 * (1.f / 256.f); o.c2 = a2 * (1.f / 256.f); }
        o.w = aw * (1.f / 256.f);
        store_px<M, false>(dst, ox, oy, o);
    }
}

// ------------------------------------------------------------------------------------------------
// pyrUp of a coarse tile held in LDS.  Thread (lane, wv) owns coarse pixel (cx, cy) and produces
// the 2x2 fine block.  ct rows are coarse rows cy0-1 .. cy0+4 through the row map
// (row -1 := row 1, row h := row h-1), columns cx0-1 .. cx0+64 (clamped; edge columns use
// OpenCV's explicit edge formulas so the clamped value is never used).
// ------------------------------------------------------------------------------------------------
constexpr int UP_TY = 4;

template <int M>
struct Up4 { typename WorkT<M>::t v[2][2][3]; };  // [dy][dx][channel]

template <int M>
__device__ __forceinline__ int up_row_map(int y, int h) {
    // borderInterpolate(2*y, 2*h, REFLECT_101) / 2
    return reflect101(2 * y, 2 * h) / 2;
}

template <int M, bool DST>
__device__ __forceinline__ void stage_coarse(Px<M> (*ct)[WAVE + 2], const LevelBuf& coarse, int cx0, int cy0) {
    for (int i = threadIdx.x; i < (UP_TY + 2) * (WAVE + 2); i += 256) {
        int ry = i / (WAVE + 2), rx = i - ry * (WAVE + 2);
        int gy = up_row_map<M>(cy0 - 1 + ry, coarse.rows);
        int gx = min(max(cx0 - 1 + rx, 0), coarse.cols - 1);
        ct[ry][rx] = load_px<M, DST>(coarse, gx, gy);
    }
}

template <int M>
__device__ __forceinline__ Up4<M> pyr_up_2x2(Px<M> (*ct)[WAVE + 2], int lane, int wv, int cx, int cw) {
    using WT = typename WorkT<M>::t;
    WT t0[3][3], t1[3][3];  // [row][channel]
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        Px<M> sm = ct[wv + rr][lane], sc = ct[wv + rr][lane + 1], sp = ct[wv + rr][lane + 2];
        WT m[3] = {sm.c0, sm.c1, sm.c2}, c[3] = {sc.c0, sc.c1, sc.c2}, p[3] = {sp.c0, sp.c1, sp.c2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (cw == 1) { t0[rr][k] = c[k] * 8; t1[rr][k] = c[k] * 8; }
            else if (cx == 0) { t0[rr][k] = c[k] * 6 + p[k] * 2; t1[rr][k] = (c[k] + p[k]) * 4; }
            else if (cx == cw - 1) { t0[rr][k] = m[k] + c[k] * 7; t1[rr][k] = c[k] * 8; }
            else { t0[rr][k] = m[k] + c[k] * 6 + p[k]; t1[rr][k] = (c[k] + p[k]) * 4; }
        }
    }
    Up4<M> u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        WT e0 = t0[0][k] + t0[1][k] * 6 + t0[2][k], e1 = t1[0][k] + t1[1][k] * 6 + t1[2][k];
        WT o0 = (t0[1][k] + t0[2][k]) * 4, o1 = (t1[1][k] + t1[2][k]) * 4;
        if constexpr (M == M_I16) {
            u.v[0][0][k] = sat_s16((e0 + 32) >> 6); u.v[0][1][k] = sat_s16((e1 + 32) >> 6);
            u.v[1][0][k] = sat_s16((o0 + 32) >> 6); u.v[1][1][k] = sat_s16((o1 + 32) >> 6);
        } else {
            u.v[0][0][k] = e0 * (1.f / 64.f); u.v[0][1][k] = e1 * (1.f / 64.f);
            u.v[1][0][k] = o0 * (1.f / 64.f); u.v[1][1][k] = o1 * (1.f / 64.f);
        }
    }
    return u;
}

// dst += cast(lap * w), dstW += w   (MultiBandBlender::feed accumulate loop)
template <int M>
__device__ __forceinline__ void accumulate(const LevelBuf& dst, int x, int y, typename WorkT<M>::t l0,
                                           typename WorkT<M>::t l1, typename WorkT<M>::t l2, float w) {
    Px<M> d = load_px<M, true>(dst, x, y);
    if constexpr (M == M_I16) {
        d.c0 = wrap_s16(d.c0 + f2s_x86((float)l0 * w));
        d.c1 = wrap_s16(d.c1 + f2s_x86((float)l1 * w));
        d.c2 = wrap_s16(d.c2 + f2s_x86((float)l2 * w));
    } else {
        d.c0 = d.c0 + l0 * w; d.c1 = d.c1 + l1 * w; d.c2 = d.c2 + l2 * w;
    }
    d.w = d.w + w;
    store_px<M, true>(dst, x, y, d);
}

template <int M, int SK>
__global__ __launch_bounds__(256) void k_lap_acc(Src0 s0, LevelBuf fine, LevelBuf coarse, LevelBuf dst, int x_tl, int y_tl) {
    __shared__ Px<M> ct[UP_TY + 2][WAVE + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cx0 = blockIdx.x * WAVE, cy0 = blockIdx.y * UP_TY;
    stage_coarse<M, false>(ct, coarse, cx0, cy0);
    __syncthreads();
    const int cx = cx0 + lane, cy = cy0 + wv;
    if (cx >= coarse.cols || cy >= coarse.rows) return;
    Up4<M> u = pyr_up_2x2<M>(ct, lane, wv, cx, coarse.cols);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            int fx = 2 * cx + dx, fy = 2 * cy + dy;
            Px<M> g = load_any<M, SK>(s0, fine, fx, fy);
            typename WorkT<M>::t l0, l1, l2;
            if constexpr (M == M_I16) {  // cv::subtract saturates
                l0 = sat_s16(g.c0 - u.v[dy][dx][0]); l1 = sat_s16(g.c1 - u.v[dy][dx][1]); l2 = sat_s16(g.c2 - u.v[dy][dx][2]);
            } else {
                l0 = g.c0 - u.v[dy][dx][0]; l1 = g.c1 - u.v[dy][dx][1]; l2 = g.c2 - u.v[dy][dx][2];
            }
            accumulate<M>(dst, x_tl + fx, y_tl + fy, l0, l1, l2, g.w);
        }
}

// top level: the Laplacian pyramid's last level is the Gaussian level itself
template <int M, int SK>
__global__ __launch_bounds__(256) void k_top_acc(Src0 s0, LevelBuf top, LevelBuf dst, int x_tl, int y_tl, int rows, int cols) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    Px<M> g = load_any<M, SK>(s0, top, x, y);
    accumulate<M>(dst, x_tl + x, y_tl + y, g.c0, g.c1, g.c2, g.w);
}

// normalizeUsingWeightMap for one pixel
template <int M>
__device__ __forceinline__ void normalise(Px<M>& d) {
    float den = d.w + WEIGHT_EPS;
    if constexpr (M == M_I16) {
        d.c0 = f2s_x86((float)d.c0 / den); d.c1 = f2s_x86((float)d.c1 / den); d.c2 = f2s_x86((float)d.c2 / den);
    } else {
        d.c0 = d.c0 / den; d.c1 = d.c1 / den; d.c2 = d.c2 / den;
    }
}

struct OutMat {  // the caller's blend() outputs
    unsigned char* img; size_t img_step; int img_f32;
    unsigned char* mask; size_t mask_step;
    int rows, cols;  // dst_roi_final_ size
};

template <int M>
__device__ __forceinline__ void write_final(const OutMat& o, int x, int y, const Px<M>& d) {
    if (x >= o.cols || y >= o.rows) return;      // crop to dst_roi_final_
    bool on = d.w > WEIGHT_EPS;                  // compare(w0, WEIGHT_EPS, CMP_GT)
    if (o.mask) o.mask[(size_t)y * o.mask_step + x] = on ? 255 : 0;
    if (o.img_f32) {
        float* q = (float*)(o.img + (size_t)y * o.img_step) + (size_t)x * 3;
        q[0] = on ? (float)d.c0 : 0.f; q[1] = on ? (float)d.c1 : 0.f; q[2] = on ? (float)d.c2 : 0.f;
    } else {
        short* q = (short*)(o.img + (size_t)y * o.img_step) + (size_t)x * 3;
        if constexpr (M == M_I16) {
            q[0] = on ? (short)d.c0 : 0; q[1] = on ? (short)d.c1 : 0; q[2] = on ? (short)d.c2 : 0;
        } else {  // saturate_cast<short>(float)
            q[0] = on ? (short)sat_s16(cvround_x86(d.c0)) : 0;
            q[1] = on ? (short)sat_s16(cvround_x86(d.c1)) : 0;
            q[2] = on ? (short)sat_s16(cvround_x86(d.c2)) : 0;
        }
    }
}

// top level of blend(): normalise in place (or straight to the caller's mat when num_bands == 0)
template <int M, bool FINAL>
__global__ __launch_bounds__(256) void k_norm_top(LevelBuf lv, OutMat out) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= lv.cols || y >= lv.rows) return;
    Px<M> d = load_px<M, true>(lv, x, y);
    normalise<M>(d);
    if constexpr (FINAL) write_final<M>(out, x, y, d);
    else store_px<M, true>(lv, x, y, d);
}

// out_{k-1} = sat(pyrUp(out_k) + normalise(dst_{k-1}))   (restoreImageFromLaplacePyr, fused normalise)
template <int M, bool FINAL>
__global__ __launch_bounds__(256) void k_collapse(LevelBuf coarse, LevelBuf fine, OutMat out) {
    __shared__ Px<M> ct[UP_TY + 2][WAVE + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cx0 = blockIdx.x * WAVE, cy0 = blockIdx.y * UP_TY;
    stage_coarse<M, true>(ct, coarse, cx0, cy0);
    __syncthreads();
    const int cx = cx0 + lane, cy = cy0 + wv;
    if (cx >= coarse.cols || cy >= coarse.rows) return;
    Up4<M> u = pyr_up_2x2<M>(ct, lane, wv, cx, coarse.cols);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            int fx = 2 * cx + dx, fy = 2 * cy + dy;
            if constexpr (FINAL) { if (fx >= out.cols || fy >= out.rows) continue; }
            Px<M> d = load_px<M, true>(fine, fx, fy);
            normalise<M>(d);
            if constexpr (M == M_I16) {  // cv::add saturates
                d.c0 = sat_s16(u.v[dy][dx][0] + d.c0); d.c1 = sat_s16(u.v[dy][dx][1] + d.c1); d.c2 = sat_s16(u.v[dy][dx][2] + d.c2);
            } else {
                d.c0 = u.v[dy][dx][0] + d.c0; d.c1 = u.v[dy][dx][1] + d.c1; d.c2 = u.v[dy][dx][2] + d.c2;
            }
            if constexpr (FINAL) write_final<M>(out, fx, fy, d);
            else store_px<M, true>(fine, fx, fy, d);
        }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
constexpr int MAX_LEVELS = 24;

size_t g_px_bytes(int prec) { return prec == M_F32 ? 16 : 8; }           // tile Gaussian record
size_t d_px_bytes(int prec) { return prec == M_I16 ? 8 : 16; }           // destination record
// algorithmic bytes (SURVEY §8(d) model): image part and weight part of a record
double alg_g(int prec) { return prec == M_F32 ? 16.0 : (prec == M_I16 ? 10.0 : 8.0); }
double alg_g_rgb(int prec) { return prec == M_F32 ? 12.0 : 6.0; }
double alg_d(int prec) { return prec == M_I16 ? 10.0 : 16.0; }
double alg_d_rgb(int prec) { return prec == M_I16 ? 6.0 : 12.0; }

}  // namespace

struct isx_blender {
    int device = 0;
    hipStream_t stream = nullptr;
    int type = ISX_BLEND_MULTI_BAND;
    int actual_num_bands = 5, num_bands = 5, prec = ISX_PREC_I16;
    bool prepared = false;
    int rx = 0, ry = 0, rw = 0, rh = 0;  // dst_roi_ (padded)
    int fw = 0, fh = 0;                  // dst_roi_final_ size
    LevelBuf dst[MAX_LEVELS];
    DevBuf dst_arena, tile_arena;
    MatStage st_img, st_mask, st_out, st_outmask;
    std::vector<unsigned char> host_tmp;
};

namespace {

int layout_levels(LevelBuf* lv, int L, int rows, int cols, int prec, bool is_dst, char* base, size_t* total) {
    size_t off = 0;
    for (int i = 0; i <= L; ++i) {
        lv[i].rows = rows; lv[i].cols = cols;
        size_t n = (size_t)rows * cols;
        size_t ib = n * (is_dst ? d_px_bytes(prec) : g_px_bytes(prec));
        lv[i].img = base ? base + off : nullptr;
        off += (ib + 255) & ~(size_t)255;
        lv[i].wgt = nullptr;
        if (prec == M_I16) {
            lv[i].wgt = base ? (float*)(base + off) : nullptr;
            off += (n * 4 + 255) & ~(size_t)255;
        }
        rows = (rows + 1) / 2; cols = (cols + 1) / 2;
    }
    *total = off;
    return ISX_OK;
}

int src_kind_of(int type) { return type == ISX_8UC3 ? SK_U8 : (type == ISX_16SC3 ? SK_S16 : SK_F32); }
double src_px_bytes(int sk) { return sk == SK_U8 ? 3.0 : (sk == SK_S16 ? 6.0 : 12.0); }

template <int M, int SK>
int run_feed(isx_blender* b, const Src0& s0, LevelBuf* g, int L, int x_tl, int y_tl) {
    hipStream_t st = b->stream;
    const int prec = M;
    // Gaussian chain (image + weight): G_{k+1} = pyrDown(G_k)
    for (int k = 0; k < L; ++k) {
        dim3 grid(cdiv(g[k + 1].cols, WAVE), cdiv(g[k + 1].rows, PD_TY));
        double in_px = (double)g[k].rows * g[k].cols, out_px = (double)g[k + 1].rows * g[k + 1].cols;
        double bytes = in_px * (k == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec)) + out_px * alg_g(prec);
        if (k == 0) ISX_LAUNCH("pyr_down_l0", bytes, st, (k_pyr_down<M, SK>), grid, dim3(256), 0, s0, g[0], g[1]);
        else ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down<M, SK_LEVEL>), grid, dim3(256), 0, s0, g[k], g[k + 1]);
    }
    // Laplacian + weighted accumulate per level
    int xt = x_tl, yt = y_tl;
    for (int k = 0; k < L; ++k) {
        dim3 grid(cdiv(g[k + 1].cols, WAVE), cdiv(g[k + 1].rows, UP_TY));
        double fine_px = (double)g[k].rows * g[k].cols, coarse_px = (double)g[k + 1].rows * g[k + 1].cols;
        double bytes = fine_px * ((k == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec)) + 2.0 * alg_d(prec)) + coarse_px * alg_g_rgb(prec);
        if (k == 0) ISX_LAUNCH("lap_acc_l0", bytes, st, (k_lap_acc<M, SK>), grid, dim3(256), 0, s0, g[0], g[1], b->dst[0], xt, yt);
        else ISX_LAUNCH("lap_acc", bytes, st, (k_lap_acc<M, SK_LEVEL>), grid, dim3(256), 0, s0, g[k], g[k + 1], b->dst[k], xt, yt);
        xt /= 2; yt /= 2;
    }
    {
        dim3 grid(cdiv(g[L].cols, 64), cdiv(g[L].rows, 4));
        double px = (double)g[L].rows * g[L].cols;
        double bytes = px * ((L == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec)) + 2.0 * alg_d(prec));
        if (L == 0) ISX_LAUNCH("top_acc", bytes, st, (k_top_acc<M, SK>), grid, dim3(256), 0, s0, g[0], b->dst[0], xt, yt, g[0].rows, g[0].cols);
        else ISX_LAUNCH("top_acc", bytes, st, (k_top_acc<M, SK_LEVEL>), grid, dim3(256), 0, s0, g[L], b->dst[L], xt, yt, g[L].rows, g[L].cols);
    }
    return ISX_OK;
}

template <int M>
int run_feed_kind(isx_blender* b, int sk, const Src0& s0, LevelBuf* g, int L, int x_tl, int y_tl) {
    switch (sk) {
        case SK_U8: return run_feed<M, SK_U8>(b, s0, g, L, x_tl, y_tl);
        case SK_S16: return run_feed<M, SK_S16>(b, s0, g, L, x_tl, y_tl);
        default: return run_feed<M, SK_F32>(b, s0, g, L, x_tl, y_tl);
    }
}

template <int M>
int run_blend(isx_blender* b, const OutMat& out) {
    hipStream_t st = b->stream;
    const int L = b->num_bands, prec = M;
    LevelBuf* d = b->dst;
    {
        dim3 grid(cdiv(d[L].cols, 64), cdiv(d[L].rows, 4));
        double px = (double)d[L].rows * d[L].cols;
        if (L == 0) ISX_LAUNCH("norm_top_final", px * alg_d(prec) + (double)out.rows * out.cols * (out.img_f32 ? 13.0 : 7.0), st, (k_norm_top<M, true>), grid, dim3(256), 0, d[L], out);
        else ISX_LAUNCH("norm_top", px * (alg_d(prec) + alg_d_rgb(prec)), st, (k_norm_top<M, false>), grid, dim3(256), 0, d[L], out);
    }
    for (int k = L; k >= 1; --k) {
        dim3 grid(cdiv(d[k].cols, WAVE), cdiv(d[k].rows, UP_TY));
        double coarse_px = (double)d[k].rows * d[k].cols, fine_px = (double)d[k - 1].rows * d[k - 1].cols;
        if (k == 1) {
            double bytes = coarse_px * alg_d_rgb(prec) + (double)out.rows * out.cols * (alg_d(prec) + (out.img_f32 ? 13.0 : 7.0));
            ISX_LAUNCH("collapse_final", bytes, st, (k_collapse<M, true>), grid, dim3(256), 0, d[1], d[0], out);
        } else {
            double bytes = coarse_px * alg_d_rgb(prec) + fine_px * (alg_d(prec) + alg_d_rgb(prec));
            ISX_LAUNCH("collapse", bytes, st, (k_collapse<M, false>), grid, dim3(256), 0, d[k], d[k - 1], out);
        }
    }
    return ISX_OK;
}

int do_prepare(isx_blender* b, int x, int y, int width, int height) {
    ISX_CHECK_ARG(width > 0 && height > 0, ISX_ERR_INVALID, "prepare: empty destination ROI %d x %d", width, height);
    ISX_HIP(hipSetDevice(b->device));
    b->fw = width; b->fh = height;
    // num_bands_ = min(actual_num_bands_, (int)ceil(log(max_len) / log(2.0)))
    double max_len = (double)(width > height ? width : height);
    int cl = (int)std::ceil(std::log(max_len) / std::log(2.0));
    b->num_bands = b->actual_num_bands < cl ? b->actual_num_bands : cl;
    int L = b->num_bands, m = 1 << L;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    b->rx = x; b->ry = y; b->rw = width; b->rh = height;
    size_t total = 0;
    layout_levels(b->dst, L, height, width, b->prec, true, nullptr, &total);
    ISX_TRY(b->dst_arena.reserve(total));
    layout_levels(b->dst, L, height, width, b->prec, true, (char*)b->dst_arena.p, &total);
    ISX_HIP(hipMemsetAsync(b->dst_arena.p, 0, total, b->stream));   // dst_.setTo(0), weights setTo(0)
    b->prepared = true;
    return ISX_OK;
}

int do_feed(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y, bool u8_entry) {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "feed: null blender");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "feed: prepare() has not been called (or blend() already released the pyramids)");
    ISX_TRY(check_mat(img, "feed: img"));
    ISX_TRY(check_mat(mask, "feed: mask"));
    if (u8_entry) ISX_CHECK_ARG(img->type == ISX_8UC3, ISX_ERR_TYPE, "feed_u8: img must be CV_8UC3, got %s", type_name(img->type));
    else {
        ISX_CHECK_ARG(img->type != ISX_8UC3, ISX_ERR_UNSUPPORTED,
                      "feed: CV_8UC3 input selects OpenCV's 8-bit pyramid, which the reference never uses; convert to CV_16SC3 (W:294) or call isx_blender_feed_u8");
        ISX_CHECK_ARG(img->type == ISX_16SC3 || (img->type == ISX_32FC3 && b->prec != ISX_PREC_I16), ISX_ERR_TYPE,
                      "feed: img must be CV_16SC3%s, got %s", b->prec != ISX_PREC_I16 ? " or CV_32FC3" : "", type_name(img->type));
    }
    ISX_CHECK_ARG(mask->type == ISX_8UC1, ISX_ERR_TYPE, "feed: mask must be CV_8U, got %s", type_name(mask->type));
    ISX_CHECK_ARG(mask->rows == img->rows && mask->cols == img->cols, ISX_ERR_SIZE, "feed: mask %dx%d does not match img %dx%d",
                  mask->cols, mask->rows, img->cols, img->rows);
    ISX_HIP(hipSetDevice(b->device));
    ISX_TRY(b->st_img.use_in(img, b->stream, "feed: img"));
    ISX_TRY(b->st_mask.use_in(mask, b->stream, "feed: mask"));
    const isx_mat& di = b->st_img.d;
    const isx_mat& dm = b->st_mask.d;

    // geometry of MultiBandBlender::feed
    const int L = b->num_bands, m = 1 << L, gap = 3 * m;
    const int brx_d = b->rx + b->rw, bry_d = b->ry + b->rh;
    int tlnx = std::max(b->rx, tl_x - gap), tlny = std::max(b->ry, tl_y - gap);
    int brnx = std::min(brx_d, tl_x + img->cols + gap), brny = std::min(bry_d, tl_y + img->rows + gap);
    tlnx = b->rx + (((tlnx - b->rx) >> L) << L);
    tlny = b->ry + (((tlny - b->ry) >> L) << L);
    int width = brnx - tlnx, height = brny - tlny;
    ISX_CHECK_ARG(width > 0 && height > 0, ISX_ERR_INVALID, "feed: tile at (%d,%d) %dx%d lies outside the prepared ROI", tl_x, tl_y, img->cols, img->rows);
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    brnx = tlnx + width; brny = tlny + height;
    int dy = std::max(brny - bry_d, 0), dx = std::max(brnx - brx_d, 0);
    tlnx -= dx; brnx -= dx; tlny -= dy; brny -= dy;
    ISX_CHECK_ARG(tlnx >= b->rx && tlny >= b->ry, ISX_ERR_INVALID, "feed: padded tile does not fit the prepared ROI");

    Src0 s0;
    s0.img = (const unsigned char*)di.data; s0.img_step = di.step;
    s0.mask = (const unsigned char*)dm.data; s0.mask_step = dm.step;
    s0.rows = img->rows; s0.cols = img->cols;
    s0.top = tl_y - tlny; s0.left = tl_x - tlnx;
    s0.height = height; s0.width = width;

    LevelBuf g[MAX_LEVELS];
    size_t total = 0;
    // tile Gaussian levels 1..L live in the tile arena; level 0 is the caller's tile
    layout_levels(g, L, height, width, b->prec, false, nullptr, &total);
    size_t skip = 0;
    {   // do not allocate level 0
        size_t n0 = (size_t)height * width;
        skip = (n0 * g_px_bytes(b->prec) + 255) & ~(size_t)255;
        if (b->prec == M_I16) skip += (n0 * 4 + 255) & ~(size_t)255;
    }
    ISX_TRY(b->tile_arena.reserve(total - skip + 256));
    layout_levels(g, L, height, width, b->prec, false, (char*)b->tile_arena.p - skip, &total);
    g[0].img = nullptr; g[0].wgt = nullptr;

    int x_tl = tlnx - b->rx, y_tl = tlny - b->ry;
    int sk = src_kind_of(img->type);
    switch (b->prec) {
        case M_I16: return run_feed_kind<M_I16>(b, sk, s0, g, L, x_tl, y_tl);
        case M_F32: return run_feed_kind<M_F32>(b, sk, s0, g, L, x_tl, y_tl);
        default: return run_feed_kind<M_F16>(b, sk, s0, g, L, x_tl, y_tl);
    }
}

}  // namespace

extern "C" {

int isx_blender_create(int type, int num_bands, int precision, int device, isx_blender** out) {
    clear_error();
    ISX_CHECK_ARG(out != nullptr, ISX_ERR_INVALID, "isx_blender_create: null out pointer");
    *out = nullptr;
    ISX_CHECK_ARG(type == ISX_BLEND_MULTI_BAND, ISX_ERR_UNSUPPORTED, "isx_blender_create: only Blender::MULTI_BAND (2) is implemented, got %d", type);
    ISX_CHECK_ARG(num_bands >= 0 && num_bands < MAX_LEVELS - 1, ISX_ERR_INVALID, "isx_blender_create: num_bands %d out of range", num_bands);
    ISX_CHECK_ARG(precision >= ISX_PREC_I16 && precision <= ISX_PREC_F16ACC32, ISX_ERR_INVALID, "isx_blender_create: bad precision %d", precision);
    int n = 0;
    ISX_HIP(hipGetDeviceCount(&n));
    ISX_CHECK_ARG(device >= 0 && device < n, ISX_ERR_INVALID, "isx_blender_create: device %d of %d", device, n);
    isx_blender* b = new (std::nothrow) isx_blender();
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_NOMEM, "isx_blender_create: out of host memory");
    b->device = device; b->type = type; b->actual_num_bands = num_bands; b->num_bands = num_bands; b->prec = precision;
    *out = b;
    return ISX_OK;
}

int isx_blender_destroy(isx_blender* b) {
    if (!b) return ISX_OK;
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->stream);
    delete b;
    return ISX_OK;
}

int isx_blender_set_stream(isx_blender* b, void* hip_stream) {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_stream: null blender");
    b->stream = (hipStream_t)hip_stream;
    return ISX_OK;
}

int isx_blender_set_num_bands(isx_blender* b, int num_bands) {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_num_bands: null blender");
    ISX_CHECK_ARG(num_bands >= 0 && num_bands < MAX_LEVELS - 1, ISX_ERR_INVALID, "setNumBands(%d) out of range", num_bands);
    b->actual_num_bands = num_bands;
    return ISX_OK;
}

int isx_blender_num_bands(isx_blender* b, int* num_bands) {
    ISX_CHECK_ARG(b != nullptr && num_bands != nullptr, ISX_ERR_INVALID, "isx_blender_num_bands: null argument");
    *num_bands = b->prepared ? b->num_bands : b->actual_num_bands;
    return ISX_OK;
}

int isx_blender_prepare(isx_blender* b, int n, const int* c, const int* s) {
    clear_error();
    ISX_CHECK_ARG(b != nullptr && c != nullptr && s != nullptr, ISX_ERR_INVALID, "prepare: null argument");
    ISX_CHECK_ARG(n > 0, ISX_ERR_INVALID, "prepare: no tiles");
    // resultRoi(corners, sizes)
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {
        ISX_CHECK_ARG(s[2 * i] > 0 && s[2 * i + 1] > 0, ISX_ERR_INVALID, "prepare: tile %d has empty size", i);
        tlx = std::min(tlx, c[2 * i]); tly = std::min(tly, c[2 * i + 1]);
        brx = std::max(brx, c[2 * i] + s[2 * i]); bry = std::max(bry, c[2 * i + 1] + s[2 * i + 1]);
    }
    return do_prepare(b, tlx, tly, brx - tlx, bry - tly);
}

int isx_blender_prepare_roi(isx_blender* b, int x, int y, int width, int height) {
    clear_error();
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "prepare: null blender");
    return do_prepare(b, x, y, width, height);
}

int isx_blender_feed(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y) {
    clear_error();
    return do_feed(b, img, mask, tl_x, tl_y, false);
}

int isx_blender_feed_u8(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y) {
    clear_error();
    return do_feed(b, img, mask, tl_x, tl_y, true);
}

int isx_blender_result_size(isx_blender* b, int* width, int* height) {
    ISX_CHECK_ARG(b != nullptr && width != nullptr && height != nullptr, ISX_ERR_INVALID, "result_size: null argument");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "result_size: prepare() has not been called");
    *width = b->fw; *height = b->fh;
    return ISX_OK;
}

int isx_blender_debug_level(isx_blender* b, int level, void* lap, float* weight, int* rows, int* cols) {
    clear_error();
    ISX_CHECK_ARG(b != nullptr && rows != nullptr && cols != nullptr, ISX_ERR_INVALID, "debug_level: null argument");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "debug_level: prepare() has not been called");
    ISX_CHECK_ARG(level >= 0 && level <= b->num_bands, ISX_ERR_INVALID, "debug_level: level %d of %d", level, b->num_bands);
    ISX_HIP(hipSetDevice(b->device));
    const LevelBuf& d = b->dst[level];
    *rows = d.rows; *cols = d.cols;
    size_t n = (size_t)d.rows * d.cols;
    ISX_HIP(hipStreamSynchronize(b->stream));
    if (b->prec == M_I16) {
        if (lap) {
            b->host_tmp.resize(n * 8);
            ISX_HIP(hipMemcpy(b->host_tmp.data(), d.img, n * 8, hipMemcpyDeviceToHost));
            const short* s = (const short*)b->host_tmp.data();
            short* o = (short*)lap;
            for (size_t i = 0; i < n; ++i) { o[3 * i] = s[4 * i]; o[3 * i + 1] = s[4 * i + 1]; o[3 * i + 2] = s[4 * i + 2]; }
        }
        if (weight) ISX_HIP(hipMemcpy(weight, d.wgt, n * 4, hipMemcpyDeviceToHost));
    } else {
        b->host_tmp.resize(n * 16);
        ISX_HIP(hipMemcpy(b->host_tmp.data(), d.img, n * 16, hipMemcpyDeviceToHost));
        const float* s = (const float*)b->host_tmp.data();
        float* o = (float*)lap;
        for (size_t i = 0; i < n; ++i) {
            if (o) { o[3 * i] = s[4 * i]; o[3 * i + 1] = s[4 * i + 1]; o[3 * i + 2] = s[4 * i + 2]; }
            if (weight) weight[i] = s[4 * i + 3];
        }
    }
    return ISX_OK;
}

int isx_blender_blend(isx_blender* b, isx_mat* dst, isx_mat* dst_mask) {
    clear_error();
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "blend: null blender");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "blend: prepare() has not been called (or blend() already released the pyramids)");
    ISX_TRY(check_mat(dst, "blend: dst"));
    ISX_CHECK_ARG(dst->type == ISX_16SC3 || (dst->type == ISX_32FC3 && b->prec != ISX_PREC_I16), ISX_ERR_TYPE,
                  "blend: dst must be CV_16SC3%s, got %s", b->prec != ISX_PREC_I16 ? " or CV_32FC3" : "", type_name(dst->type));
    ISX_CHECK_ARG(dst->rows == b->fh && dst->cols == b->fw, ISX_ERR_SIZE, "blend: dst is %dx%d, result is %dx%d", dst->cols, dst->rows, b->fw, b->fh);
    if (dst_mask) {
        ISX_TRY(check_mat(dst_mask, "blend: dst_mask"));
        ISX_CHECK_ARG(dst_mask->type == ISX_8UC1, ISX_ERR_TYPE, "blend: dst_mask must be CV_8U, got %s", type_name(dst_mask->type));
        ISX_CHECK_ARG(dst_mask->rows == b->fh && dst_mask->cols == b->fw, ISX_ERR_SIZE, "blend: dst_mask is %dx%d, result is %dx%d",
                      dst_mask->cols, dst_mask->rows, b->fw, b->fh);
    }
    ISX_HIP(hipSetDevice(b->device));
    ISX_TRY(b->st_out.use_out(dst, b->stream, "blend: dst"));
    if (dst_mask) ISX_TRY(b->st_outmask.use_out(dst_mask, b->stream, "blend: dst_mask"));
    OutMat o;
    o.img = (unsigned char*)b->st_out.d.data; o.img_step = b->st_out.d.step; o.img_f32 = dst->type == ISX_32FC3;
    o.mask = dst_mask ? (unsigned char*)b->st_outmask.d.data : nullptr;
    o.mask_step = dst_mask ? b->st_outmask.d.step : 0;
    o.rows = b->fh; o.cols = b->fw;
    int rc;
    switch (b->prec) {
        case M_I16: rc = run_blend<M_I16>(b, o); break;
        case M_F32: rc = run_blend<M_F32>(b, o); break;
        default: rc = run_blend<M_F16>(b, o); break;
    }
    ISX_TRY(rc);
    ISX_TRY(b->st_out.finish_out(b->stream));
    if (dst_mask) ISX_TRY(b->st_outmask.finish_out(b->stream));
    b->prepared = false;   // dst_pyr_laplace_.clear(); dst_band_weights_.clear()
    return ISX_OK;
}

}  // extern "C"
