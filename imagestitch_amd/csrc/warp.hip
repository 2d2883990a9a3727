// warp.hip — cylindrical / spherical rotation warper for MI355X (gfx950).
// Replaces the reference's in-tree warper (W:30-161: mapForward, mapBackward, detectResultRoi,
// setCameraParams, buildMaps, warp) and the cv::remap call it ends in (W:157; arithmetic spec:
// SURVEY.md §8(a) A8).  buildMaps + remap are fused: the back-projection is evaluated per
// destination pixel in registers and the maps are never written to HBM (isx_warper_build_maps +
// isx_remap exist for callers that keep the maps of a fixed rig).
//
// Transcendentals: mapBackward's sinf/cosf depend only on the destination COLUMN (u) and, for the
// spherical projector, on the destination ROW (v), so the host evaluates them once per column /
// row with the same libm the reference code would call and the kernels read them from two small
// tables — the per-pixel work is the 3x3 transform, two IEEE divisions and the sampler, all of
// which the GPU reproduces bit-for-bit (no FMA contraction, correctly rounded div / sqrt).
#include "isx_device.hpp"
#include "isx_internal.hpp"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <cstring>
#include <new>

using namespace isx;
using namespace isxd;

// roihost.cpp (plain C++, AVX2 where the CPU has it): detectResultRoi's border pixels ranked on the caller's thread
extern "C" int isx_roi_border_host(const float r_kinv[9], int spherical, int sw, int sh, int* cand_xy, int cap, float* scratch, int isa);

namespace {

constexpr float PI_F = (float)3.1415926535897932384626433832795;

struct Proj {
    float r_kinv[9], k_rinv[9];
    float scale;
    int kind;
};

struct MapTabs {            // device tables of the separable part of mapBackward
    const float* col_s;     // sinf(u / scale)            per destination column
    const float* col_c;     // cosf(u / scale)
    const float* row_a;     // cyl: v / scale             sph: sinf(pi - v / scale)
    const float* row_b;     // cyl: unused                sph: cosf(pi - v / scale)
};

// mapBackward (W:46-63) with the transcendental part tabulated
__device__ __forceinline__ void map_backward(const Proj& p, const MapTabs& t, int dx, int dy, float& x, float& y) {
    float x_, y_, z_;
    if (p.kind == ISX_WARP_CYLINDRICAL) {
        x_ = t.col_s[dx]; y_ = t.row_a[dy]; z_ = t.col_c[dx];             // W:51-53
    } else {
        float sinv = t.row_a[dy];
        x_ = sinv * t.col_s[dx]; y_ = t.row_b[dy]; z_ = sinv * t.col_c[dx];
    }
    float z;
    x = p.k_rinv[0] * x_ + p.k_rinv[1] * y_ + p.k_rinv[2] * z_;            // W:56
    y = p.k_rinv[3] * x_ + p.k_rinv[4] * y_ + p.k_rinv[5] * z_;            // W:57
    z = p.k_rinv[6] * x_ + p.k_rinv[7] * y_ + p.k_rinv[8] * z_;            // W:58
    if (z > 0) { x /= z; y /= z; }                                        // W:60
    else x = y = -1;                                                       // W:61
}

struct SrcView {
    const unsigned char* data;
    size_t step;
    int rows, cols;
};

__device__ __forceinline__ int clamp_short(int v) { return min(max(v, -32768), 32767); }

// cv::remap, INTER_NEAREST on a u8 / f32 image of CN channels
template <class T, int CN>
__device__ __forceinline__ void sample_nearest(const SrcView& s, float mx, float my, int border, T* out) {
    int sx = clamp_short(cvround_x86(mx)), sy = clamp_short(cvround_x86(my));
    if (!((unsigned)sx < (unsigned)s.cols && (unsigned)sy < (unsigned)s.rows)) {
        if (border == ISX_BORDER_CONSTANT) {
#pragma unroll
            for (int c = 0; c < CN; ++c) out[c] = 0;
            return;
        }
        sx = border_index(sx, s.cols, border); sy = border_index(sy, s.rows, border);
    }
    const T* q = (const T*)(s.data + (size_t)sy * s.step) + (size_t)sx * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) out[c] = q[c];
}

// cv::remap, INTER_LINEAR: coordinates quantised to 1/32 px (cvRound(x*32)), taps through
// borderInterpolate; u8 uses the 15-bit fixed-point table, f32 the float table.
template <class T, int CN>
__device__ __forceinline__ void sample_linear(const SrcView& s, float mx, float my, int border, T* out, bool ties_even = false) {
    int isx = cvround_x86(mx * 32.f), isy = cvround_x86(my * 32.f);
    int fx = isx & 31, fy = isy & 31;
    int sx = clamp_short(isx >> 5), sy = clamp_short(isy >> 5);
    if (border == ISX_BORDER_CONSTANT && (sx >= s.cols || sx + 1 < 0 || sy >= s.rows || sy + 1 < 0)) {
#pragma unroll
        for (int c = 0; c < CN; ++c) out[c] = 0;
        return;
    }
    int sx0 = border_index(sx, s.cols, border), sx1 = border_index(sx + 1, s.cols, border);
    int sy0 = border_index(sy, s.rows, border), sy1 = border_index(sy + 1, s.rows, border);
    const bool ok00 = sx0 >= 0 && sy0 >= 0, ok01 = sx1 >= 0 && sy0 >= 0, ok10 = sx0 >= 0 && sy1 >= 0, ok11 = sx1 >= 0 && sy1 >= 0;
    const T* r0 = (const T*)(s.data + (size_t)max(sy0, 0) * s.step);
    const T* r1 = (const T*)(s.data + (size_t)max(sy1, 0) * s.step);
    const int o0 = max(sx0, 0) * CN, o1 = max(sx1, 0) * CN;
    if constexpr (sizeof(T) == 1) {
        // BilinearTab_i: (32-fx)(32-fy)*32 ... ; entry (0,0) saturates to {32767,0,0,1}
        int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
        if ((fx | fy) == 0) { w0 = 32767; w3 = 1; }
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            int v0 = ok00 ? r0[o0 + c] : 0, v1 = ok01 ? r0[o1 + c] : 0, v2 = ok10 ? r1[o0 + c] : 0, v3 = ok11 ? r1[o1 + c] : 0;
            const int sum = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
            int r = (sum + (1 << 14)) >> 15;                               // FixedPtCast: round half up (OpenCV's CPU remap)
            // ISX_INTER_TIES_EVEN: the same sum as OpenCV's OpenCL (UMat) remap rounds it — half to even
            if (ties_even) r = (fx | fy) == 0 ? v0 : r - (((sum & 32767) == (1 << 14)) & (r & 1));
            out[c] = (T)sat_u8(r);
        }
    } else {
        float ax1 = fx * (1.f / 32.f), ax0 = 1.f - ax1, ay1 = fy * (1.f / 32.f), ay0 = 1.f - ay1;
        float w0 = ay0 * ax0, w1 = ay0 * ax1, w2 = ay1 * ax0, w3 = ay1 * ax1;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            float v0 = ok00 ? r0[o0 + c] : 0.f, v1 = ok01 ? r0[o1 + c] : 0.f, v2 = ok10 ? r1[o0 + c] : 0.f, v3 = ok11 ? r1[o1 + c] : 0.f;
            out[c] = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
        }
    }
}

// generic warp: one destination pixel per thread, 64 x 4 pixels per block
template <class T, int CN>
__global__ __launch_bounds__(256) void k_warp(Proj p, MapTabs t, SrcView src, unsigned char* dst, size_t dstep,
                                              int dw, int dh, int interp, int border) {
    int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    float mx, my;
    map_backward(p, t, dx, dy, mx, my);
    T o[CN];
    if (interp == ISX_INTER_NEAREST) sample_nearest<T, CN>(src, mx, my, border, o);
    else sample_linear<T, CN>(src, mx, my, border, o, (interp & ISX_INTER_TIES_EVEN) != 0);
    T* q = (T*)(dst + (size_t)dy * dstep) + (size_t)dx * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) q[c] = o[c];
}

// cv::remap with the caller's CV_32FC1 maps (W:157 as a function of its own: buildMaps once, remap per frame)
template <class T, int CN>
__global__ __launch_bounds__(256) void k_remap(SrcView src, const unsigned char* xmap, size_t xstep, const unsigned char* ymap, size_t ystep,
                                               unsigned char* dst, size_t dstep, int dw, int dh, int interp, int border) {
    int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const float mx = ((const float*)(xmap + (size_t)dy * xstep))[dx], my = ((const float*)(ymap + (size_t)dy * ystep))[dx];
    T o[CN];
    if (interp == ISX_INTER_NEAREST) sample_nearest<T, CN>(src, mx, my, border, o);
    else sample_linear<T, CN>(src, mx, my, border, o, (interp & ISX_INTER_TIES_EVEN) != 0);
    T* q = (T*)(dst + (size_t)dy * dstep) + (size_t)dx * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) q[c] = o[c];
}

// W:229 + W:232 fused: image LINEAR / REFLECT and mask NEAREST / CONSTANT from one map evaluation.
// OUT16: write the image as CV_16SC3 (the convertTo(CV_16S) of W:294 folded in; u8 -> s16 is exact).
//
// One thread produces FOUR consecutive destination pixels.  The bilinear footprint of a pixel is two
// runs of 6 contiguous bytes (p00|p01 and p10|p11); each run is fetched with ONE 12-byte load from the
// enclosing 4-byte-aligned address and realigned with v_alignbyte, instead of six byte loads
// (the byte-load version was bound by the texture-address unit: 18 memory instructions per pixel).
// Pixels whose taps touch the image border (reflected, hence not adjacent) or the last bytes of the
// buffer take the generic per-byte path.  When the destination rows are 4-byte aligned (VEC) the four
// pixels are stored as dwords.
// order-preserving float <-> uint key for atomicMin / atomicMax
__device__ __forceinline__ unsigned fkey(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float fkey_inv(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(b);
#else
    memcpy(&f, &b, 4);
#endif
    return f;
}

struct U3 { unsigned x, y, z; };

// border / buffer-end pixels of the fused kernel: generic per-byte sampler, kept out of line so that
// the hot path stays small.  Returns b | g << 8 | r << 16.
__device__ __noinline__ unsigned slow_bilinear_u8x3(const unsigned char* data, unsigned step, int rows, int cols, float mx, float my) {
    SrcView s{data, step, rows, cols};
    unsigned char o[3];
    sample_linear<unsigned char, 3>(s, mx, my, ISX_BORDER_REFLECT, o);
    return (unsigned)o[0] | ((unsigned)o[1] << 8) | ((unsigned)o[2] << 16);
}
__device__ __noinline__ unsigned slow_nearest_u8(const unsigned char* data, unsigned step, int rows, int cols, float mx, float my) {
    SrcView s{data, step, rows, cols};
    unsigned char m;
    sample_nearest<unsigned char, 1>(s, mx, my, ISX_BORDER_CONSTANT, &m);
    return m;
}

// (p00*w0 + p01*w1 + p10*w2 + p11*w3 + 2^14) >> 15 with 24-bit multiplies (bytes x 15-bit weights)
__device__ __forceinline__ unsigned fix15(unsigned a, unsigned b, unsigned c, unsigned d, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    return (__umul24(a, w0) + __umul24(b, w1) + __umul24(c, w2) + __umul24(d, w3) + (1u << 14)) >> 15;
}

template <bool OUT16, bool VEC>
__global__ __launch_bounds__(256) void k_warp_img_mask(Proj p, MapTabs t, SrcView img, SrcView msk, int has_mask,
                                                       unsigned char* dimg, size_t dimg_step, unsigned char* dmask,
                                                       size_t dmask_step, int dw, int dh,
                                                       const unsigned* roi_keys, int4 planned, int* mismatches) {
    // planned (sync-free) runs: the ROI scan that preceded this launch on the stream left its extrema in
    // roi_keys; one thread compares them with the ROI the caller planned (detectResultRoi's int casts)
    if (roi_keys != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        int tlx = f2i_x86(fkey_inv(roi_keys[0])), tly = f2i_x86(fkey_inv(roi_keys[1]));
        int brx = f2i_x86(fkey_inv(roi_keys[2])), bry = f2i_x86(fkey_inv(roi_keys[3]));
        if (tlx != planned.x || tly != planned.y || brx != planned.z || bry != planned.w) atomicAdd(mismatches, 1);
        // re-arm the keys for the next scan on this stream (saves two memset launches per tile)
        unsigned* k = const_cast<unsigned*>(roi_keys);
        k[0] = 0xffffffffu; k[1] = 0xffffffffu; k[2] = 0u; k[3] = 0u; k[4] = 0u;
    }
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx0 >= dw || dy >= dh) return;
    const int n = min(4, dw - dx0);
    // stage 1: the separable part of mapBackward for 4 columns (tables are padded to a multiple of 4)
    const float4 cs4 = *(const float4*)(t.col_s + dx0), cc4 = *(const float4*)(t.col_c + dx0);
    const float cs[4] = {cs4.x, cs4.y, cs4.z, cs4.w}, cc[4] = {cc4.x, cc4.y, cc4.z, cc4.w};
    const float ra = t.row_a[dy], rb = t.row_b[dy];
    // stage 2: maps, fixed-point coordinates and the two 12-byte windows of every pixel.
    // The host guarantees rows * step < 2^31, so byte offsets are 32-bit.
    const unsigned char* base = img.data;
    const unsigned step = (unsigned)img.step;
    const unsigned mis = (unsigned)((uintptr_t)base & 3);                     // misalignment of the base pointer
    const unsigned safe_end = (unsigned)(img.rows - 1) * step + (unsigned)img.cols * 3 + mis;
    const float hx = (float)img.cols - 0.5f, hy = (float)img.rows - 0.5f;
    const float hi_x = ((img.cols - 1) & 1) ? hx : __uint_as_float(__float_as_uint(hx) + 1u);
    const float hi_y = ((img.rows - 1) & 1) ? hy : __uint_as_float(__float_as_uint(hy) + 1u);
    float mx[4], my[4];
    unsigned wq[4];        // fx | fy << 8
    unsigned o0[4], o1[4]; // byte offsets of the two rows relative to the ALIGNED base (base - mis)
    bool fast[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x_, y_, z_;
        if (p.kind == ISX_WARP_CYLINDRICAL) { x_ = cs[k]; y_ = ra; z_ = cc[k]; }          // W:51-53
        else { x_ = ra * cs[k]; y_ = rb; z_ = ra * cc[k]; }
        float x = p.k_rinv[0] * x_ + p.k_rinv[1] * y_ + p.k_rinv[2] * z_;                  // W:56
        float y = p.k_rinv[3] * x_ + p.k_rinv[4] * y_ + p.k_rinv[5] * z_;                  // W:57
        float z = p.k_rinv[6] * x_ + p.k_rinv[7] * y_ + p.k_rinv[8] * z_;                  // W:58
        if (z > 0) { x /= z; y /= z; } else x = y = -1;                                   // W:60-61
        mx[k] = x; my[k] = y;
        const int isx = cvround_x86(x * 32.f), isy = cvround_x86(y * 32.f);
        wq[k] = (unsigned)(isx & 31) | ((unsigned)(isy & 31) << 8);
        const int sx = isx >> 5, sy = isy >> 5;   // no short saturation needed on the fast path (sx < cols <= 32767... checked below)
        fast[k] = k < n && (unsigned)sx < (unsigned)(img.cols - 1) && (unsigned)sy < (unsigned)(img.rows - 1);
        const unsigned a0 = __umul24((unsigned)sy, step) + __umul24((unsigned)sx, 3u) + mis;   // step < 2^24 (host); 24-bit multiplies are full-rate
        // the aligned 12-byte window of row 1 must end inside the buffer
        fast[k] = fast[k] && ((a0 + step) & ~3u) + 12 <= safe_end;
        o0[k] = fast[k] ? a0 : 0;
        o1[k] = fast[k] ? a0 + step : 0;
    }
    // stage 3: all eight loads in flight together
    const unsigned char* abase = base - mis;
    U3 v0[4], v1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v0[k] = *(const U3*)(abase + (o0[k] & ~3u));
        v1[k] = *(const U3*)(abase + (o1[k] & ~3u));
    }
    // stage 4: realign + fixed-point bilinear (BilinearTab_i weights, (sum + 2^14) >> 15)
    unsigned px[4];   // b | g << 8 | r << 16
    unsigned m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (fast[k]) {
            const unsigned s0 = o0[k] & 3, s1 = o1[k] & 3;
            const unsigned l0 = __builtin_amdgcn_alignbyte(v0[k].y, v0[k].x, s0), h0 = __builtin_amdgcn_alignbyte(v0[k].z, v0[k].y, s0);
            const unsigned l1 = __builtin_amdgcn_alignbyte(v1[k].y, v1[k].x, s1), h1 = __builtin_amdgcn_alignbyte(v1[k].z, v1[k].y, s1);
            const unsigned fx = wq[k] & 255, fy = wq[k] >> 8, gx = 32u - fx;
            // BilinearTab_i weights are 32 * wx * wy with wx in {32 - fx, fx}, wy in {32 - fy, fy} (entry (0,0) = {32767, 0, 0, 1}
            // gives p00 like everything else here), so (sum + 2^14) >> 15 == (S + 512) >> 10 with S = SUM p * wx * wy evaluated
            // separably in exact integers: the horizontal blends are byte dot products straight on the aligned windows
            // (l = b0 g0 r0 b1, h = g1 r1 . .), one v_dot4_u32_u8 for blue, two chained ones for green and red
            const unsigned wb = gx | (fx << 24), wgl = gx << 8, wgh = fx, wrl = gx << 16, wrh = fx << 8;
            const unsigned t0b = __builtin_amdgcn_udot4(l0, wb, 0u, false), t1b = __builtin_amdgcn_udot4(l1, wb, 0u, false);
            const unsigned t0g = __builtin_amdgcn_udot4(h0, wgh, __builtin_amdgcn_udot4(l0, wgl, 0u, false), false);
            const unsigned t1g = __builtin_amdgcn_udot4(h1, wgh, __builtin_amdgcn_udot4(l1, wgl, 0u, false), false);
            const unsigned t0r = __builtin_amdgcn_udot4(h0, wrh, __builtin_amdgcn_udot4(l0, wrl, 0u, false), false);
            const unsigned t1r = __builtin_amdgcn_udot4(h1, wrh, __builtin_amdgcn_udot4(l1, wrl, 0u, false), false);
            const unsigned gy = 32u - fy;
            const unsigned c0 = (__umul24(t0b, gy) + __umul24(t1b, fy) + 512u) >> 10;
            const unsigned c1 = (__umul24(t0g, gy) + __umul24(t1g, fy) + 512u) >> 10;
            const unsigned c2 = (__umul24(t0r, gy) + __umul24(t1r, fy) + 512u) >> 10;
            px[k] = c0 | (c1 << 8) | (c2 << 16);   // each c <= 255: the weights sum to 2^15
        } else if (k < n) {
            px[k] = slow_bilinear_u8x3(base, step, img.rows, img.cols, mx[k], my[k]);
        } else px[k] = 0;
        if (k < n) {
            if (has_mask) m[k] = slow_nearest_u8(msk.data, (unsigned)msk.step, msk.rows, msk.cols, mx[k], my[k]);
            else {  // masks[i].setTo(255) (W:213-214) warped NEAREST / CONSTANT: 255 iff cvRound(x) in [0, cols) and cvRound(y) in [0, rows).
                // Round-half-even makes that an interval test: x >= -0.5 (the tie -0.5 rounds to 0) and x < cols - 0.5, the
                // upper tie cols - 0.5 rounding to cols - 1 exactly when cols - 1 is even (hi_* is then one ulp larger).  NaN fails.
                m[k] = (mx[k] >= -0.5f && mx[k] < hi_x && my[k] >= -0.5f && my[k] < hi_y) ? 255u : 0u;
            }
        } else m[k] = 0;
    }
    // stage 5: stores
    if (VEC && n == 4) {
        if constexpr (OUT16) {
            unsigned* q = (unsigned*)(dimg + (__umul24((unsigned)dy, (unsigned)dimg_step) + (unsigned)dx0 * 6u));
#pragma unroll
            for (int k = 0; k < 4; k += 2) {   // 2 pixels = 6 shorts = 3 dwords
                const unsigned a = px[k], b2 = px[k + 1];
                q[3 * (k / 2)] = (a & 255) | (((a >> 8) & 255) << 16);
                q[3 * (k / 2) + 1] = ((a >> 16) & 255) | ((b2 & 255) << 16);
                q[3 * (k / 2) + 2] = ((b2 >> 8) & 255) | (((b2 >> 16) & 255) << 16);
            }
        } else {
            unsigned* q = (unsigned*)(dimg + (__umul24((unsigned)dy, (unsigned)dimg_step) + (unsigned)dx0 * 3u));
            q[0] = px[0] | (px[1] << 24);
            q[1] = (px[1] >> 8) | (px[2] << 16);
            q[2] = (px[2] >> 16) | (px[3] << 8);
        }
        *(unsigned*)(dmask + (__umul24((unsigned)dy, (unsigned)dmask_step) + (unsigned)dx0)) = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
    } else {
        for (int k = 0; k < n; ++k) {
            if constexpr (OUT16) {
                short* q = (short*)(dimg + (size_t)dy * dimg_step) + (size_t)(dx0 + k) * 3;
                q[0] = (short)(px[k] & 255); q[1] = (short)((px[k] >> 8) & 255); q[2] = (short)((px[k] >> 16) & 255);
            } else {
                unsigned char* q = dimg + (size_t)dy * dimg_step + (size_t)(dx0 + k) * 3;
                q[0] = (unsigned char)px[k]; q[1] = (unsigned char)(px[k] >> 8); q[2] = (unsigned char)(px[k] >> 16);
            }
            dmask[(size_t)dy * dmask_step + dx0 + k] = (unsigned char)m[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_warp_tile: the fused image + mask warp of a tile whose source mask is all 255 (W:213-214) - the hot kernel.
// Same outputs as k_warp_img_mask, about half its VALU instructions (that kernel is VALU-issue-bound, not HBM-bound):
//  * one thread = 4 consecutive columns x WR consecutive rows.  The separable parts of W:56-58 are hoisted: for the
//    cylindrical projector k_rinv[0,3,6] * sin(u) and k_rinv[2,5,8] * cos(u) depend on the column only (24 products, once
//    per thread), k_rinv[1,4,7] * (v / scale) on the row only (3 per row); per pixel the transform is its six additions.
//    Every product is rounded on its own in W:56-58 (no FMA), so hoisting them changes no bit.
//  * the arithmetic of two pixels rides in one packed instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 round each
//    half separately, like their scalar forms).
//  * x / z and y / z share z: one v_rcp_f32 per pixel and the hardware's own division recurrence written out with packed
//    FMAs - r1 = r0 + r0 (1 - z r0); q0 = a r1; q1 = q0 + r1 (a - z q0); q = q1 + r1 (a - z q1) - which is bit for bit what
//    v_div_scale / v_div_fmas / v_div_fixup compute when they do not rescale, i.e. for 2^-20 <= z <= 2^20 and a quotient
//    that is neither denormal nor huge (isx_selftest_division compares the two on random operands).  Pixels with z outside
//    that range (z <= 0, W:61, included) and quotients outside the source image are not produced here at all (see below),
//    and a numerator too small for the recurrence's error terms (|a| < 2^-60, quotient < 2^-40) gives cvRound(q * 32) = 0
//    whichever way its last bits fall.
//  * cvRound(q * 32) is the magic-number add (1.5 * 2^23: the FPU's own round-half-even, bits 0..22 then hold the integer),
//    after one v_med3_f32 that clamps the coordinate to the range in which both bilinear rows lie inside the image and maps
//    NaN to the lower bound.  A pixel is "fast" iff clamping changed nothing (one compare per axis) and z is in range.
//  * fast pixels read their two 12-byte windows (always in bounds: the clamped coordinates are used for the addresses) and
//    blend them with v_dot4_u32_u8 / v_mad_u32_u24 as k_warp_img_mask does; their mask byte is 255 (in range implies
//    cvRound(x) in [0, cols), cvRound(y) in [0, rows)).
//  * everything else - reflected borders, z <= 0, the partial thread at the right edge, sources too small for a window - is
//    fixed up after the row loop by the generic per-byte sampler, kept out of line: the hot loop has no call in it.
// ------------------------------------------------------------------------------------------------
constexpr float DIV_Z_LO = 9.5367431640625e-07f;   // 2^-20
constexpr float DIV_Z_HI = 1048576.f;              // 2^20
constexpr float RNE_MAGIC = 12582912.f;            // 1.5 * 2^23

struct TileDst {
    unsigned char* img; unsigned img_step;
    unsigned char* mask; unsigned mask_step;
    int w, h;
    int bx0;        // first 64-column block of this launch (isx_warper_set_dst_columns), 0 = the whole tile
    int xg;         // > 0: runs of xg horizontally adjacent blocks go to ONE XCD (see k_warp_tile); xmagic = 2^32 / gridDim.x + 1
    unsigned xmagic;
};

// The rows of a thread that did not qualify for the fast path (bit i of `slow`: the thread's four pixels of row row0 + i),
// by the generic arithmetic - z <= 0 sentinel (W:61), BORDER_REFLECT taps, per-byte loads and stores.  Out of line and
// written for few registers (the kernel's register count is the larger of its own and its callee's): one pixel at a time,
// no unrolling.  The pointers point into the kernel-argument segment.
// cv::borderInterpolate(p, n, BORDER_REFLECT) as a plain loop (isxd::reflect falls back to an out-of-line call, and a call
// inside the fix-up would push its live values into the callee-saved registers v40+, which count for the kernel)
__device__ __forceinline__ int reflect_inline(int p, int n) {
    if (n == 1) return 0;
    while ((unsigned)p >= (unsigned)n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

template <bool OUT16>
__device__ __forceinline__ void warp_tile_fixup(const Proj* p, const MapTabs* t, const SrcView* img, const TileDst* d, int dx0, int row0, unsigned slow, int nrows, const unsigned char* lut = nullptr) {
    const unsigned char* data = img->data;
    const size_t step = img->step;
    const int rows = img->rows, cols = img->cols;
#pragma unroll 1
    for (int j = 0; j < 4 * nrows; ++j) {
        const int i = j >> 2, k = j & 3, dy = row0 + i, dx = dx0 + k;
        if (dy >= d->h || dx >= d->w || !((slow >> i) & 1u)) continue;
        float mx, my;
        map_backward(*p, *t, dx, dy, mx, my);
        // cv::remap INTER_LINEAR, BORDER_REFLECT on CV_8UC3 (sample_linear's arithmetic)
        const int isx = cvround_x86(mx * 32.f), isy = cvround_x86(my * 32.f);
        const int fx = isx & 31, fy = isy & 31;
        const int sx = clamp_short(isx >> 5), sy = clamp_short(isy >> 5);
        const unsigned char* r0 = data + (size_t)reflect_inline(sy, rows) * step;
        const unsigned char* r1 = data + (size_t)reflect_inline(sy + 1, rows) * step;
        const int c0 = reflect_inline(sx, cols) * 3, c1 = reflect_inline(sx + 1, cols) * 3;
        int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
        if ((fx | fy) == 0) { w0 = 32767; w3 = 1; }
        // masks[i].setTo(255) (W:213-214) warped NEAREST / CONSTANT: 255 iff cvRound(x) in [0, cols) and cvRound(y) in [0, rows)
        const int nx = clamp_short(cvround_x86(mx)), ny = clamp_short(cvround_x86(my));
        if (d->mask) d->mask[(size_t)dy * d->mask_step + dx] = ((unsigned)nx < (unsigned)cols && (unsigned)ny < (unsigned)rows) ? 255 : 0;
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            int v = sat_u8((r0[c0 + c] * w0 + r0[c1 + c] * w1 + r1[c0 + c] * w2 + r1[c1 + c] * w3 + (1 << 14)) >> 15);
            if (lut) v = lut[v];
            if constexpr (OUT16) ((short*)(d->img + (size_t)dy * d->img_step))[(size_t)dx * 3 + c] = (short)v;
            else (d->img + (size_t)dy * d->img_step)[(size_t)dx * 3 + c] = (unsigned char)v;
        }
    }
}

// a * b + c on 24-bit unsigned operands (v_mul_u32_u24 / v_mad_u32_u24, full rate).  Not inline assembly: an asm statement that
// reads the result of a v_dot4 a few instructions earlier escapes the compiler's hazard recogniser (gfx950 wants wait states between
// a DOT instruction and a VALU read of its result) and reads a stale register.
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }

// lut (GAIN kernels): GainCompensator::apply (W:241-244) folded into the warp's store - multiply(image, gain, image) on a CV_8U image is
// saturate_cast<uchar>(cvRound((double)byte * gain)) per byte, a function of the byte alone: 256 entries computed on the host with the
// arithmetic of isx_gain_apply (isx_warper_set_gain), applied to the remapped byte (NOT to the source: remap of scaled pixels differs)
// The table travels in the kernel arguments (256 bytes): changing the gain between two tiles costs no upload and no synchronisation.
struct WarpTileArgs { Proj p; MapTabs t; SrcView img; TileDst d; unsigned lut[64]; };
typedef unsigned WV3 __attribute__((ext_vector_type(3), aligned(1)));      // a 12-byte run of a destination row, any alignment
typedef unsigned WV1 __attribute__((aligned(1)));

// one pixel of the fixed-point bilinear from its two 12-byte windows: window pixel 0 weighs wA, pixel 1 weighs wB (wA + wB = 32),
// the upper row gy, the lower fy (gy + fy = 32): (sum of the BilinearTab_i products + 2^14) >> 15 == (S + 512) >> 10 with the
// separable exact S (see k_warp_img_mask).  Returns b | g << 8 | r << 16.
// the same from the taps' own bytes: l = b0 g0 r0 b1, h = g1 r1 . . of the upper (0) and lower (1) row
__device__ __forceinline__ unsigned sample_taps(unsigned l0, unsigned h0, unsigned l1, unsigned h1, unsigned wA, unsigned wB, unsigned gy, unsigned fy) {
    const unsigned wb = wA | (wB << 24), wgl = wA << 8, wgh = wB, wrl = wA << 16, wrh = wB << 8;
    const unsigned t0b = __builtin_amdgcn_udot4(l0, wb, 0u, false), t1b = __builtin_amdgcn_udot4(l1, wb, 0u, false);
    const unsigned t0g = __builtin_amdgcn_udot4(h0, wgh, __builtin_amdgcn_udot4(l0, wgl, 0u, false), false);
    const unsigned t1g = __builtin_amdgcn_udot4(h1, wgh, __builtin_amdgcn_udot4(l1, wgl, 0u, false), false);
    const unsigned t0r = __builtin_amdgcn_udot4(h0, wrh, __builtin_amdgcn_udot4(l0, wrl, 0u, false), false);
    const unsigned t1r = __builtin_amdgcn_udot4(h1, wrh, __builtin_amdgcn_udot4(l1, wrl, 0u, false), false);
    const unsigned c0 = mad24(t0b, gy, mad24(t1b, fy, 512u)) >> 10;
    const unsigned c1 = mad24(t0g, gy, mad24(t1g, fy, 512u)) >> 10;
    const unsigned c2 = mad24(t0r, gy, mad24(t1r, fy, 512u)) >> 10;
    return c0 | (c1 << 8) | (c2 << 16);
}
__device__ __forceinline__ unsigned sample_windows(const U3& v0, const U3& v1, unsigned o0, unsigned o1, unsigned wA, unsigned wB, unsigned gy, unsigned fy) {
    const unsigned s0 = o0 & 3u, s1 = o1 & 3u;
    const unsigned l0 = __builtin_amdgcn_alignbyte(v0.y, v0.x, s0), h0 = __builtin_amdgcn_alignbyte(v0.z, v0.y, s0);   // b0 g0 r0 b1 | g1 r1 . .
    const unsigned l1 = __builtin_amdgcn_alignbyte(v1.y, v1.x, s1), h1 = __builtin_amdgcn_alignbyte(v1.z, v1.y, s1);
    return sample_taps(l0, h0, l1, h1, wA, wB, gy, fy);
}

// cv::borderInterpolate(p, n, BORDER_REFLECT) for p in [-n, 2n - 1] (at most one reflection): p < 0 -> -p - 1 = ~p, p >= n -> 2n - 1 - p
__device__ __forceinline__ int reflect_once(int p, int n2m1) {
    const int q = p ^ (p >> 31);
    return min(q, n2m1 - q);
}

// ablation study of the tile warp (tools/ab_libs.sh builds, profiles/round3_warp_ablation.txt): bit 0 no stores, bit 1 every window from
// one cache-resident source row, bit 2 no transform (coordinates = a shift), bit 3 non-temporal stores; WARP_WAVES waves per workgroup
#ifndef WARP_ABL
#define WARP_ABL 0
#endif
#ifndef WARP_WAVES
#define WARP_WAVES 4
#endif
#ifndef WARP_WPE
#define WARP_WPE 7
#endif
// MASK = false: the image alone - RotationWarper::warp(img, K, R, INTER_LINEAR, BORDER_REFLECT) as the reference calls it (W:229), the
// mask being a call of its own (W:232, k_warp_mask_tile); d.mask is then null
// The tile's geometry without the gain table: what a BATCHED launch carries per tile (k_warp_tile_batch)
struct WarpTileGeom { Proj p; MapTabs t; SrcView img; TileDst d; };
// ka: the same four structs where they lie in the kernel-argument segment (the out-of-line fix-up takes pointers); klut: the gain table there
template <int KIND, bool OUT16, bool VEC, bool MASK, bool GAIN>
__device__ __forceinline__ void warp_tile_body(const Proj& p, const MapTabs& t, const SrcView& img, const TileDst& d, const WarpTileGeom* ka, const unsigned* klut) {
    __shared__ unsigned char s_lut[GAIN ? 256 : 4];
    if constexpr (GAIN) {      // the 256-entry gain table, once per workgroup (before any thread leaves: everyone reaches the barrier)
        if (threadIdx.x < 64) ((unsigned*)s_lut)[threadIdx.x] = klut[threadIdx.x];
        __syncthreads();
    }
    // A wave is 64 pixels wide and 4 rows tall (16 lanes x 4 pixels per row), a block 64 x 16: the band of border pixels along the
    // left and right edge of the warped tile is a few dozen pixels wide, so with 256 x 1 waves every row's first and last wave crossed
    // it (tier 2); with 64 x 4 waves a quarter as many do.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Which block of the tile this workgroup is.  The hardware deals the workgroups of a launch to the 8 XCDs round robin in dispatch order (x
    // fastest), so horizontally adjacent blocks - whose 192-byte row segments of the source share the 128-byte lines at their common edge -
    // always sit on different XCDs, and each XCD's L2 fetches the shared lines itself: 47.4 MB fetched per 4K tile for 24.9 MB of source
    // (profiles/round5_traffic.json).  With d.xg > 0 the blocks are renumbered inside every aligned group of 8 xg consecutive ones so that XCD k
    // receives the k-th run of xg ADJACENT blocks of the group, one after the other: neighbours meet in one L2, and every XCD still gets every
    // eighth run of the tile (the plain round robin's even spread of cheap and dear blocks - round 2's band order lost on exactly that).
    unsigned bxu = blockIdx.x, byu = blockIdx.y;
    if (d.xg > 0) {
        const unsigned gx = gridDim.x, total = gx * gridDim.y, L = blockIdx.y * gx + blockIdx.x, grp = 8u * (unsigned)d.xg;
        unsigned lin = L;
        if (L < total - total % grp) {
            const unsigned base = L - L % grp, r = L - base, xcd = r & 7u, i = r >> 3;     // the i-th block this XCD receives from the group
            lin = base + xcd * (unsigned)d.xg + i;
        }
        byu = __umulhi(lin, d.xmagic);          // lin / gx (exact: lin * gx < 2^32)
        bxu = lin - byu * gx;
    }
    const int dx0 = (((int)bxu + d.bx0) * 16 + (lane & 15)) * 4;
    // Row blocks are taken alternately from the top and from the bottom of the tile: the rows near the tile's upper and lower edge are
    // where waves cross the image border (tier 2 below, the occasional generic pixel) and live several times longer than interior
    // waves - dispatched first they overlap with the rest of the launch, dispatched last they were its tail.
    const int by = (byu & 1u) ? (int)gridDim.y - 1 - (int)(byu >> 1) : (int)(byu >> 1);
    const int dy = by * (4 * WARP_WAVES) + wv * 4 + (lane >> 4);
    if (dy >= d.h || dx0 >= d.w) return;
    const bool whole = VEC && dx0 + 4 <= d.w;       // four real columns and dword-aligned rows: vector stores
    // ---- mapBackward (W:46-63): the transform of the thread's four columns in this row -------------------------------
    const float4 cs4 = *(const float4*)(t.col_s + dx0), cc4 = *(const float4*)(t.col_c + dx0);   // tables are padded to 4 floats
    const f32x2 cs[2] = {{cs4.x, cs4.y}, {cs4.z, cs4.w}}, cc[2] = {{cc4.x, cc4.y}, {cc4.z, cc4.w}};
    // The x and y rows of k_rinv carry the factor 32 of cv::remap's fixed-point coordinate cvRound(32 x): a power of two commutes with
    // every rounding on the way (products, sums, the division), so (32 k) . v / z has the bits of 32 (k . v / z).
    const float kx0 = p.k_rinv[0] * 32.f, kx1 = p.k_rinv[1] * 32.f, kx2 = p.k_rinv[2] * 32.f, ky0 = p.k_rinv[3] * 32.f, ky1 = p.k_rinv[4] * 32.f, ky2 = p.k_rinv[5] * 32.f;
    f32x2 X[2], Y[2], Z[2];
    if constexpr (KIND == ISX_WARP_CYLINDRICAL) {
        const float ra = t.row_a[dy];                                                            // y_ = v / scale  W:49,52
        const f32x2 qx = splat2(kx1 * ra), qy = splat2(ky1 * ra), qz = splat2(p.k_rinv[7] * ra);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            X[h] = (splat2(kx0) * cs[h] + qx) + splat2(kx2) * cc[h];                             // W:56 (x 32)
            Y[h] = (splat2(ky0) * cs[h] + qy) + splat2(ky2) * cc[h];                             // W:57 (x 32)
            Z[h] = (splat2(p.k_rinv[6]) * cs[h] + qz) + splat2(p.k_rinv[8]) * cc[h];             // W:58
        }
    } else {
        const float ra = t.row_a[dy], rb = t.row_b[dy];                                         // sinf(pi - v), cosf(pi - v)
        const f32x2 qx = splat2(kx1 * rb), qy = splat2(ky1 * rb), qz = splat2(p.k_rinv[7] * rb);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 x_ = splat2(ra) * cs[h], z_ = splat2(ra) * cc[h];
            X[h] = (splat2(kx0) * x_ + qx) + splat2(kx2) * z_;
            Y[h] = (splat2(ky0) * x_ + qy) + splat2(ky2) * z_;
            Z[h] = (splat2(p.k_rinv[6]) * x_ + qz) + splat2(p.k_rinv[8]) * z_;
        }
    }
    // 32 x / z, 32 y / z (W:60): cvRound of these is cv::remap's fixed-point coordinate
    float tx[4], ty[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 r0 = {__builtin_amdgcn_rcpf(Z[h].x), __builtin_amdgcn_rcpf(Z[h].y)};
        const f32x2 r1 = refine_rcp(Z[h], r0);
        const f32x2 ax = div_by_refined(X[h], Z[h], r1), ay = div_by_refined(Y[h], Z[h], r1);
        tx[2 * h] = ax.x; tx[2 * h + 1] = ax.y; ty[2 * h] = ay.x; ty[2 * h + 1] = ay.y;
    }
    if (WARP_ABL & 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { tx[k] = (float)(32 * (dx0 + k) + 7); ty[k] = (float)(32 * dy + 5); Z[k >> 1] = splat2(1.f); }
    }
    // z of the four pixels inside the division's guarded range?  (A NaN slips through min / max and is caught below: its quotient
    // is NaN, which no clamp leaves unchanged.)  Otherwise - z <= 0 (W:61) included - the generic code path does the thread's row.
    const float zmin = fminf(fminf(Z[0].x, Z[0].y), fminf(Z[1].x, Z[1].y)), zmax = fmaxf(fmaxf(Z[0].x, Z[0].y), fmaxf(Z[1].x, Z[1].y));
    const int rows = img.rows, cols = img.cols;
    const unsigned step = (unsigned)img.step;
    const unsigned mis = (unsigned)((uintptr_t)img.data & 3);
    const unsigned char* abase = img.data - mis;
    bool generic = !((zmin >= DIV_Z_LO) & (zmax <= DIV_Z_HI)) || cols < 2 || rows < 3;
    // ---- tier 1: both bilinear rows and columns inside the image (and not in its last row: the 12-byte window of the lower row then
    // ends inside the buffer).  One clamp + one compare per axis; the clamped coordinate makes every address valid.
    const float lo = -0.5f;
    const float hi_x = __uint_as_float(__float_as_uint((float)(32 * (cols - 1)) - 0.5f) - 1u);     // largest float below 32 (cols - 1) - 1/2
    const float hi_y = __uint_as_float(__float_as_uint((float)(32 * (rows - 2)) - 0.5f) - 1u);
    const unsigned addr_c = mis - (0x20000u * step) - 0x60000u;   // mantissa of (t + 1.5 * 2^23) = 2^22 + cvRound(t): sx = (m >> 5) - 2^17
    unsigned px[4];
    float cxs[4], cys[4];
    bool all1 = !generic;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cxs[k] = __builtin_amdgcn_fmed3f(tx[k], lo, hi_x); cys[k] = __builtin_amdgcn_fmed3f(ty[k], lo, hi_y);
        all1 = all1 & (cxs[k] == tx[k]) & (cys[k] == ty[k]);
    }
    unsigned m4 = 0xffffffffu;              // tier-1 pixels: cvRound(x) in [0, cols), cvRound(y) in [0, rows) -> mask 255 (W:213-214, W:232)
    if (__builtin_amdgcn_ballot_w64(!all1) == 0ull) {      // wave-uniform: every pixel of the wave is a tier-1 pixel
        // (Byte-misaligned global_load_dwordx2 at the taps themselves - no window, no v_alignbyte - was tried in round 3: right bytes, but
        // this gather then takes 62 us per pair of tiles instead of 45; the regular 6-byte stride of the collapse step does not mind.)
        unsigned o0[4], fxy[4];
        U3 v0[4], v1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned bx = __float_as_uint(cxs[k] + RNE_MAGIC), by = __float_as_uint(cys[k] + RNE_MAGIC);
            fxy[k] = (bx & 31u) | ((by & 31u) << 8);
            o0[k] = mad24(__builtin_amdgcn_ubfe(by, 5, 18), step, mad24(__builtin_amdgcn_ubfe(bx, 5, 18), 3u, addr_c));
            if (WARP_ABL & 2) o0[k] = mad24(__builtin_amdgcn_ubfe(bx, 5, 18), 3u, mis + 12u) - 0x60000u;      // row 0: cache-resident
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {       // all eight loads in flight together
            v0[k] = *(const U3*)(abase + (o0[k] & ~3u));
            v1[k] = *(const U3*)(abase + ((o0[k] + step) & ~3u));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned fx = fxy[k] & 255u, fy = fxy[k] >> 8;
            px[k] = sample_windows(v0[k], v1[k], o0[k], o0[k] + step, 32u - fx, fx, 32u - fy, fy);
        }
    } else {
        // ---- tier 2: a wave that crosses the image border.  BORDER_REFLECT keeps the two taps of an axis adjacent (possibly in reverse
        // order) or puts them on the same pixel, so the two 12-byte windows still hold every tap: the window starts at the smaller
        // tap column and the horizontal weights go where the taps landed.  Interior pixels of the wave take the same code.
        const float big = 2097152.f;                   // |32 x| < 2^21: the magic-number rounding stays exact, |sx| < 2^16
        const float mhx = (float)(32 * cols - 16), mhy = (float)(32 * rows - 16);      // 32 (cols - 1/2): the upper tie of cvRound(x) <= cols - 1
        const float mx_hi = ((cols - 1) & 1) ? mhx : __uint_as_float(__float_as_uint(mhx) + 1u);   // ... rounds to cols - 1 iff that is even
        const float my_hi = ((rows - 1) & 1) ? mhy : __uint_as_float(__float_as_uint(mhy) + 1u);
        // A window is read from its aligned start; at the very end of the buffer that start is pulled back to the buffer's last aligned 12
        // bytes (end4: the end rounded up to a dword - the dword holding the last valid byte is readable as a whole) and the taps sit up
        // to six bytes into it: the window's dwords are then rotated by one before the usual byte alignment.
        const unsigned end4 = ((unsigned)(rows - 1) * step + (unsigned)cols * 3u + mis + 3u) & ~3u;
        const bool big_enough = (cols >= 2) & (rows >= 3) & (end4 >= 12u);
        const unsigned last_win = end4 >= 12u ? end4 - 12u : 0u;
        unsigned al0[4], al1[4], wab[4];      // aligned window starts
        unsigned shifts = 0;                  // byte offset of the taps inside their windows (0 .. 6), three bits per window
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float cx = __builtin_amdgcn_fmed3f(tx[k], -big, big), cy = __builtin_amdgcn_fmed3f(ty[k], -big, big);
            const int isx = (int)(__float_as_uint(cx + RNE_MAGIC) - 0x4B400000u), isy = (int)(__float_as_uint(cy + RNE_MAGIC) - 0x4B400000u);
            const int sx = isx >> 5, sy = isy >> 5, fx = isx & 31, fy = isy & 31;
            // one reflection at most: sx, sx + 1 in [-cols, 2 cols - 1], same for the rows
            // (sources too small for a window - a single column, fewer than 3 rows, fewer than 12 bytes - take the generic path and issue no window load)
            const bool ok = big_enough & (cx == tx[k]) & (cy == ty[k]) & ((unsigned)(sx + cols) < (unsigned)(3 * cols - 1)) & ((unsigned)(sy + rows) < (unsigned)(3 * rows - 1));
            const int c0 = reflect_once(sx, 2 * cols - 1), c1 = reflect_once(sx + 1, 2 * cols - 1);
            const int r0 = reflect_once(sy, 2 * rows - 1), r1 = reflect_once(sy + 1, 2 * rows - 1);
            const int cb = min(min(c0, c1), cols - 2);                               // window = pixels cb, cb + 1
            const unsigned wA = (c0 == cb ? 32u - fx : 0u) + (c1 == cb ? (unsigned)fx : 0u);
            if (!ok) generic = true;
            const unsigned cb3 = ok ? (unsigned)cb * 3u + mis : mis;
            const unsigned q0 = ok ? __umul24((unsigned)r0, step) + cb3 : mis, q1 = ok ? __umul24((unsigned)r1, step) + cb3 : mis;   // the taps' byte offsets
            const unsigned a0 = min(q0 & ~3u, last_win), a1 = min(q1 & ~3u, last_win);
            al0[k] = a0; al1[k] = a1;
            shifts |= ((q0 - a0) << (6 * k)) | ((q1 - a1) << (6 * k + 3));
            wab[k] = wA | ((unsigned)fy << 8);
            // mask of an all-255 source, NEAREST / CONSTANT: 255 iff cvRound(x) in [0, cols) and cvRound(y) in [0, rows) - an interval test
            // on 32 x, 32 y (round-half-even: the tie -1/2 rounds to 0, the upper tie to cols - 1 iff cols - 1 is even; NaN fails)
            const bool inside = (tx[k] >= -16.f) & (tx[k] < mx_hi) & (ty[k] >= -16.f) & (ty[k] < my_hi);
            if (!inside) m4 &= ~(255u << (8 * k));
        }
        U3 w0[4], w1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w0[k] = *(const U3*)(abase + al0[k]);
            w1[k] = *(const U3*)(abase + al1[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned wA = wab[k] & 255u, fy = wab[k] >> 8;
            const unsigned s0 = (shifts >> (6 * k)) & 7u, s1 = (shifts >> (6 * k + 3)) & 7u;
            if (s0 & 4u) w0[k] = U3{w0[k].y, w0[k].z, 0u};
            if (s1 & 4u) w1[k] = U3{w1[k].y, w1[k].z, 0u};
            px[k] = sample_windows(w0[k], w1[k], s0, s1, wA, 32u - wA, 32u - fy, fy);
        }
    }
    if constexpr (GAIN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = (unsigned)s_lut[px[k] & 255u] | ((unsigned)s_lut[(px[k] >> 8) & 255u] << 8) | ((unsigned)s_lut[(px[k] >> 16) & 255u] << 16);
    }
    // ---- stores ------------------------------------------------------------------------------------------------------------
    if ((WARP_ABL & 1) && d.w > -3) { if (px[0] + px[1] + px[2] + px[3] + m4 == 0x12345u) d.mask[0] = 1; return; }
    if (whole) {
        // one vector store per run, typed for ANY alignment (WV3 / WV1: aligned(1)): a dense cv::Mat row of 3425 CV_8UC3 pixels starts on
        // an odd byte, unaligned global access is legal on this part and a wave's runs are contiguous either way
        if constexpr (OUT16) {
            unsigned char* q = d.img + (__umul24((unsigned)dy, d.img_step) + (unsigned)dx0 * 6u);
#pragma unroll
            for (int k = 0; k < 4; k += 2) {   // 2 pixels = 6 shorts = 3 dwords
                const unsigned a0 = px[k], b2 = px[k + 1];
                *(WV3*)(q + 12 * (k / 2)) = WV3{(a0 & 255) | (((a0 >> 8) & 255) << 16), ((a0 >> 16) & 255) | ((b2 & 255) << 16), ((b2 >> 8) & 255) | (((b2 >> 16) & 255) << 16)};
            }
        } else {
            unsigned char* q = d.img + (__umul24((unsigned)dy, d.img_step) + (unsigned)dx0 * 3u);
            const unsigned q0 = px[0] | (px[1] << 24);                                   // b0 g0 r0 b1
            const unsigned q1 = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u);        // g1 r1 b2 g2
            const unsigned q2 = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u);        // r2 b3 g3 r3
            if (WARP_ABL & 8) __builtin_nontemporal_store(WV3{q0, q1, q2}, (WV3*)q);
            else *(WV3*)q = WV3{q0, q1, q2};
        }
        if constexpr (MASK) {
            if (WARP_ABL & 8) __builtin_nontemporal_store(m4, (WV1*)(d.mask + (__umul24((unsigned)dy, d.mask_step) + (unsigned)dx0)));
            else *(WV1*)(d.mask + (__umul24((unsigned)dy, d.mask_step) + (unsigned)dx0)) = m4;
        }
    } else {    // the partial group at the right edge, or destination rows that are not dword aligned: per-pixel stores
#pragma unroll 1
        for (int k = 0; k < 4 && dx0 + k < d.w; ++k) {
            const unsigned v = k == 0 ? px[0] : (k == 1 ? px[1] : (k == 2 ? px[2] : px[3]));
            if constexpr (OUT16) {
                short* q = (short*)(d.img + (size_t)dy * d.img_step) + (size_t)(dx0 + k) * 3;
                q[0] = (short)(v & 255); q[1] = (short)((v >> 8) & 255); q[2] = (short)((v >> 16) & 255);
            } else {
                unsigned char* q = d.img + (size_t)dy * d.img_step + (size_t)(dx0 + k) * 3;
                q[0] = (unsigned char)v; q[1] = (unsigned char)(v >> 8); q[2] = (unsigned char)(v >> 16);
            }
            if constexpr (MASK) d.mask[(size_t)dy * d.mask_step + dx0 + k] = (unsigned char)(m4 >> (8 * k));
        }
    }
    // ---- the rare rest (z out of the guarded range incl. the z <= 0 sentinel, more than one reflection, sources too small for a
    // window, the last columns of the buffer's last row): the whole 4-pixel row of the thread again, by the generic code path
    if (generic) warp_tile_fixup<OUT16>(&ka->p, &ka->t, &ka->img, &ka->d, dx0, dy, 1u, 1, GAIN ? (const unsigned char*)klut : nullptr);
}
static_assert(offsetof(WarpTileArgs, lut) == sizeof(WarpTileGeom), "WarpTileArgs = WarpTileGeom + the gain table");
template <int KIND, bool OUT16, bool VEC, bool MASK = true, bool GAIN = false>
__global__ __launch_bounds__(64 * WARP_WAVES) __attribute__((amdgpu_waves_per_eu(WARP_WPE))) void k_warp_tile(WarpTileArgs a) {
    const WarpTileArgs* ka = (const WarpTileArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    warp_tile_body<KIND, OUT16, VEC, MASK, GAIN>(a.p, a.t, a.img, a.d, (const WarpTileGeom*)ka, ka->lut);
}
// Several tiles in ONE launch (blockIdx.z = tile; round 6, isx_warper_begin_batch): the planned warps of a step are independent of each other,
// and as separate launches on one stream the second waited for the first one's last waves - the long-lived ones that cross the image border -
// and paid a dispatch ramp of its own (~3 us of a 175 us step per extra launch).  The grid is the largest tile's; blocks past a smaller
// tile's edge leave at once.  Same body, same bits.
constexpr int WARP_BATCH_MAX = 8;
struct WarpTileBatch { WarpTileGeom a[WARP_BATCH_MAX]; };
static_assert(sizeof(WarpTileBatch) <= 4096, "kernel-argument limit");
template <int KIND, bool OUT16, bool VEC>
__global__ __launch_bounds__(64 * WARP_WAVES) __attribute__((amdgpu_waves_per_eu(WARP_WPE))) void k_warp_tile_batch(WarpTileBatch b) {
    const WarpTileGeom& a = b.a[blockIdx.z];
    const WarpTileGeom* ka = &((const WarpTileBatch*)__builtin_amdgcn_kernarg_segment_ptr())->a[blockIdx.z];
    warp_tile_body<KIND, OUT16, VEC, true, false>(a.p, a.t, a.img, a.d, ka, nullptr);
}


// ------------------------------------------------------------------------------------------------
// k_warp_mask_tile: RotationWarper::warp(mask, K, R, INTER_NEAREST, BORDER_CONSTANT) of a CV_8U mask (W:232) as a call of its own - what
// the reference's main() and the cv adapter (include/imagestitch_cv.hpp) issue after the image's warp.  The source mask is read, whatever
// it holds: dst = mask(cvRound(y), cvRound(x)) inside, 0 outside (cv::remap NEAREST / BORDER_CONSTANT, coordinates saturated to short).
// One thread = 4 consecutive destination columns: the transform of k_warp_tile (hoisted separable products, shared-reciprocal division, the
// bits of W:56-60) without the factor 32, cvRound as the magic-number add after a clamp to +-2^21 (NaN goes to the lower bound: outside,
// as cvRound's INT_MIN saturated to -32768 is), four byte gathers, one dword store.  z outside the division's guarded range (z <= 0,
// W:61, included): map_backward + sample_nearest for the thread's pixels.
// ------------------------------------------------------------------------------------------------
struct WarpMaskArgs { Proj p; MapTabs t; SrcView src; unsigned char* dst; unsigned dst_step; int w, h; };

template <int KIND, bool VEC>
__global__ __launch_bounds__(256) void k_warp_mask_tile(WarpMaskArgs a) {
    const Proj& p = a.p; const MapTabs& t = a.t; const SrcView& src = a.src;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dx0 = (blockIdx.x * 16 + (lane & 15)) * 4;
    const int dy = blockIdx.y * 16 + wv * 4 + (lane >> 4);
    if (dy >= a.h || dx0 >= a.w) return;
    const float4 cs4 = *(const float4*)(t.col_s + dx0), cc4 = *(const float4*)(t.col_c + dx0);   // tables are padded to 4 floats
    const f32x2 cs[2] = {{cs4.x, cs4.y}, {cs4.z, cs4.w}}, cc[2] = {{cc4.x, cc4.y}, {cc4.z, cc4.w}};
    f32x2 X[2], Y[2], Z[2];
    if constexpr (KIND == ISX_WARP_CYLINDRICAL) {
        const float ra = t.row_a[dy];
        const f32x2 qx = splat2(p.k_rinv[1] * ra), qy = splat2(p.k_rinv[4] * ra), qz = splat2(p.k_rinv[7] * ra);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            X[h] = (splat2(p.k_rinv[0]) * cs[h] + qx) + splat2(p.k_rinv[2]) * cc[h];             // W:56
            Y[h] = (splat2(p.k_rinv[3]) * cs[h] + qy) + splat2(p.k_rinv[5]) * cc[h];             // W:57
            Z[h] = (splat2(p.k_rinv[6]) * cs[h] + qz) + splat2(p.k_rinv[8]) * cc[h];             // W:58
        }
    } else {
        const float ra = t.row_a[dy], rb = t.row_b[dy];
        const f32x2 qx = splat2(p.k_rinv[1] * rb), qy = splat2(p.k_rinv[4] * rb), qz = splat2(p.k_rinv[7] * rb);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 x_ = splat2(ra) * cs[h], z_ = splat2(ra) * cc[h];
            X[h] = (splat2(p.k_rinv[0]) * x_ + qx) + splat2(p.k_rinv[2]) * z_;
            Y[h] = (splat2(p.k_rinv[3]) * x_ + qy) + splat2(p.k_rinv[5]) * z_;
            Z[h] = (splat2(p.k_rinv[6]) * x_ + qz) + splat2(p.k_rinv[8]) * z_;
        }
    }
    float tx[4], ty[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 r0 = {__builtin_amdgcn_rcpf(Z[h].x), __builtin_amdgcn_rcpf(Z[h].y)};
        const f32x2 r1 = refine_rcp(Z[h], r0);
        const f32x2 ax = div_by_refined(X[h], Z[h], r1), ay = div_by_refined(Y[h], Z[h], r1);    // W:60
        tx[2 * h] = ax.x; tx[2 * h + 1] = ax.y; ty[2 * h] = ay.x; ty[2 * h + 1] = ay.y;
    }
    const float zmin = fminf(fminf(Z[0].x, Z[0].y), fminf(Z[1].x, Z[1].y)), zmax = fmaxf(fmaxf(Z[0].x, Z[0].y), fmaxf(Z[1].x, Z[1].y));
    const int rows = src.rows, cols = src.cols;
    const unsigned step = (unsigned)src.step;
    unsigned m4 = 0u;
    if ((zmin >= DIV_Z_LO) & (zmax <= DIV_Z_HI)) {
        const float big = 2097152.f;            // 2^21: beyond any source (cols, rows <= 32767), the magic-number rounding exact inside
        unsigned off[4];
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float cx = __builtin_amdgcn_fmed3f(tx[k], -big, big), cy = __builtin_amdgcn_fmed3f(ty[k], -big, big);
            const int ix = (int)(__float_as_uint(cx + RNE_MAGIC) - 0x4B400000u), iy = (int)(__float_as_uint(cy + RNE_MAGIC) - 0x4B400000u);   // cvRound
            in[k] = ((unsigned)ix < (unsigned)cols) & ((unsigned)iy < (unsigned)rows);
            off[k] = in[k] ? __umul24((unsigned)iy, step) + (unsigned)ix : 0u;
        }
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = src.data[off[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) m4 |= (in[k] ? v[k] : 0u) << (8 * k);
    } else {
#pragma unroll 1
        for (int k = 0; k < 4 && dx0 + k < a.w; ++k) {
            float mx, my;
            map_backward(p, t, dx0 + k, dy, mx, my);
            unsigned char m;
            sample_nearest<unsigned char, 1>(src, mx, my, ISX_BORDER_CONSTANT, &m);
            m4 |= (unsigned)m << (8 * k);
        }
    }
    unsigned char* q = a.dst + (__umul24((unsigned)dy, a.dst_step) + (unsigned)dx0);
    if (VEC && dx0 + 4 <= a.w) *(WV1*)q = m4;
    else for (int k = 0; k < 4 && dx0 + k < a.w; ++k) q[k] = (unsigned char)(m4 >> (8 * k));
}

// the recurrence of k_warp_tile against the compiler's IEEE division on n pseudo-random operand pairs of the guarded range
__global__ void k_selftest_division(unsigned long long seed, int n, unsigned* mismatches) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    auto next = [&]() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 32); };
    // z: any float in [2^-20, 2^20]; a: any finite float of either sign with |a| in [2^-60, 2^60], or exactly 0
    const unsigned zr = next(), ar = next();
    const float z = __uint_as_float(((107u + zr % 41u) << 23) | (next() & 0x7fffffu));
    float a = __uint_as_float((ar & 0x80000000u) | ((67u + (ar >> 8) % 121u) << 23) | (next() & 0x7fffffu));
    if ((ar & 0xffu) == 0) a = 0.f;
    if (!(z >= DIV_Z_LO && z <= DIV_Z_HI)) return;
    const f32x2 Z = {z, z}, A = {a, -a};
    const f32x2 r0 = {__builtin_amdgcn_rcpf(z), __builtin_amdgcn_rcpf(z)};
    const f32x2 q = div_by_refined(A, Z, refine_rcp(Z, r0));
    const float e0 = a / z, e1 = -a / z;
    // (a zero quotient may come out with the other sign - (-0) / z gives +0 here - which cvRound(q * 32) does not see)
    const bool same0 = __float_as_uint(q.x) == __float_as_uint(e0) || (q.x == 0.f && e0 == 0.f);
    const bool same1 = __float_as_uint(q.y) == __float_as_uint(e1) || (q.y == 0.f && e1 == 0.f);
    if (!same0 || !same1) atomicAdd(mismatches, 1u);
}

// buildMaps (W:133-141), API parity only
__global__ __launch_bounds__(256) void k_build_maps(Proj p, MapTabs t, float* xmap, size_t xstep, float* ymap, size_t ystep, int dw, int dh) {
    int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    float mx, my;
    map_backward(p, t, dx, dy, mx, my);
    ((float*)((char*)xmap + (size_t)dy * xstep))[dx] = mx;
    ((float*)((char*)ymap + (size_t)dy * ystep))[dx] = my;
}

// ---- detectResultRoi (W:64-88): full forward scan, min / max reduction ----------------------------
// detectResultRoi scans EVERY source pixel through mapForward (W:72-81): 8.3 M atan2f + sqrtf + divisions
// per 4K tile, VALU-bound.  Only the four extrema matter, so the scan ranks pixels by two cheap
// strictly monotone stand-ins instead (about 40 VALU instructions per pixel instead of 90):
//   u = scale * atan2f(x_, z_)         ~  d = "diamond angle" of (x_, z_) in (-2, 2]   (one v_rcp_f32)
//   v = scale * y_ / sqrtf(x_^2+z_^2)  ~  q = y_ * rsq(x_^2 + z_^2)                     (one v_rsq_f32)
// The extremal pixels found this way (plus every pixel within a tolerance that covers the stand-ins'
// few-ulp error) are then evaluated EXACTLY: on the host with the host's own libm in the synchronous
// path (= what the reference binary computes), or compared with proxy-space thresholds of the planned
// ROI in the sync-free path.
__device__ __forceinline__ void forward_proxy(const Proj& p, float x, float y, float& d, float& q) {
    const float x_ = p.r_kinv[0] * x + p.r_kinv[1] * y + p.r_kinv[2];
    const float y_ = p.r_kinv[3] * x + p.r_kinv[4] * y + p.r_kinv[5];
    const float z_ = p.r_kinv[6] * x + p.r_kinv[7] * y + p.r_kinv[8];
    const float ax = fabsf(x_), az = fabsf(z_);
    const float t = ax * __builtin_amdgcn_rcpf(ax + az);      // |x_| / (|x_| + |z_|) in [0, 1]
    d = copysignf(z_ >= 0.f ? t : 2.f - t, x_);
    q = y_ * __builtin_amdgcn_rsqf(x_ * x_ + z_ * z_);
}

// keys[0..3] = min d, min q, max d, max q (as fkey).  One block scans a band of ROI_ROWS rows, reduces
// through shuffles + LDS and touches the four global keys only when it improves them (520 K contended
// atomics cost 3 ms on this part; a few hundred cost nothing).
// rows: rows per block.  ROI_ROWS for the verification scans of planned warps, which run on a side stream under the step's own kernels
// (few long-lived waves disturb those least: DESIGN.md §6); SYNC_ROWS for the synchronous form, where the host waits for the result
// (eight times the waves: 15 -> 6 us for a 4K source).  blk (optional): every block's own four extrema, which let the candidate pass
// skip the blocks that cannot hold a candidate.
constexpr int ROI_ROWS = 128;
constexpr int SYNC_ROWS = 32;
constexpr int CAND_ROWS = 16;    // rows per block of k_roi_candidates
__global__ __launch_bounds__(256) void k_roi_scan(Proj p, int sw, int sh, unsigned* keys, int rows, float4* blk) {
    __shared__ float red[4][4];
    float tl_u = 3.402823466e+38f, tl_v = 3.402823466e+38f, br_u = -3.402823466e+38f, br_v = -3.402823466e+38f;
    const int y0 = blockIdx.y * rows, y1 = min(y0 + rows, sh);
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x < sw) {
#pragma unroll 4
        for (int y = y0; y < y1; ++y) {
            float u, v;
            forward_proxy(p, (float)x, (float)y, u, v);
            tl_u = (u < tl_u) ? u : tl_u; tl_v = (v < tl_v) ? v : tl_v;     // NaN never wins, as with (std::min)(tl, u)
            br_u = (br_u < u) ? u : br_u; br_v = (br_v < v) ? v : br_v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tl_u = fminf(tl_u, __shfl_xor(tl_u, o)); tl_v = fminf(tl_v, __shfl_xor(tl_v, o));
        br_u = fmaxf(br_u, __shfl_xor(br_u, o)); br_v = fmaxf(br_v, __shfl_xor(br_v, o));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = tl_u; red[1][wv] = tl_v; red[2][wv] = br_u; red[3][wv] = br_v; }
    __syncthreads();
    if (blk != nullptr && threadIdx.x == 0)
        blk[blockIdx.y * gridDim.x + blockIdx.x] = make_float4(fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3])), fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3])),
                                                               fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3])), fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3])));
    if (keys != nullptr && threadIdx.x < 4) {     // (same-address atomics serialise at the memory side: affordable from a few hundred blocks only)
        const int k = threadIdx.x;
        float a = red[k][0], b = red[k][1], c = red[k][2], d = red[k][3];
        if (k < 2) {
            unsigned key = fkey(fminf(fminf(a, b), fminf(c, d)));
            if (key < __hip_atomic_load(&keys[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&keys[k], key);
        } else {
            unsigned key = fkey(fmaxf(fmaxf(a, b), fmaxf(c, d)));
            if (key > __hip_atomic_load(&keys[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&keys[k], key);
        }
    }
}

// second pass of the synchronous path: every pixel whose stand-in is within the tolerance of one of the
// four extrema; the host re-evaluates exactly those with mapForward and its own libm
// The scan's grid in x, CAND_ROWS rows per block in y (finer than the scan's, so that the few blocks that do hold candidates are
// short).  Every block first reduces blk (a few hundred float4, L2-resident) to the four global
// extrema - no atomics on shared keys, which serialise at the memory side when thousands of blocks issue them - and then holds a
// candidate exactly when one of its OWN extrema is within the tolerance of the global one (the stand-ins are recomputed here by the
// same instructions), so all but a handful of blocks leave right away.
__global__ __launch_bounds__(256) void k_roi_candidates(Proj p, int sw, int sh, int* cand_xy, int cap, int* count, int rows, const float4* blk, int scan_rows, int scan_gy) {
    __shared__ float red[4][4];
    float dmin = 3.402823466e+38f, qmin = 3.402823466e+38f, dmax = -3.402823466e+38f, qmax = -3.402823466e+38f;
    const int nblk = gridDim.x * scan_gy;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const float4 b = blk[i];
        dmin = fminf(dmin, b.x); qmin = fminf(qmin, b.y); dmax = fmaxf(dmax, b.z); qmax = fmaxf(qmax, b.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dmin = fminf(dmin, __shfl_xor(dmin, o)); qmin = fminf(qmin, __shfl_xor(qmin, o));
        dmax = fmaxf(dmax, __shfl_xor(dmax, o)); qmax = fmaxf(qmax, __shfl_xor(qmax, o));
    }
    if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; red[0][wv] = dmin; red[1][wv] = qmin; red[2][wv] = dmax; red[3][wv] = qmax; }
    __syncthreads();
    dmin = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3])); qmin = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
    dmax = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3])); qmax = fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3]));
    // tolerances that cover the few-ulp error of v_rcp / v_rsq and of the host's own atan2f many times over (computed here so that
    // the host needs no look at the extrema between the two kernels: one round trip per detectResultRoi)
    const float tol_d = 7.62939453125e-06f;                                   // 2^-17 of a (-2, 2] range
    const float tol_q = 4e-6f * fmaxf(fabsf(qmin), fabsf(qmax)) + 1e-9f;
    const float4 mine = blk[(blockIdx.y * rows / scan_rows) * gridDim.x + blockIdx.x];      // the scan block this (finer) block lies in
    if (!(mine.x <= dmin + tol_d || mine.z >= dmax - tol_d || mine.y <= qmin + tol_q || mine.w >= qmax - tol_q)) return;
    const int y0 = blockIdx.y * rows, y1 = min(y0 + rows, sh);
    for (int y = y0; y < y1; ++y)
        for (int x = blockIdx.x * 256 + threadIdx.x; x < sw; x += gridDim.x * 256) {
            float d, q;
            forward_proxy(p, (float)x, (float)y, d, q);
            if (d <= dmin + tol_d || d >= dmax - tol_d || q <= qmin + tol_q || q >= qmax - tol_q) {
                int i = atomicAdd(count, 1);
                if (i < cap) { cand_xy[2 * i] = x; cand_xy[2 * i + 1] = y; }
            }
        }
}

// ---- spherical projector: detectResultRoiByBorder on the device ---------------------------------------------------------------------
// OpenCV's SphericalWarper finds its ROI from the source's BORDER pixels only (plus two pole tests, which are plain arithmetic on
// K and R and stay on the host): 2 (W + H) mapForwards.  Round 2 evaluated them on the host (0.33 ms for an 8K tile, memoised) and had no
// planned form.  Here one workgroup ranks the border pixels by monotone stand-ins as the cylinder's scan does -
//   u = scale * atan2f(x_, z_)                   ~ the diamond angle d of (x_, z_)
//   v = scale * (pi - acosf(w)), w = y_ / |r|    ~ w itself (v increases with w; NaN -> 0 as `w == w ? w : 0`)
// - and either hands the pixels within a tolerance of the four extrema to the host (synchronous form: exact mapForward with the host's
// libm on a handful of points, one round trip) or folds the extrema into the four keys that k_roi_check_rearm compares with the planned
// ROI's stand-in intervals (planned form: no host round trip, capturable).
__device__ __forceinline__ void forward_proxy_sph(const Proj& p, float x, float y, float& d, float& q) {
    const float x_ = p.r_kinv[0] * x + p.r_kinv[1] * y + p.r_kinv[2];
    const float y_ = p.r_kinv[3] * x + p.r_kinv[4] * y + p.r_kinv[5];
    const float z_ = p.r_kinv[6] * x + p.r_kinv[7] * y + p.r_kinv[8];
    const float ax = fabsf(x_), az = fabsf(z_);
    const float t = ax * __builtin_amdgcn_rcpf(ax + az);
    d = copysignf(z_ >= 0.f ? t : 2.f - t, x_);
    const float w = y_ * __builtin_amdgcn_rsqf(x_ * x_ + y_ * y_ + z_ * z_);
    q = (w == w) ? w : 0.f;
}
__device__ __forceinline__ void border_point(int i, int sw, int sh, int& x, int& y) {      // i in [0, 2 sw + 2 sh): top, bottom, left, right
    if (i < sw) { x = i; y = 0; }
    else if (i < 2 * sw) { x = i - sw; y = sh - 1; }
    else if (i < 2 * sw + sh) { x = 0; y = i - 2 * sw; }
    else { x = sw - 1; y = i - 2 * sw - sh; }
}
// keys != nullptr: planned form (atomics on the four keys, one block: no contention).  cand != nullptr: synchronous form.
__global__ __launch_bounds__(1024) void k_roi_border_sph(Proj p, int sw, int sh, unsigned* keys, int* cand_xy, int cap, int* count) {
    __shared__ float red[4][16];
    const int n = 2 * sw + 2 * sh;
    float dmin = 3.402823466e+38f, qmin = 3.402823466e+38f, dmax = -3.402823466e+38f, qmax = -3.402823466e+38f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        int x, y; float d, q;
        border_point(i, sw, sh, x, y);
        if (p.kind == ISX_WARP_SPHERICAL) forward_proxy_sph(p, (float)x, (float)y, d, q); else forward_proxy(p, (float)x, (float)y, d, q);
        dmin = (d < dmin) ? d : dmin; qmin = (q < qmin) ? q : qmin; dmax = (dmax < d) ? d : dmax; qmax = (qmax < q) ? q : qmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dmin = fminf(dmin, __shfl_xor(dmin, o)); qmin = fminf(qmin, __shfl_xor(qmin, o));
        dmax = fmaxf(dmax, __shfl_xor(dmax, o)); qmax = fmaxf(qmax, __shfl_xor(qmax, o));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = dmin; red[1][wv] = qmin; red[2][wv] = dmax; red[3][wv] = qmax; }
    __syncthreads();
    dmin = red[0][0]; qmin = red[1][0]; dmax = red[2][0]; qmax = red[3][0];
#pragma unroll
    for (int k = 1; k < 16; ++k) { dmin = fminf(dmin, red[0][k]); qmin = fminf(qmin, red[1][k]); dmax = fmaxf(dmax, red[2][k]); qmax = fmaxf(qmax, red[3][k]); }
    if (keys != nullptr && threadIdx.x == 0) {
        atomicMin(&keys[0], fkey(dmin)); atomicMin(&keys[1], fkey(qmin)); atomicMax(&keys[2], fkey(dmax)); atomicMax(&keys[3], fkey(qmax));
    }
    if (cand_xy == nullptr) return;
    const float tol_d = 7.62939453125e-06f;                                   // as k_roi_candidates
    const float tol_q = 4e-6f * fmaxf(fabsf(qmin), fabsf(qmax)) + 1e-9f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        int x, y; float d, q;
        border_point(i, sw, sh, x, y);
        if (p.kind == ISX_WARP_SPHERICAL) forward_proxy_sph(p, (float)x, (float)y, d, q); else forward_proxy(p, (float)x, (float)y, d, q);
        if (d <= dmin + tol_d || d >= dmax - tol_d || q <= qmin + tol_q || q >= qmax - tol_q) {
            const int j = atomicAdd(count, 1);
            if (j < cap) { cand_xy[2 * j] = x; cand_xy[2 * j + 1] = y; }
        }
    }
}

// sync-free path: the scanned extrema must lie inside the stand-in intervals of the planned ROI
// (lo/hi: min d, min q, max d, max q; margins already applied by the host)
struct RoiBounds { float lo[4], hi[4]; };
__global__ void k_roi_check_rearm(unsigned* keys, RoiBounds b, int* mismatches) {
    bool ok = true;
    for (int k = 0; k < 4; ++k) {
        const float v = fkey_inv(keys[k]);
        ok = ok && v >= b.lo[k] && v <= b.hi[k];
    }
    if (!ok) atomicAdd(mismatches, 1);
    keys[0] = 0xffffffffu; keys[1] = 0xffffffffu; keys[2] = 0u; keys[3] = 0u; keys[4] = 0u;
}

// The synchronous border scan with its answer delivered straight into pinned host memory: detectResultRoi has to hand the corner to the
// host (W:160), and the round trip - kernel, copy of {count, candidates}, a kernel that re-arms the device-side keys, a stream
// synchronisation - cost the caller's thread 45 us per call, four times per pair in the reference's own call sequence (W:229, W:232 each
// run detectResultRoi).  Here ONE workgroup ranks the 2 (W + H) border pixels (stand-ins kept in registers between the two passes while
// they fit: 16 per thread), counts the candidates in LDS, writes them and the count into the caller's pinned block and publishes a
// sequence number with a system-scope release store; the host polls that word (detect_roi) - no copy, no second kernel, no
// hipStreamSynchronize.  Candidates beyond the pinned block's CAND_FIRST also go to the device buffer (the host fetches them: rare).
constexpr int PIN_CAND = 1024;      // == CAND_FIRST (checked where that is defined)
struct RoiPin { int seq, count, pad[14]; int cand[2 * PIN_CAND]; };
__global__ __launch_bounds__(1024) void k_roi_border_pin(Proj p, int sw, int sh, RoiPin* pin, int seq, int* cand_dev, int cap) {
    __shared__ float red[4][16];
    __shared__ int s_count;
    constexpr int PER = 16;
    const int n = 2 * sw + 2 * sh;
    const bool fits = n <= PER * 1024;
    float dv[PER], qv[PER];
    float dmin = 3.402823466e+38f, qmin = 3.402823466e+38f, dmax = -3.402823466e+38f, qmax = -3.402823466e+38f;
    if (threadIdx.x == 0) s_count = 0;
    auto proxy = [&](int i, float& d, float& q) {
        int x, y;
        border_point(i, sw, sh, x, y);
        if (p.kind == ISX_WARP_SPHERICAL) forward_proxy_sph(p, (float)x, (float)y, d, q); else forward_proxy(p, (float)x, (float)y, d, q);
    };
    if (fits) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + 1024 * k;
            float d = 0.f, q = 0.f;
            if (i < n) {
                proxy(i, d, q);
                dmin = (d < dmin) ? d : dmin; qmin = (q < qmin) ? q : qmin; dmax = (dmax < d) ? d : dmax; qmax = (qmax < q) ? q : qmax;
            }
            dv[k] = d; qv[k] = q;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 1024) {
            float d, q;
            proxy(i, d, q);
            dmin = (d < dmin) ? d : dmin; qmin = (q < qmin) ? q : qmin; dmax = (dmax < d) ? d : dmax; qmax = (qmax < q) ? q : qmax;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dmin = fminf(dmin, __shfl_xor(dmin, o)); qmin = fminf(qmin, __shfl_xor(qmin, o));
        dmax = fmaxf(dmax, __shfl_xor(dmax, o)); qmax = fmaxf(qmax, __shfl_xor(qmax, o));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = dmin; red[1][wv] = qmin; red[2][wv] = dmax; red[3][wv] = qmax; }
    __syncthreads();
    dmin = red[0][0]; qmin = red[1][0]; dmax = red[2][0]; qmax = red[3][0];
#pragma unroll
    for (int k = 1; k < 16; ++k) { dmin = fminf(dmin, red[0][k]); qmin = fminf(qmin, red[1][k]); dmax = fmaxf(dmax, red[2][k]); qmax = fmaxf(qmax, red[3][k]); }
    const float tol_d = 7.62939453125e-06f;                                   // as k_roi_candidates
    const float tol_q = 4e-6f * fmaxf(fabsf(qmin), fabsf(qmax)) + 1e-9f;
    bool wrote = false;
    auto take = [&](int i, float d, float q) {
        if (d <= dmin + tol_d || d >= dmax - tol_d || q <= qmin + tol_q || q >= qmax - tol_q) {
            int x, y;
            border_point(i, sw, sh, x, y);
            const int j = atomicAdd(&s_count, 1);
            if (j < PIN_CAND) { pin->cand[2 * j] = x; pin->cand[2 * j + 1] = y; wrote = true; }
            if (j < cap) { cand_dev[2 * j] = x; cand_dev[2 * j + 1] = y; }
        }
    };
    if (fits) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + 1024 * k;
            if (i < n) take(i, dv[k], qv[k]);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 1024) {
            float d, q;
            proxy(i, d, q);
            take(i, d, q);
        }
    }
    if (wrote) __threadfence_system();          // the candidates are in host memory before the sequence number is
    __syncthreads();
    if (threadIdx.x == 0) {
        pin->count = s_count;
        __threadfence_system();
        __hip_atomic_store(&pin->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void k_roi_rearm(unsigned* keys) {
    keys[0] = 0xffffffffu; keys[1] = 0xffffffffu; keys[2] = 0u; keys[3] = 0u; keys[4] = 0u;
}

// ---- host-side scalar restatements used for parameter set-up only (O(W+H) work) -----------------
// K.inv() on 3x3 CV_32F: closed form in double, rounded once (OpenCV cv::invert, n == 3);
// Mat products of CV_32F: double accumulation, rounded once (cv::gemm GEMMSingleMul<float,double>).
void mat3_mul(const float a[9], const float b[9], float c[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (double)a[i * 3 + k] * (double)b[k * 3 + j];
            c[i * 3 + j] = (float)s;
        }
}
bool mat3_inv(const float s[9], float d[9]) {
    auto S = [&](int i, int j) { return (double)s[i * 3 + j]; };
    double det = S(0, 0) * (S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) - S(0, 1) * (S(1, 0) * S(2, 2) - S(1, 2) * S(2, 0)) +
                 S(0, 2) * (S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0));
    if (det == 0.0) { for (int i = 0; i < 9; ++i) d[i] = 0.f; return false; }
    double id = 1.0 / det;
    d[0] = (float)((S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) * id);
    d[1] = (float)((S(0, 2) * S(2, 1) - S(0, 1) * S(2, 2)) * id);
    d[2] = (float)((S(0, 1) * S(1, 2) - S(0, 2) * S(1, 1)) * id);
    d[3] = (float)((S(1, 2) * S(2, 0) - S(1, 0) * S(2, 2)) * id);
    d[4] = (float)((S(0, 0) * S(2, 2) - S(0, 2) * S(2, 0)) * id);
    d[5] = (float)((S(0, 2) * S(1, 0) - S(0, 0) * S(1, 2)) * id);
    d[6] = (float)((S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0)) * id);
    d[7] = (float)((S(0, 1) * S(2, 0) - S(0, 0) * S(2, 1)) * id);
    d[8] = (float)((S(0, 0) * S(1, 1) - S(0, 1) * S(1, 0)) * id);
    return true;
}

inline int f2i_host(float v) { return (std::fabs(v) < 2147483648.0f) ? (int)v : INT_MIN; }

// mapForward on the host (W:36-45 / SphericalProjector): used on O(W+H) points only — the spherical
// border scan and the cylindrical extremum candidates
void map_forward_host(const Proj& p, float x, float y, float& u, float& v) {
    float x_ = p.r_kinv[0] * x + p.r_kinv[1] * y + p.r_kinv[2];
    float y_ = p.r_kinv[3] * x + p.r_kinv[4] * y + p.r_kinv[5];
    float z_ = p.r_kinv[6] * x + p.r_kinv[7] * y + p.r_kinv[8];
    u = p.scale * atan2f(x_, z_);
    if (p.kind == ISX_WARP_CYLINDRICAL) v = p.scale * y_ / sqrtf(x_ * x_ + z_ * z_);
    else {
        float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        v = p.scale * (PI_F - acosf(w == w ? w : 0));
    }
}

// the stand-ins of forward_proxy as functions of u and v (monotone increasing), in double
double proxy_d_of_u(double u, double scale) {
    double th = u / scale;
    const double pi = 3.14159265358979323846;
    th = std::max(-pi, std::min(pi, th));
    double sx = std::sin(th), cz = std::cos(th);
    double t = std::fabs(sx) / (std::fabs(sx) + std::fabs(cz));
    double dd = cz >= 0.0 ? t : 2.0 - t;
    return th < 0.0 ? -dd : dd;
}
double proxy_q_of_v(double v, double scale) { return v / scale; }
// spherical: v = scale (pi - acos w)  =>  w = cos(pi - v / scale) = -cos(v / scale), increasing on [0, pi scale]
double proxy_w_of_v(double v, double scale) {
    const double pi = 3.14159265358979323846;
    const double a = std::max(0.0, std::min(pi, v / scale));
    return -std::cos(a);
}

// SphericalWarper::detectResultRoi's pole tests (OpenCV warpers.cpp): is the projection's north (v = pi scale) / south (v = 0) pole
// inside the source image?  Plain arithmetic on K and R^T, no transcendental: always on the host.
void sph_poles(const float k[9], const float rinv[9], int sw, int sh, bool* north, bool* south, float margin = 0.f) {
    *north = *south = false;
    float x = rinv[1], y = rinv[4], z = rinv[7];
    if (y > 0.f) {
        float x_ = (k[0] * x + k[1] * y) / z + k[2], y_ = k[4] * y / z + k[5];
        if (x_ > -margin && x_ < sw + margin && y_ > -margin && y_ < sh + margin) *north = true;
    }
    x = rinv[1]; y = -rinv[4]; z = rinv[7];
    if (y > 0.f) {
        float x_ = (k[0] * x + k[1] * y) / z + k[2], y_ = k[4] * y / z + k[5];
        if (x_ > -margin && x_ < sw + margin && y_ > -margin && y_ < sh + margin) *south = true;
    }
}

// are the extrema of the cylindrical mapForward over the sw x sh image attained on its border?  (see flush_verify)
bool cyl_extrema_on_border(const Proj& p, const float k[9], const float rinv[9], int sw, int sh) {
    const float xs[2] = {0.f, (float)(sw - 1)}, ys[2] = {0.f, (float)(sh - 1)};
    for (float x : xs)
        for (float y : ys) {
            const float z_ = p.r_kinv[6] * x + p.r_kinv[7] * y + p.r_kinv[8];
            if (!(z_ > 1e-6f)) return false;
        }
    // sph_poles reproduces OpenCV's SphericalWarper quirk - its south-pole test negates only y - so for rinv[4] < 0 (a camera pitched past
    // the zenith or rolled by more than a right angle) it tests the mirror image of the axis' true projection about cy, and with an
    // off-centre principal point a pole inside the image could be reported outside.  The proof below is only used where the quirk cannot
    // matter; everything else keeps the reference's full scan (W:72-81).
    if (!(rinv[4] > 0.f)) return false;
    bool north, south;
    sph_poles(k, rinv, sw, sh, &north, &south, 2.f);         // (a pole within two pixels of the border counts as inside)
    return !north && !south;
}

// static_cast<int>(extremum) == bound  <=>  extremum in (bound - 1, bound] / [bound, bound + 1) / (-1, 1)
void trunc_interval(int bound, double& lo, double& hi) {
    if (bound < 0) { lo = bound - 1.0; hi = bound; }
    else if (bound > 0) { lo = bound; hi = bound + 1.0; }
    else { lo = -1.0; hi = 1.0; }
}

}  // namespace

struct isx_warper {
    int kind = 0, device = 0;
    float scale = 1.f;
    hipStream_t stream = nullptr;
    // device scratch
    DevBuf tabs, scan;       // tables; {keys[4], count, mismatches, cand...}
    // planned (sync-free) warps run their ROI scan + device-side check on a side stream: the scan is
    // VALU-bound (atan2f / sqrt / div per source pixel) and overlaps with the memory-bound kernels of the
    // main stream; nothing on the main stream depends on it (isx_warper_plan_status joins both).
    hipStream_t side = nullptr;
    hipEvent_t ev_warp = nullptr;   // recorded on the main stream after a planned warp: its scan starts behind it
    hipEvent_t ev_scan = nullptr;   // recorded on the side stream after the check: isx_warper_join waits on it
    DevBuf scan_side;        // {keys[4], count, mismatches} used on the side stream only
    DevBuf scan_blk;         // the synchronous scan's per-block extrema (k_roi_scan -> k_roi_candidates)
    void* pin = nullptr;     // pinned host landing zone of detectResultRoi's {keys, count, first candidates}
    double gain = 1.0;       // isx_warper_set_gain: folded into the fused tile warp's store (1.0 = off)
    unsigned char gain_lut[256] = {};   // its 256-entry table (handed to the kernel in its arguments)
    int verify_dropped = 0;  // verifications discarded under ISX_VERIFY_NEVER (isx_warper_plan_status reports them)
    RoiPin* pin2 = nullptr;  // pinned block the border scan writes its answer into (k_roi_border_pin); pin_seq: the call number it publishes
    int pin_seq = 0;
    // isx_warper_set_deferred_verify: planned warps queue their verification; isx_warper_verify enqueues the
    // queued scans behind the main stream's position AT THAT CALL (e.g. after the last warp of a step, so
    // that they run under the memory-bound pyramid kernels instead of under the next tile's warp)
    struct Pending { Proj proj; int sw, sh; int planned[4]; float k[9], rinv[9]; };
    std::vector<Pending> pending;
    bool defer_verify = false;
    MatStage st_src, st_mask, st_dst, st_dmask, st_x, st_y;
    // cache of mapBackward tables, one entry per (kind, scale, roi): a rig's tiles alternate between a few ROIs
    // (round 5: 1024 entries, each with its own host copy.  With 16 entries and one shared host buffer a panorama of more than 16 tiles through
    // one handle missed on EVERY warp - two stream synchronisations and 7 000 sinf / cosf per call: the 64-tile step was host-bound at 68 us per
    // 27 us warp kernel, 9.8 ms against a kernel sum of 7.3, profiles/round5_many_tiles_trace.txt)
    struct TabEntry { int kind; float scale; int roi[4]; const float* dev; };
    std::vector<TabEntry> tab_cache;
    std::unordered_multimap<unsigned long long, size_t> tab_index;   // hash of (kind, scale, roi) -> entry (no linear scan of 1024 entries per warp)
    // the tables live in a few 4 MiB chunks, bump-allocated (a DevBuf per entry was a megabyte per entry - reserve() rounds up - i.e. up to a
    // gigabyte per warper at the old cap of 1024 entries): at most TAB_BYTES of them; a full arena is drained and started over
    std::vector<std::unique_ptr<DevBuf>> tab_chunks;
    size_t tab_chunk = 0, tab_used = 0;    // the chunk being filled and the bytes used in it
    long long tab_resets = 0;
    std::vector<float> tab_host;           // staging of the one table being built (its upload is waited for)
    std::vector<int> host_cand;
    std::vector<float> host_scratch;      // the host border scan's stand-ins (roihost.cpp)
    // isx_warper_begin_batch .. isx_warper_end_batch: the fused tile warps in between are collected and leave as ONE launch (k_warp_tile_batch)
    struct BatchItem { int variant; WarpTileGeom g; unsigned gx, gy; double bytes; };
    bool batching = false;
    std::vector<BatchItem> batch;
    float k[9], rinv[9];
    Proj proj;
    hipStream_t roi_stream = nullptr;  // the synchronous ROI scans' stream (roi_stream_of)
    int col0 = 0, col1 = 0;            // isx_warper_set_dst_columns: the warped tile's columns the next fused warps produce (0, 0 = all)
    // isx_warper_set_roi_cache: detectResultRoi is a pure function of (projection, source size); a fixed rig asks for the
    // same few again and again.  Opt-in: remembered results are returned without the scan and its host round trip.
    struct RoiEntry { Proj proj; int sw, sh; int roi[4]; float mm[4]; float k[9], rinv[9]; };
    std::vector<RoiEntry> roi_cache;
    bool roi_cache_on = false;
    // The spherical ROI is a scan of the source's border on the HOST (0.3 ms for an 8K tile): a pure function with no device work and
    // no synchronisation to preserve, so its results are always remembered (per projection and source size, 1024 entries).
    std::vector<RoiEntry> sph_memo;
};

namespace {

constexpr int CAND_CAP = 1 << 16;
constexpr int CAND_FIRST = 1024;   // candidates that travel with the count in the one copy of detectResultRoi
static_assert(CAND_FIRST == PIN_CAND, "k_roi_border_pin's pinned block holds CAND_FIRST candidates");

int set_camera(isx_warper* w, const float K[9], const float R[9]) {
    ISX_CHECK_ARG(K != nullptr && R != nullptr, ISX_ERR_INVALID, "setCameraParams: K and R must be 3x3 CV_32F (got null)");  // W:94-95
    for (int i = 0; i < 9; ++i) {
        ISX_CHECK_ARG(std::isfinite(K[i]) && std::isfinite(R[i]), ISX_ERR_INVALID, "setCameraParams: K / R contain non-finite values");
        w->k[i] = K[i];                                                     // W:98-101
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) w->rinv[i * 3 + j] = R[j * 3 + i];   // W:103
    float kinv[9];
    mat3_inv(K, kinv);
    mat3_mul(R, kinv, w->proj.r_kinv);                                      // W:108
    mat3_mul(K, w->rinv, w->proj.k_rinv);                                   // W:113
    w->proj.scale = w->scale;
    w->proj.kind = w->kind;
    return ISX_OK;
}

// The collected tile warps of a batch (isx_warper_begin_batch): runs of the same kernel variant leave as one launch of up to WARP_BATCH_MAX tiles
// (blockIdx.z = tile, the grid the largest tile's); a run of one takes the ordinary kernel.
template <int KD, bool O16, bool V>
int launch_warp_batch(hipStream_t st, const isx_warper::BatchItem* it, int n) {
    if (n == 1) {
        WarpTileArgs a{it[0].g.p, it[0].g.t, it[0].g.img, it[0].g.d, {}};
        ISX_LAUNCH("warp_tile", it[0].bytes, st, (k_warp_tile<KD, O16, V>), dim3(it[0].gx, it[0].gy), dim3(64 * WARP_WAVES), 0, a);
        return ISX_OK;
    }
    WarpTileBatch b;
    memset(&b, 0, sizeof(b));
    unsigned gx = 0, gy = 0;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        b.a[i] = it[i].g;
        b.a[i].d.xg = 0;          // (the XCD-run order is a function of the launch's own grid: off in a shared one)
        gx = std::max(gx, it[i].gx); gy = std::max(gy, it[i].gy); bytes += it[i].bytes;
    }
    ISX_LAUNCH("warp_tile", bytes, st, (k_warp_tile_batch<KD, O16, V>), dim3(gx, gy, (unsigned)n), dim3(64 * WARP_WAVES), 0, b);
    return ISX_OK;
}
int flush_warp_batch(isx_warper* w) {
    if (w->batch.empty()) return ISX_OK;
    std::vector<isx_warper::BatchItem> items;
    items.swap(w->batch);
    for (size_t i = 0; i < items.size();) {
        size_t j = i + 1;
        while (j < items.size() && j - i < (size_t)WARP_BATCH_MAX && items[j].variant == items[i].variant) ++j;
        const isx_warper::BatchItem* it = &items[i];
        const int n = (int)(j - i), v = items[i].variant;
        int rc = ISX_OK;
        switch (v) {
            case 0: rc = launch_warp_batch<ISX_WARP_CYLINDRICAL, false, false>(w->stream, it, n); break;
            case 1: rc = launch_warp_batch<ISX_WARP_CYLINDRICAL, false, true>(w->stream, it, n); break;
            case 2: rc = launch_warp_batch<ISX_WARP_CYLINDRICAL, true, false>(w->stream, it, n); break;
            case 3: rc = launch_warp_batch<ISX_WARP_CYLINDRICAL, true, true>(w->stream, it, n); break;
            case 4: rc = launch_warp_batch<ISX_WARP_SPHERICAL, false, false>(w->stream, it, n); break;
            case 5: rc = launch_warp_batch<ISX_WARP_SPHERICAL, false, true>(w->stream, it, n); break;
            case 6: rc = launch_warp_batch<ISX_WARP_SPHERICAL, true, false>(w->stream, it, n); break;
            default: rc = launch_warp_batch<ISX_WARP_SPHERICAL, true, true>(w->stream, it, n); break;
        }
        if (rc != ISX_OK) return rc;
        i = j;
    }
    return ISX_OK;
}

// Enqueue the queued verification scans of planned warps on the side stream, behind the main stream's
// current position.  The scan is VALU-bound like the warp kernel: it should run under memory-bound work.
int flush_verify(isx_warper* w, hipEvent_t after = nullptr) {
    ISX_TRY(flush_warp_batch(w));      // (the scans start behind the warps they verify)
    if (w->pending.empty()) return ISX_OK;
    // ISX_VERIFY_NEVER: a measurement aid (what the verification scans cost a step).  A run under it is not a verified run and cannot pass
    // for one: isx_warper_plan_status answers ISX_ERR_PLAN once a verification has been dropped here.
    static const bool never = getenv("ISX_VERIFY_NEVER") != nullptr;
    if (never) { w->verify_dropped += (int)w->pending.size(); w->pending.clear(); return ISX_OK; }
    hipStream_t st = w->stream;
    if (!w->side) {
        // one verification stream per DEVICE, shared by every warper on it (never destroyed): a batch of pairs would otherwise
        // bring one stream per warper and run out of hardware queues (16 pairs: 44 instead of 51 Gpix/s)
        static std::mutex mu;
        static hipStream_t shared[64] = {};
        {
            std::lock_guard<std::mutex> lk(mu);
            const int d = w->device >= 0 && w->device < 64 ? w->device : 0;
            if (!shared[d]) ISX_HIP(hipStreamCreateWithFlags(&shared[d], hipStreamNonBlocking));
            w->side = shared[d];
        }
        ISX_HIP(hipEventCreateWithFlags(&w->ev_warp, hipEventDisableTiming));
        ISX_HIP(hipEventCreateWithFlags(&w->ev_scan, hipEventDisableTiming));
    }
    // The scans read the projection and the source size, never an image: nothing they need comes from the main stream.  The event only
    // PLACES them (full scans are VALU-bound and should start under memory-bound work).  A border scan is one workgroup: it starts right
    // away, with no event on the main stream at all - unless that stream is being captured, where the wait is what forks the side
    // stream into the graph.
    static const bool ordered_always = getenv("ISX_VERIFY_ORDERED") != nullptr;      // A/B aid: the placed form for every scan
    bool cheap = !ordered_always;
    for (const isx_warper::Pending& pd : w->pending)
        cheap = cheap && (pd.proj.kind == ISX_WARP_SPHERICAL || cyl_extrema_on_border(pd.proj, pd.k, pd.rinv, pd.sw, pd.sh));
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (cheap) ISX_HIP(hipStreamIsCapturing(st, &cs));
    if (cheap && cs == hipStreamCaptureStatusNone) {
        // nothing to wait for
    } else if (after) ISX_HIP(hipStreamWaitEvent(w->side, after, 0));
    else {
        ISX_HIP(hipEventRecord(w->ev_warp, st));
        ISX_HIP(hipStreamWaitEvent(w->side, w->ev_warp, 0));
    }
    if (!w->scan_side.p) {
        ISX_TRY(w->scan_side.reserve(64));
        ISX_HIP(hipMemsetAsync(w->scan_side.p, 0, 64, w->side));
        ISX_HIP(hipMemsetAsync(w->scan_side.p, 0xff, 2 * sizeof(unsigned), w->side));
    }
    unsigned* sk = (unsigned*)w->scan_side.p;
    for (const isx_warper::Pending& pd : w->pending) {
        const bool sph = pd.proj.kind == ISX_WARP_SPHERICAL;
        bool north = false, south = false;
        if (sph) {
            ISX_LAUNCH("roi_border_sph", 0.0, w->side, k_roi_border_sph, dim3(1), dim3(1024), 0, pd.proj, pd.sw, pd.sh, sk, (int*)nullptr, 0, (int*)nullptr);
            sph_poles(pd.k, pd.rinv, pd.sw, pd.sh, &north, &south);
        } else if (cyl_extrema_on_border(pd.proj, pd.k, pd.rinv, pd.sw, pd.sh)) {
            // The verification of a PLANNED cylindrical ROI (a guard against a stale plan; the ROI itself came from the reference's full
            // scan, W:64-88): with the whole image in front of the camera (z_ > 0 at the four corners, z_ is linear in x, y) u = atan2(x_, z_)
            // is monotone along every line of the image, and v = y_ / |(x_, z_)| is the tangent of the latitude, which has no local
            // extremum on a sphere away from its poles - neither of which lies inside the image.  Both extrema are then attained on the
            // image's border: 2 (W + H) points in one workgroup instead of a 14 us scan of every source pixel beside the blend's kernels.
            ISX_LAUNCH("roi_border", 0.0, w->side, k_roi_border_sph, dim3(1), dim3(1024), 0, pd.proj, pd.sw, pd.sh, sk, (int*)nullptr, 0, (int*)nullptr);
        } else {
            dim3 sgrid(cdiv(pd.sw, 256), cdiv(pd.sh, ROI_ROWS));   // ~8 K waves at 4K: one full-occupancy round
            ISX_LAUNCH("roi_scan", 0.0, w->side, k_roi_scan, sgrid, dim3(256), 0, pd.proj, pd.sw, pd.sh, sk, ROI_ROWS, (float4*)nullptr);
        }
        RoiBounds rb;
        for (int k = 0; k < 4; ++k) {
            double lo, hi;
            trunc_interval(pd.planned[k], lo, hi);
            const bool is_u = (k == 0 || k == 2);
            double plo = is_u ? proxy_d_of_u(lo, pd.proj.scale) : (sph ? proxy_w_of_v(lo, pd.proj.scale) : proxy_q_of_v(lo, pd.proj.scale));
            double phi = is_u ? proxy_d_of_u(hi, pd.proj.scale) : (sph ? proxy_w_of_v(hi, pd.proj.scale) : proxy_q_of_v(hi, pd.proj.scale));
            if (sph) {
                // a pole inside the image replaces the border's extremum by its own value (min / max with 0 or pi scale): where the planned
                // bound IS the pole's, the border's extremum only has to lie on the border's side of it
                const double pv = 3.14159265358979323846 * pd.proj.scale;
                const double pole[2][4] = {{0.0, pv, 0.0, pv}, {0.0, 0.0, 0.0, 0.0}};
                for (int q = 0; q < 2; ++q)
                    if ((q == 0 ? north : south) && f2i_host((float)pole[q][k]) == pd.planned[k]) {
                        if (k < 2) phi = is_u ? 2.0 : 1.0; else plo = is_u ? -2.0 : -1.0;
                    }
            }
            // margin: the stand-ins carry a few ulp of error; an extremum this close to an integer boundary is
            // not flagged (the check is a guard against a stale plan, not a proof)
            const double m = 4e-6 * std::max(1.0, std::max(std::fabs(plo), std::fabs(phi)));
            rb.lo[k] = (float)(plo - m); rb.hi[k] = (float)(phi + m);
        }
        ISX_LAUNCH("roi_check", 0.0, w->side, k_roi_check_rearm, dim3(1), dim3(1), 0, sk, rb, (int*)(sk + 5));
    }
    ISX_HIP(hipEventRecord(w->ev_scan, w->side));
    w->pending.clear();
    return ISX_OK;
}

// detectResultRoi.  Cylindrical: full scan on the GPU (W:72-81) + host refinement of u.
// Spherical: OpenCV's detectResultRoiByBorder + pole tests — O(W+H) points, evaluated on the host.
// The stream the synchronous ROI scans run on: one per device, shared by every warper on it (see flush_verify on why not one each).
// detectResultRoi is a function of the projection and the source SIZE - it reads no image - so it need not queue behind whatever
// the handle's stream still holds (the previous blend, typically): the host gets its corner after the scan alone, W:160 is kept,
// and the strict call sequence stops being bound by one full GPU drain per warp.
int roi_stream_of(isx_warper* w, hipStream_t* out) {
    if (!w->roi_stream) {
        static std::mutex mu;
        static hipStream_t shared[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        const int d = w->device >= 0 && w->device < 64 ? w->device : 0;
        if (!shared[d]) ISX_HIP(hipStreamCreateWithFlags(&shared[d], hipStreamNonBlocking));      // (a high-priority stream changes nothing here: a resident wave is not preempted; measured 40 us per call beside a saturating kernel either way)
        w->roi_stream = shared[d];
    }
    *out = w->roi_stream;
    return ISX_OK;
}

// The synchronous border scan (k_roi_border_pin) on the ROI stream, its candidates in w->host_cand when this returns: one launch, then the
// caller's thread polls the sequence number the kernel publishes in pinned memory (ISX_ROI_POLL=0: the round-3 form - kernel, copy, re-arm
// kernel, hipStreamSynchronize - for A/B runs).
// detectResultRoi's answer from the candidates of a scan: mapForward with the host's libm on exactly those (W:72-86; spherical:
// detectResultRoiByBorder's truncation of the border's extrema, then OpenCV's two pole tests)
void roi_from_candidates(const Proj& p, const float k[9], const float rinv[9], int sw, int sh, const int* cand, int n, int roi[4], float mm_out[4]) {
    float tl_u = std::numeric_limits<float>::max(), tl_v = tl_u, br_u = -tl_u, br_v = -tl_u;     // W:66-69
    for (int i = 0; i < n; ++i) {
        float u, v;
        map_forward_host(p, (float)cand[2 * i], (float)cand[2 * i + 1], u, v);
        tl_u = (std::min)(tl_u, u); tl_v = (std::min)(tl_v, v);                                  // W:77-78
        br_u = (std::max)(br_u, u); br_v = (std::max)(br_v, v);
    }
    if (p.kind == ISX_WARP_SPHERICAL) {
        tl_u = (float)f2i_host(tl_u); tl_v = (float)f2i_host(tl_v); br_u = (float)f2i_host(br_u); br_v = (float)f2i_host(br_v);
        bool north, south;
        sph_poles(k, rinv, sw, sh, &north, &south);
        if (north) {
            const float pv = (float)(3.1415926535897932384626433832795 * p.scale);
            tl_u = (std::min)(tl_u, 0.f); tl_v = (std::min)(tl_v, pv); br_u = (std::max)(br_u, 0.f); br_v = (std::max)(br_v, pv);
        }
        if (south) { tl_u = (std::min)(tl_u, 0.f); tl_v = (std::min)(tl_v, 0.f); br_u = (std::max)(br_u, 0.f); br_v = (std::max)(br_v, 0.f); }
    }
    mm_out[0] = tl_u; mm_out[1] = tl_v; mm_out[2] = br_u; mm_out[3] = br_v;
    roi[0] = f2i_host(tl_u); roi[1] = f2i_host(tl_v); roi[2] = f2i_host(br_u); roi[3] = f2i_host(br_v);   // W:83-86
}

// Round 6: the ranking runs on the caller's thread (roihost.cpp: no launch, nothing to queue behind, nothing to wait for - the scan reads the
// projection and the source size only); ISX_ROI_HOST=0 restores the device forms below for A/B runs and for their tests.
int border_scan_host(const float* r_kinv, bool sph, int sw, int sh, std::vector<int>& cand_out, std::vector<float>& scratch, int isa, int* n_out) {
    scratch.resize((size_t)4 * ((size_t)sw + (size_t)sh));
    if (cand_out.size() < (size_t)2 * CAND_FIRST) cand_out.resize((size_t)2 * CAND_FIRST);
    int n = isx_roi_border_host(r_kinv, sph ? 1 : 0, sw, sh, cand_out.data(), (int)(cand_out.size() / 2), scratch.data(), isa);
    ISX_CHECK_ARG(n <= CAND_CAP, ISX_ERR_UNSUPPORTED, "detectResultRoi: %d extremum candidates exceed the refinement buffer (%d)", n, CAND_CAP);
    if ((size_t)n > cand_out.size() / 2) {       // rare: more candidates than the list held
        cand_out.resize((size_t)2 * n);
        n = isx_roi_border_host(r_kinv, sph ? 1 : 0, sw, sh, cand_out.data(), n, scratch.data(), isa);
    }
    ISX_CHECK_ARG(n > 0, ISX_ERR_INVALID, "detectResultRoi: mapForward is not finite anywhere on the border of the %d x %d source (bad K / R / scale?)", sw, sh);
    cand_out.resize((size_t)n * 2);
    *n_out = n;
    return ISX_OK;
}

int border_scan_sync(isx_warper* w, int sw, int sh, hipStream_t st, const char* label, int* n_out) {
    static const bool host_scan = [] { const char* e = getenv("ISX_ROI_HOST"); return !(e && e[0] == '0'); }();
    if (host_scan) return border_scan_host(w->proj.r_kinv, w->kind == ISX_WARP_SPHERICAL, sw, sh, w->host_cand, w->host_scratch, 0, n_out);
    static const bool poll = [] { const char* e = getenv("ISX_ROI_POLL"); return !(e && e[0] == '0'); }();
    unsigned* keys = (unsigned*)w->scan.p;
    int* count = (int*)(keys + 4);
    int* cand = (int*)((char*)w->scan.p + 64);
    int n = 0;
    const int* first = nullptr;
    if (poll) {
        if (!w->pin2) {
            ISX_HIP(hipHostMalloc((void**)&w->pin2, sizeof(RoiPin), hipHostMallocCoherent | hipHostMallocMapped));
            memset(w->pin2, 0, sizeof(RoiPin));
        }
        const int seq = ++w->pin_seq;
        ISX_LAUNCH(label, 0.0, st, k_roi_border_pin, dim3(1), dim3(1024), 0, w->proj, sw, sh, w->pin2, seq, cand, CAND_CAP);
        (void)hipStreamQuery(st);           // the dispatch is on its way before the polling starts
        const auto t0 = std::chrono::steady_clock::now();
        bool synced = false;
        while (__atomic_load_n(&w->pin2->seq, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
            if (!synced && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {   // a busy GPU: wait the ordinary way (and report its errors)
                ISX_HIP(hipStreamSynchronize(st));
                synced = true;
                ISX_CHECK_ARG(__atomic_load_n(&w->pin2->seq, __ATOMIC_ACQUIRE) == seq, ISX_ERR_HIP, "detectResultRoi: the border scan finished without publishing its result");
            }
        }
        n = w->pin2->count;
        first = w->pin2->cand;
    } else {
        ISX_LAUNCH(label, 0.0, st, k_roi_border_sph, dim3(1), dim3(1024), 0, w->proj, sw, sh, (unsigned*)nullptr, cand, CAND_CAP, count);
        if (!w->pin) ISX_HIP(hipHostMalloc(&w->pin, 64 + (size_t)CAND_FIRST * 8, hipHostMallocDefault));
        ISX_HIP(hipMemcpyAsync(w->pin, w->scan.p, 64 + (size_t)CAND_FIRST * 8, hipMemcpyDeviceToHost, st));
        ISX_LAUNCH("roi_rearm", 0.0, st, k_roi_rearm, dim3(1), dim3(1), 0, keys);
        ISX_HIP(hipStreamSynchronize(st));
        n = ((const int*)w->pin)[4];
        first = (const int*)((const char*)w->pin + 64);
    }
    ISX_CHECK_ARG(n <= CAND_CAP, ISX_ERR_UNSUPPORTED, "detectResultRoi: %d extremum candidates exceed the refinement buffer (%d)", n, CAND_CAP);
    ISX_CHECK_ARG(n > 0, ISX_ERR_INVALID, "detectResultRoi: mapForward is not finite anywhere on the border of the %d x %d source (bad K / R / scale?)", sw, sh);
    w->host_cand.resize((size_t)n * 2);
    std::memcpy(w->host_cand.data(), first, (size_t)std::min(n, CAND_FIRST) * 8);
    if (n > CAND_FIRST)     // rare: more candidates than the pinned block carries
        ISX_HIP(hipMemcpy(w->host_cand.data() + 2 * (size_t)CAND_FIRST, cand + 2 * (size_t)CAND_FIRST, (size_t)(n - CAND_FIRST) * 8, hipMemcpyDeviceToHost));
    *n_out = n;
    return ISX_OK;
}

int detect_roi(isx_warper* w, int sw, int sh, int roi[4], float mm[4], bool sync_free, const int* planned) {
    hipStream_t st = w->stream;
    if (!sync_free) ISX_TRY(roi_stream_of(w, &st));
    size_t need = 64 + (size_t)CAND_CAP * 8;
    if (!w->scan.p) {   // keys armed once here; every consumer re-arms them (k_warp_img_mask / k_roi_rearm)
        ISX_TRY(w->scan.reserve(need));
        ISX_HIP(hipMemsetAsync(w->scan.p, 0, 64, st));
        ISX_HIP(hipMemsetAsync(w->scan.p, 0xff, 2 * sizeof(unsigned), st));
    }
    unsigned* keys = (unsigned*)w->scan.p;
    int* count = (int*)(keys + 4);
    int* cand = (int*)((char*)w->scan.p + 64);
    if (w->kind == ISX_WARP_SPHERICAL && !sync_free) {
        for (const auto& e : w->sph_memo)
            if (e.sw == sw && e.sh == sh && memcmp(&e.proj, &w->proj, sizeof(Proj)) == 0 && memcmp(e.k, w->k, sizeof(e.k)) == 0 && memcmp(e.rinv, w->rinv, sizeof(e.rinv)) == 0) {
                std::copy(e.roi, e.roi + 4, roi);
                if (mm) std::copy(e.mm, e.mm + 4, mm);
                return ISX_OK;
            }
        // detectResultRoiByBorder: the border pixels ranked on the device, the candidates for the four extrema evaluated here with the
        // host's libm (exactly the values the host-only scan of round 2 took its minima / maxima over), then OpenCV's two pole tests
        ISX_TRY(roi_stream_of(w, &st));
        int n = 0;
        ISX_TRY(border_scan_sync(w, sw, sh, st, "roi_border_sph", &n));
        float emm[4];
        roi_from_candidates(w->proj, w->k, w->rinv, sw, sh, w->host_cand.data(), n, roi, emm);
        if (mm) std::copy(emm, emm + 4, mm);
        const float tl_u = emm[0], tl_v = emm[1], br_u = emm[2], br_v = emm[3];
        if (w->sph_memo.size() >= (w->roi_cache_on ? (size_t)1024 : (size_t)64)) w->sph_memo.erase(w->sph_memo.begin());   // (1024 only with the fixed-rig hint: a miss scans the list)
        isx_warper::RoiEntry e;
        e.proj = w->proj; e.sw = sw; e.sh = sh;
        std::copy(w->k, w->k + 9, e.k); std::copy(w->rinv, w->rinv + 9, e.rinv);
        std::copy(roi, roi + 4, e.roi);
        e.mm[0] = tl_u; e.mm[1] = tl_v; e.mm[2] = br_u; e.mm[3] = br_v;
        w->sph_memo.push_back(e);
        return ISX_OK;
    }
    // (without isx_warper_set_roi_cache the list holds the LAST result only: the reference asks for the same ROI twice in a row - warp(img, K, R)
    // then warp(mask, K, R), W:229,232 - and the second answer is the first one's, a pure function of the same arguments)
    if (!sync_free)
        for (const auto& e : w->roi_cache)
            if (e.sw == sw && e.sh == sh && memcmp(&e.proj, &w->proj, sizeof(Proj)) == 0) {
                std::copy(e.roi, e.roi + 4, roi);
                if (mm) std::copy(e.mm, e.mm + 4, mm);
                return ISX_OK;
            }
    if (sync_free) {
        isx_warper::Pending pd;
        pd.proj = w->proj; pd.sw = sw; pd.sh = sh;
        std::copy(planned, planned + 4, pd.planned);
        std::copy(w->k, w->k + 9, pd.k); std::copy(w->rinv, w->rinv + 9, pd.rinv);
        w->pending.push_back(pd);
        if (!w->defer_verify && !w->batching) return flush_verify(w);      // (a collecting batch: its verifications start when it ends)
        return ISX_OK;
    }
    // cylindrical: min keys start at 0xffffffff, max keys and the candidate count at 0 (armed by the
    // previous consumer); the mismatch counter keys[5] is sticky
    // Round 3: where both extrema of u and v provably lie on the image's border (cyl_extrema_on_border: the image in front of the camera, no
    // pole of the cylinder within two pixels of it) the scan is the 2 (W + H) border pixels in one workgroup instead of all W x H: away from
    // the poles neither u nor v has a stationary point, so a pixel one step inside the border differs from the border's extremum by about a
    // whole unit (|grad| ~ scale / focal per pixel), four orders of magnitude above the rounding of mapForward - the extrema over the border
    // ARE the extrema over the image, and the candidates below are evaluated exactly as before.  (ISX_ROI_FULL_SCAN: the full scan always.)
    static const bool full_always = getenv("ISX_ROI_FULL_SCAN") != nullptr;
    const bool border_only = !full_always && cyl_extrema_on_border(w->proj, w->k, w->rinv, sw, sh);
    dim3 grid(cdiv(sw, 256), cdiv(sh, SYNC_ROWS));
    float4* blk = nullptr;
    int n = 0;
    if (border_only) {
        ISX_TRY(border_scan_sync(w, sw, sh, st, "roi_border", &n));
    } else {
    ISX_TRY(w->scan_blk.reserve((size_t)grid.x * grid.y * sizeof(float4)));
    blk = (float4*)w->scan_blk.p;
    ISX_LAUNCH("roi_scan", 0.0, st, k_roi_scan, grid, dim3(256), 0, w->proj, sw, sh, (unsigned*)nullptr, SYNC_ROWS, blk);
    // The scan ranked the pixels by the stand-ins (d, q).  Collect every pixel whose stand-in is within
    // a tolerance of one of the four extrema and evaluate mapForward on exactly those with the host's
    // libm: the result is what the reference code computes on this host.  Scan, candidate pass, the copy of
    // {count, first candidates} into pinned memory and the re-arming of the keys are enqueued back to back:
    // ONE stream synchronisation per detectResultRoi (the corner must reach the host, W:148-150,160).
    static_assert(SYNC_ROWS % CAND_ROWS == 0, "a candidate block lies inside one scan block");
    ISX_LAUNCH("roi_candidates", 0.0, st, k_roi_candidates, dim3(grid.x, cdiv(sh, CAND_ROWS)), dim3(256), 0, w->proj, sw, sh, cand, CAND_CAP, count, CAND_ROWS,
               (const float4*)blk, SYNC_ROWS, (int)grid.y);
    if (!w->pin) ISX_HIP(hipHostMalloc(&w->pin, 64 + (size_t)CAND_FIRST * 8, hipHostMallocDefault));
    ISX_HIP(hipMemcpyAsync(w->pin, w->scan.p, 64 + (size_t)CAND_FIRST * 8, hipMemcpyDeviceToHost, st));
    ISX_LAUNCH("roi_rearm", 0.0, st, k_roi_rearm, dim3(1), dim3(1), 0, keys);
    ISX_HIP(hipStreamSynchronize(st));
    n = ((const int*)w->pin)[4];
    ISX_CHECK_ARG(n <= CAND_CAP, ISX_ERR_UNSUPPORTED, "detectResultRoi: %d extremum candidates exceed the refinement buffer (%d)", n, CAND_CAP);
    ISX_CHECK_ARG(n > 0, ISX_ERR_INVALID, "detectResultRoi: mapForward is not finite anywhere on the %d x %d source (bad K / R / scale?)", sw, sh);
    w->host_cand.resize((size_t)n * 2);
    std::memcpy(w->host_cand.data(), (const char*)w->pin + 64, (size_t)std::min(n, CAND_FIRST) * 8);
    if (n > CAND_FIRST)   // rare: more candidates than the first copy carried
        ISX_HIP(hipMemcpy(w->host_cand.data() + 2 * (size_t)CAND_FIRST, cand + 2 * (size_t)CAND_FIRST, (size_t)(n - CAND_FIRST) * 8, hipMemcpyDeviceToHost));
    }
    float emm[4];
    roi_from_candidates(w->proj, w->k, w->rinv, sw, sh, w->host_cand.data(), n, roi, emm);
    if (mm) std::copy(emm, emm + 4, mm);
    const float tl_uf = emm[0], tl_vf = emm[1], br_uf = emm[2], br_vf = emm[3];
    {
        const size_t cap = w->roi_cache_on ? 1024 : 1;
        while (w->roi_cache.size() >= cap) w->roi_cache.erase(w->roi_cache.begin());
        isx_warper::RoiEntry e;
        memset(&e, 0, sizeof(e));
        e.proj = w->proj; e.sw = sw; e.sh = sh;
        std::copy(roi, roi + 4, e.roi);
        e.mm[0] = tl_uf; e.mm[1] = tl_vf; e.mm[2] = br_uf; e.mm[3] = br_vf;
        w->roi_cache.push_back(e);
    }
    return ISX_OK;
}

// per-column / per-row tables of mapBackward's transcendental part, cached per (kind, scale, roi)
int make_tabs(isx_warper* w, const int roi[4], MapTabs* t) {
    int mw = roi[2] - roi[0] + 1, mh = roi[3] - roi[1] + 1;
    const int mwp = (mw + 3) & ~3, mhp = (mh + 3) & ~3;   // segments padded to 16 bytes: the fused kernel loads float4
    size_t n = (size_t)2 * mwp + 2 * mhp;
    // A fixed rig (the reference, every BASELINE config) asks for the same few ROIs for ever: up to 1024 of them stay (a 64-tile panorama
    // through one handle needs 64; with 16 every planned warp recomputed 7 000 sinf / cosf, DESIGN.md §3 "Round 5").  A caller whose camera
    // changes with every frame never hits: for it the cache is bounded by BYTES (TAB_BYTES of tables, ~350 ROIs of 4K tiles) and looked up
    // through a hash, not scanned; when the arena or the entry list is full the stream is drained once and the cache starts over.
    constexpr size_t TAB_SLOTS = 1024, TAB_BYTES = (size_t)16 << 20, TAB_CHUNK = (size_t)4 << 20;
    unsigned long long key = 1469598103934665603ull;
    {
        unsigned sbits; memcpy(&sbits, &w->scale, 4);
        const unsigned words[6] = {(unsigned)w->kind, sbits, (unsigned)roi[0], (unsigned)roi[1], (unsigned)roi[2], (unsigned)roi[3]};
        for (unsigned v : words) { key ^= v; key *= 1099511628211ull; }
    }
    isx_warper::TabEntry* e = nullptr;
    {
        auto r = w->tab_index.equal_range(key);
        for (auto it = r.first; it != r.second; ++it) {
            isx_warper::TabEntry& c = w->tab_cache[it->second];
            if (c.kind == w->kind && c.scale == w->scale && std::equal(roi, roi + 4, c.roi)) { e = &c; break; }
        }
    }
    if (!e) {
        const size_t need = (n * sizeof(float) + 255) & ~(size_t)255;
        ISX_CHECK_ARG(need <= TAB_CHUNK, ISX_ERR_UNSUPPORTED, "warp: a %d x %d warped tile needs %zu bytes of column / row tables (limit %zu)", mw, mh, need, TAB_CHUNK);
        auto start_over = [&]() -> int {      // enqueued kernels (and collected ones: isx_warper_begin_batch) may still read the old tables: drain first
            ISX_TRY(flush_warp_batch(w));
            ISX_HIP(hipStreamSynchronize(w->stream));
            w->tab_cache.clear(); w->tab_index.clear(); w->tab_chunk = 0; w->tab_used = 0; ++w->tab_resets;
            return ISX_OK;
        };
        if (w->tab_cache.size() >= TAB_SLOTS) ISX_TRY(start_over());
        if (!w->tab_chunks.empty() && w->tab_used + need > TAB_CHUNK) {
            if ((w->tab_chunk + 2) * TAB_CHUNK > TAB_BYTES) ISX_TRY(start_over());
            else { ++w->tab_chunk; w->tab_used = 0; }
        }
        if (w->tab_chunk >= w->tab_chunks.size()) {
            w->tab_chunks.emplace_back(new DevBuf());
            ISX_TRY(w->tab_chunks.back()->reserve(TAB_CHUNK));
        }
        float* dev = (float*)((char*)w->tab_chunks[w->tab_chunk]->p + w->tab_used);
        w->tab_host.assign(n, 0.f);
        float* cs = w->tab_host.data(); float* cc = cs + mwp; float* ra = cc + mwp; float* rb = ra + mhp;
        for (int i = 0; i < mw; ++i) {
            float u = (float)(roi[0] + i);
            u /= w->scale;                                 // W:48
            cs[i] = sinf(u); cc[i] = cosf(u);              // W:51,53
        }
        for (int i = 0; i < mh; ++i) {
            float v = (float)(roi[1] + i);
            v /= w->scale;                                 // W:49
            if (w->kind == ISX_WARP_CYLINDRICAL) { ra[i] = v; rb[i] = 0.f; }          // W:52
            else { ra[i] = sinf(PI_F - v); rb[i] = cosf(PI_F - v); }
        }
        // the padding repeats the last column / row: the fused kernel's partial 4-pixel group at the right edge then computes ordinary
        // pixels in its unused lanes (zeros would make z = 0 there and send the whole group down the generic path, once per row)
        for (int i = mw; i < mwp; ++i) { cs[i] = cs[mw - 1]; cc[i] = cc[mw - 1]; }
        for (int i = mh; i < mhp; ++i) { ra[i] = ra[mh - 1]; rb[i] = rb[mh - 1]; }
        ISX_HIP(hipMemcpyAsync(dev, w->tab_host.data(), n * sizeof(float), hipMemcpyHostToDevice, w->stream));
        // A miss is a planning-time event: the upload is simply waited for (so one staging vector serves).  (Tried without the wait in round 5, every
        // entry keeping a host copy alive: the world-8 rehearsal of bench.py then delivered a wrong mosaic in 4 runs of 7 - on this runtime an asynchronous copy
        // from PAGEABLE memory is not something a kernel launched right behind it on the same stream can rely on; 8 of 8 runs pass with the wait.)
        ISX_HIP(hipStreamSynchronize(w->stream));
        w->tab_used += need;
        w->tab_cache.emplace_back();
        e = &w->tab_cache.back();
        e->kind = w->kind; e->scale = w->scale; std::copy(roi, roi + 4, e->roi); e->dev = dev;
        w->tab_index.emplace(key, w->tab_cache.size() - 1);
    }
    const float* base = e->dev;
    t->col_s = base; t->col_c = base + mwp; t->row_a = base + 2 * mwp; t->row_b = base + 2 * mwp + mhp;
    return ISX_OK;
}

int check_roi_sane(const int roi[4]) {
    ISX_CHECK_ARG(roi[2] >= roi[0] && roi[3] >= roi[1] && (long long)roi[2] - roi[0] < 65536 && (long long)roi[3] - roi[1] < 65536,
                  ISX_ERR_INVALID, "detectResultRoi produced a degenerate ROI [%d,%d]-[%d,%d] (bad K / R / scale?)", roi[0], roi[1], roi[2], roi[3]);
    return ISX_OK;
}

int warp_common(isx_warper* w, const isx_mat* src, const isx_mat* src_mask, const float K[9], const float R[9], int interp, int border,
                isx_mat* dst, isx_mat* dst_mask, int corner[2], const int* planned, bool fused, bool verify_plan = true) {
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "warp: null warper");
    ISX_TRY(check_mat(src, "warp: src"));
    ISX_TRY(check_mat(dst, "warp: dst"));
    ISX_HIP(hipSetDevice(w->device));
    ISX_TRY(set_camera(w, K, R));
    int roi[4];
    bool ranged = false;
    int rc0 = 0, rc1 = 0;
    if (planned) std::copy(planned, planned + 4, roi);   // the verifying scan is enqueued behind the warp kernel (below)
    else ISX_TRY(detect_roi(w, src->cols, src->rows, roi, nullptr, false, nullptr));
    ISX_TRY(check_roi_sane(roi));
    const int dw = roi[2] - roi[0] + 1, dh = roi[3] - roi[1] + 1;   // dst.create(roi.height + 1, roi.width + 1)  W:150
    ISX_CHECK_ARG(dst->rows == dh && dst->cols == dw, ISX_ERR_SIZE, "warp: dst is %dx%d, the warped tile is %dx%d (query isx_warper_roi first)",
                  dst->cols, dst->rows, dw, dh);
    MapTabs t;
    ISX_TRY(make_tabs(w, roi, &t));
    hipStream_t st = w->stream;
    ISX_TRY(w->st_src.use_in(src, st, "warp: src"));
    ISX_TRY(w->st_dst.use_out(dst, st, "warp: dst"));
    SrcView sv{(const unsigned char*)w->st_src.d.data, w->st_src.d.step, src->rows, src->cols};
    dim3 grid(cdiv(dw, 64), cdiv(dh, 4));
    double spx = (double)src->rows * src->cols, dpx = (double)dw * dh;
    if (fused) {
        ISX_CHECK_ARG(src->type == ISX_8UC3, ISX_ERR_TYPE, "warp_with_mask: src_img must be CV_8UC3, got %s", type_name(src->type));
        ISX_CHECK_ARG((unsigned long long)w->st_src.d.step * src->rows < (1ull << 31) && w->st_src.d.step < (1u << 24) && src->cols <= 32767 && src->rows <= 32767,
                      ISX_ERR_UNSUPPORTED, "warp_with_mask: source larger than 2 GiB, 16 MiB per row or 32767 pixels per side");
        ISX_CHECK_ARG(dst->type == ISX_8UC3 || dst->type == ISX_16SC3, ISX_ERR_TYPE, "warp_with_mask: dst_img must be CV_8UC3 or CV_16SC3, got %s", type_name(dst->type));
        ISX_TRY(check_mat(dst_mask, "warp_with_mask: dst_mask"));
        ISX_CHECK_ARG(dst_mask->type == ISX_8UC1, ISX_ERR_TYPE, "warp_with_mask: dst_mask must be CV_8U, got %s", type_name(dst_mask->type));
        ISX_CHECK_ARG(dst_mask->rows == dh && dst_mask->cols == dw, ISX_ERR_SIZE, "warp_with_mask: dst_mask is %dx%d, the warped tile is %dx%d",
                      dst_mask->cols, dst_mask->rows, dw, dh);
        SrcView mv{nullptr, 0, src->rows, src->cols};
        if (src_mask) {
            ISX_TRY(check_mat(src_mask, "warp_with_mask: src_mask"));
            ISX_CHECK_ARG(src_mask->type == ISX_8UC1, ISX_ERR_TYPE, "warp_with_mask: src_mask must be CV_8U, got %s", type_name(src_mask->type));
            ISX_CHECK_ARG(src_mask->rows == src->rows && src_mask->cols == src->cols, ISX_ERR_SIZE, "warp_with_mask: src_mask size differs from src_img");
            ISX_TRY(w->st_mask.use_in(src_mask, st, "warp_with_mask: src_mask"));
            mv.data = (const unsigned char*)w->st_mask.d.data; mv.step = w->st_mask.d.step;
        }
        ISX_TRY(w->st_dmask.use_out(dst_mask, st, "warp_with_mask: dst_mask"));
        double bytes = spx * (src_mask ? 4.0 : 3.0) + dpx * (dst->type == ISX_16SC3 ? 7.0 : 4.0);
        // dword stores need 4-byte aligned destination rows (pitch-aligned mats; a dense cv::Mat whose
        // row length is not a multiple of 4 bytes takes the per-pixel store path)
        const isx_mat& dd = w->st_dst.d;
        const isx_mat& dm = w->st_dmask.d;
        // k_warp_tile stores 12-byte runs at any alignment (a dense cv::Mat row need not start on a dword); the caller-mask kernel keeps its
        // dword stores for aligned rows.  ISX_WARP_VEC=0: per-pixel stores (A/B runs)
        static const bool vec_any = [] { const char* e = getenv("ISX_WARP_VEC"); return !(e && e[0] == '0'); }();
        const bool vec_al = ((uintptr_t)dd.data % 4 == 0) && (dd.step % 4 == 0) && ((uintptr_t)dm.data % 4 == 0) && (dm.step % 4 == 0);
        const bool vec = src_mask ? vec_al : vec_any;
        ISX_CHECK_ARG(dd.step < (1u << 24) && dm.step < (1u << 24) && (unsigned long long)dd.step * dh < (1ull << 32), ISX_ERR_UNSUPPORTED,
                      "warp_with_mask: destination larger than 4 GiB or 16 MiB per row");
        dim3 grid4(cdiv(dw, 256), cdiv(dh, 4));
        // sync path: the scan ran on this stream and the host has consumed its keys; the kernel re-arms them.
        // planned path: scan + check run on the side stream, nothing to do here.
        const unsigned* plan_keys = nullptr;
        int* plan_mism = nullptr;
        const int4 plan4 = make_int4(0, 0, 0, 0);
#define ISX_WARP_FUSED(O16, V)                                                                                                   \
        ISX_LAUNCH("warp_img_mask", bytes, st, (k_warp_img_mask<O16, V>), grid4, dim3(256), 0, w->proj, t, sv, mv, src_mask ? 1 : 0, \
                   (unsigned char*)dd.data, dd.step, (unsigned char*)dm.data, dm.step, dw, dh, plan_keys, plan4, plan_mism)
        // the hot kernel: a tile whose mask is all 255 (W:213-214).  Launch names are the kernels' names without the k_ (they can be
        // found in a rocprofv3 kernel trace as they are)
        // isx_warper_set_dst_columns: only the 64-column blocks that hold columns [col0, col1) of the warped tile are computed; the
        // kernel's right crop is the range's end, its left end the block boundary at or below col0
        int bx0 = 0, wcrop = dw;
        if (w->col1 > w->col0 && !src_mask) {
            bx0 = std::min(w->col0, dw - 1) / 64; wcrop = std::min(w->col1, dw);
            bytes *= (double)(cdiv(wcrop, 64) - bx0) / cdiv(dw, 64);
        }
        const dim3 gridt(cdiv(wcrop, 64) - bx0, cdiv(dh, 4 * WARP_WAVES));
        const bool gained = w->gain != 1.0;
        ISX_CHECK_ARG(!(gained && src_mask), ISX_ERR_UNSUPPORTED, "warp_with_mask: isx_warper_set_gain applies to tiles warped with the all-255 mask (src_mask == NULL)");
        static const int warp_xg = [] { const char* e = getenv("ISX_WARP_XG"); return e ? atoi(e) : 0; }();
        const int xg = (warp_xg > 0 && gridt.x >= 2 && (unsigned long long)gridt.x * gridt.y * gridt.x < (1ull << 32)) ? warp_xg : 0;   // (one block column: nothing to pair, and the magic would overflow)
        WarpTileArgs wta{w->proj, t, sv, TileDst{(unsigned char*)dd.data, (unsigned)dd.step, (unsigned char*)dm.data, (unsigned)dm.step, wcrop, dh, bx0, xg, 0xFFFFFFFFu / gridt.x + 1u}, {}};
        if (gained) memcpy(wta.lut, w->gain_lut, 256);
#define ISX_WARP_TILE(KD, O16, V)                                                                                                            \
    do {                                                                                                                                     \
        if (gained) ISX_LAUNCH("warp_tile", bytes, st, (k_warp_tile<KD, O16, V, true, true>), gridt, dim3(64 * WARP_WAVES), 0, wta);          \
        else ISX_LAUNCH("warp_tile", bytes, st, (k_warp_tile<KD, O16, V>), gridt, dim3(64 * WARP_WAVES), 0, wta);                              \
    } while (0)
#define ISX_WARP_TILE_K(O16, V) do { if (w->kind == ISX_WARP_CYLINDRICAL) ISX_WARP_TILE(ISX_WARP_CYLINDRICAL, O16, V); else ISX_WARP_TILE(ISX_WARP_SPHERICAL, O16, V); } while (0)
        const bool collect = !src_mask && w->batching && !gained && dst->device >= 0 && dst_mask->device >= 0 && src->device >= 0;
        if (!collect) ISX_TRY(flush_warp_batch(w));      // whatever is launched here goes out BEHIND what was collected so far
        if (collect) {
            // collected: leaves with the other tiles of the batch as one launch (flush_warp_batch)
            isx_warper::BatchItem bi;
            bi.variant = (w->kind == ISX_WARP_CYLINDRICAL ? 0 : 4) | (dst->type == ISX_16SC3 ? 2 : 0) | (vec ? 1 : 0);
            bi.g = WarpTileGeom{wta.p, wta.t, wta.img, wta.d};
            bi.gx = gridt.x; bi.gy = gridt.y; bi.bytes = bytes;
            w->batch.push_back(bi);
        } else if (!src_mask) {
            if (dst->type == ISX_16SC3) { if (vec) ISX_WARP_TILE_K(true, true); else ISX_WARP_TILE_K(true, false); }
            else { if (vec) ISX_WARP_TILE_K(false, true); else ISX_WARP_TILE_K(false, false); }
        } else if (dst->type == ISX_16SC3) { if (vec) ISX_WARP_FUSED(true, true); else ISX_WARP_FUSED(true, false); }
        else { if (vec) ISX_WARP_FUSED(false, true); else ISX_WARP_FUSED(false, false); }
#undef ISX_WARP_TILE_K
#undef ISX_WARP_TILE
#undef ISX_WARP_FUSED
        // a column range (isx_warper_set_dst_columns) leaves the other columns of the mats as they are - also of host mats, of which only the
        // computed columns come back from the staging buffers
        ranged = w->col1 > w->col0 && !src_mask;
        rc0 = bx0 * 64; rc1 = wcrop;
        if (ranged) ISX_TRY(w->st_dmask.finish_out_cols(st, rc0, rc1)); else ISX_TRY(w->st_dmask.finish_out(st));
        if (planned && verify_plan) {
            int scratch[4];
            ISX_TRY(detect_roi(w, src->cols, src->rows, scratch, nullptr, true, planned));
        }
    } else {
        ISX_TRY(flush_warp_batch(w));      // (a plain warp() inside a batch: behind the collected tile warps)
        ISX_CHECK_ARG(dst->type == src->type, ISX_ERR_TYPE, "warp: dst type %s differs from src type %s", type_name(dst->type), type_name(src->type));
        ISX_CHECK_ARG(interp == ISX_INTER_NEAREST || interp == ISX_INTER_LINEAR || interp == (ISX_INTER_LINEAR | ISX_INTER_TIES_EVEN), ISX_ERR_UNSUPPORTED,
                      "warp: interpolation %d (only NEAREST and LINEAR)", interp);
        ISX_CHECK_ARG(border >= ISX_BORDER_CONSTANT && border <= ISX_BORDER_REFLECT_101, ISX_ERR_UNSUPPORTED, "warp: border mode %d", border);
        double bytes = (spx + dpx) * mat_elem_size(src->type);
        unsigned char* dp = (unsigned char*)w->st_dst.d.data;
        size_t ds = w->st_dst.d.step;
        // The two calls the reference makes per tile (W:229 image LINEAR / REFLECT, W:232 mask NEAREST / CONSTANT) take the tile kernels:
        // k_warp_tile without its mask output, k_warp_mask_tile.  Same limits as the fused entry (32-bit offsets from 24-bit multiplies).
        static const bool tile_path = [] { const char* e = getenv("ISX_WARP_LITERAL_FAST"); return !(e && e[0] == '0'); }();
        const bool small = (unsigned long long)w->st_src.d.step * src->rows < (1ull << 31) && w->st_src.d.step < (1u << 24) && src->cols <= 32767 && src->rows <= 32767 &&
                           ds < (1u << 24) && (unsigned long long)ds * dh < (1ull << 32);
        if (tile_path && small && src->type == ISX_8UC3 && interp == ISX_INTER_LINEAR && border == ISX_BORDER_REFLECT) {
            static const bool vec = [] { const char* e = getenv("ISX_WARP_VEC"); return !(e && e[0] == '0'); }();
            const dim3 gridt(cdiv(dw, 64), cdiv(dh, 4 * WARP_WAVES));
            static const int warp_xg = [] { const char* e = getenv("ISX_WARP_XG"); return e ? atoi(e) : 0; }();
            const int xg = (warp_xg > 0 && gridt.x >= 2 && (unsigned long long)gridt.x * gridt.y * gridt.x < (1ull << 32)) ? warp_xg : 0;
            const WarpTileArgs wta{w->proj, t, sv, TileDst{dp, (unsigned)ds, nullptr, 0u, dw, dh, 0, xg, 0xFFFFFFFFu / gridt.x + 1u}, {}};
#define ISX_WARP_IMG(KD, V) ISX_LAUNCH("warp_tile_img", bytes, st, (k_warp_tile<KD, false, V, false>), gridt, dim3(64 * WARP_WAVES), 0, wta)
            if (w->kind == ISX_WARP_CYLINDRICAL) { if (vec) ISX_WARP_IMG(ISX_WARP_CYLINDRICAL, true); else ISX_WARP_IMG(ISX_WARP_CYLINDRICAL, false); }
            else { if (vec) ISX_WARP_IMG(ISX_WARP_SPHERICAL, true); else ISX_WARP_IMG(ISX_WARP_SPHERICAL, false); }
#undef ISX_WARP_IMG
        } else if (tile_path && small && src->type == ISX_8UC1 && interp == ISX_INTER_NEAREST && border == ISX_BORDER_CONSTANT) {
            static const bool vec = [] { const char* e = getenv("ISX_WARP_VEC"); return !(e && e[0] == '0'); }();
            const WarpMaskArgs wma{w->proj, t, sv, dp, (unsigned)ds, dw, dh};
            const dim3 gridm(cdiv(dw, 64), cdiv(dh, 16));
#define ISX_WARP_MSK(KD, V) ISX_LAUNCH("warp_tile_mask", bytes, st, (k_warp_mask_tile<KD, V>), gridm, dim3(256), 0, wma)
            if (w->kind == ISX_WARP_CYLINDRICAL) { if (vec) ISX_WARP_MSK(ISX_WARP_CYLINDRICAL, true); else ISX_WARP_MSK(ISX_WARP_CYLINDRICAL, false); }
            else { if (vec) ISX_WARP_MSK(ISX_WARP_SPHERICAL, true); else ISX_WARP_MSK(ISX_WARP_SPHERICAL, false); }
#undef ISX_WARP_MSK
        } else
        switch (src->type) {
            case ISX_8UC3: ISX_LAUNCH("warp", bytes, st, (k_warp<unsigned char, 3>), grid, dim3(256), 0, w->proj, t, sv, dp, ds, dw, dh, interp, border); break;
            case ISX_8UC1: ISX_LAUNCH("warp", bytes, st, (k_warp<unsigned char, 1>), grid, dim3(256), 0, w->proj, t, sv, dp, ds, dw, dh, interp, border); break;
            case ISX_32FC3: ISX_LAUNCH("warp", bytes, st, (k_warp<float, 3>), grid, dim3(256), 0, w->proj, t, sv, dp, ds, dw, dh, interp, border); break;
            case ISX_32FC1: ISX_LAUNCH("warp", bytes, st, (k_warp<float, 1>), grid, dim3(256), 0, w->proj, t, sv, dp, ds, dw, dh, interp, border); break;
            default: return fail(ISX_ERR_TYPE, "warp: src type %s is not supported (CV_8UC1/3, CV_32FC1/3)", type_name(src->type));
        }
    }
    if (ranged) ISX_TRY(w->st_dst.finish_out_cols(st, rc0, rc1)); else ISX_TRY(w->st_dst.finish_out(st));
    if (corner) { corner[0] = roi[0]; corner[1] = roi[1]; }   // return dst_roi.tl()  W:160
    return ISX_OK;
}

}  // namespace

extern "C" {

int isx_warper_create(int kind, float scale, int device, isx_warper** out) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(out != nullptr, ISX_ERR_INVALID, "isx_warper_create: null out pointer");
    *out = nullptr;
    ISX_CHECK_ARG(kind == ISX_WARP_CYLINDRICAL || kind == ISX_WARP_SPHERICAL, ISX_ERR_INVALID, "isx_warper_create: unknown warper kind %d", kind);
    ISX_CHECK_ARG(std::isfinite(scale) && scale > 0.f, ISX_ERR_INVALID, "isx_warper_create: scale must be positive, got %g", (double)scale);
    int n = 0;
    ISX_HIP(hipGetDeviceCount(&n));
    ISX_CHECK_ARG(device >= 0 && device < n, ISX_ERR_INVALID, "isx_warper_create: device %d of %d", device, n);
    isx_warper* w = new (std::nothrow) isx_warper();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_NOMEM, "isx_warper_create: out of host memory");
    w->kind = kind; w->scale = scale; w->device = device;
    *out = w;
    return ISX_OK;
} ISX_EXIT("isx_warper_create")

int isx_warper_destroy(isx_warper* w) ISX_ENTRY {
    if (!w) return ISX_OK;
    (void)hipSetDevice(w->device);
    (void)hipStreamSynchronize(w->stream);
    if (w->side) { (void)hipStreamSynchronize(w->side); (void)hipEventDestroy(w->ev_warp); (void)hipEventDestroy(w->ev_scan); }   // the side stream is shared per device
    if (w->pin) (void)hipHostFree(w->pin);
    if (w->pin2) (void)hipHostFree(w->pin2);
    delete w;
    return ISX_OK;
} ISX_EXIT("isx_warper_destroy")

int isx_warper_begin_batch(isx_warper* w) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_begin_batch: null warper");
    ISX_HIP(hipSetDevice(w->device));
    ISX_TRY(flush_warp_batch(w));
    w->batching = true;
    return ISX_OK;
} ISX_EXIT("isx_warper_begin_batch")

int isx_warper_end_batch(isx_warper* w) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_end_batch: null warper");
    ISX_HIP(hipSetDevice(w->device));
    w->batching = false;
    ISX_TRY(flush_warp_batch(w));
    if (!w->defer_verify) return flush_verify(w);        // the planned warps' verification scans, behind the launch they verify
    return ISX_OK;
} ISX_EXIT("isx_warper_end_batch")

int isx_warper_set_stream(isx_warper* w, void* hip_stream) ISX_ENTRY {
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_set_stream: null warper");
    ISX_TRY(flush_warp_batch(w));      // collected warps leave on the stream they were issued for
    w->stream = (hipStream_t)hip_stream;
    return ISX_OK;
} ISX_EXIT("isx_warper_set_stream")

int isx_warper_set_gain(isx_warper* w, double gain) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_set_gain: null warper");
    if (gain == w->gain) return ISX_OK;
    if (gain != 1.0) {
        // saturate_cast<uchar>(cvRound((double)v * gain)) for every byte value, with isx_gain_apply's arithmetic (cvtsd2si: ties to even,
        // NaN / overflow -> INT_MIN -> 0)
        for (int v = 0; v < 256; ++v) {
            const double t = std::nearbyint((double)v * gain);
            const int iv = (t >= -2147483648.0 && t <= 2147483647.0) ? (int)t : INT_MIN;
            w->gain_lut[v] = (unsigned char)((unsigned)iv <= 255u ? iv : (iv > 0 ? 255 : 0));
        }
    }
    w->gain = gain;
    return ISX_OK;
} ISX_EXIT("isx_warper_set_gain")

int isx_warper_set_roi_cache(isx_warper* w, int on) ISX_ENTRY {
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_set_roi_cache: null warper");
    w->roi_cache_on = on != 0;
    if (!on) w->roi_cache.clear();      // (the last result is remembered again from the next call on)
    return ISX_OK;
} ISX_EXIT("isx_warper_set_roi_cache")

int isx_warper_set_dst_columns(isx_warper* w, int col0, int col1) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_set_dst_columns: null warper");
    ISX_CHECK_ARG((col0 == 0 && col1 == 0) || (col0 >= 0 && col1 > col0), ISX_ERR_INVALID, "isx_warper_set_dst_columns: columns [%d, %d)", col0, col1);
    w->col0 = col0; w->col1 = col1;
    return ISX_OK;
} ISX_EXIT("isx_warper_set_dst_columns")

int isx_warper_camera(isx_warper* w, const float K[9], const float R[9], float r_kinv[9], float k_rinv[9]) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_camera: null warper");
    ISX_TRY(set_camera(w, K, R));
    if (r_kinv) std::copy(w->proj.r_kinv, w->proj.r_kinv + 9, r_kinv);
    if (k_rinv) std::copy(w->proj.k_rinv, w->proj.k_rinv + 9, k_rinv);
    return ISX_OK;
} ISX_EXIT("isx_warper_camera")

int isx_warper_roi(isx_warper* w, int src_w, int src_h, const float K[9], const float R[9], int roi[4], float minmax[4]) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr && roi != nullptr, ISX_ERR_INVALID, "isx_warper_roi: null argument");
    ISX_CHECK_ARG(src_w > 0 && src_h > 0, ISX_ERR_INVALID, "isx_warper_roi: empty source size %d x %d", src_w, src_h);
    ISX_HIP(hipSetDevice(w->device));
    ISX_TRY(set_camera(w, K, R));
    return detect_roi(w, src_w, src_h, roi, minmax, false, nullptr);
} ISX_EXIT("isx_warper_roi")

namespace {
int build_maps_common(isx_warper* w, int src_w, int src_h, const float K[9], const float R[9], isx_mat* xmap, isx_mat* ymap, int roi[4], bool given_roi) {
    ISX_CHECK_ARG(w != nullptr && roi != nullptr, ISX_ERR_INVALID, "buildMaps: null argument");
    ISX_TRY(check_mat(xmap, "buildMaps: xmap"));
    ISX_TRY(check_mat(ymap, "buildMaps: ymap"));
    ISX_CHECK_ARG(xmap->type == ISX_32FC1 && ymap->type == ISX_32FC1, ISX_ERR_TYPE, "buildMaps: maps must be CV_32FC1");
    ISX_HIP(hipSetDevice(w->device));
    ISX_TRY(set_camera(w, K, R));
    if (!given_roi) ISX_TRY(detect_roi(w, src_w, src_h, roi, nullptr, false, nullptr));
    ISX_TRY(check_roi_sane(roi));
    const int dw = roi[2] - roi[0] + 1, dh = roi[3] - roi[1] + 1;   // W:128-129
    ISX_CHECK_ARG(xmap->rows == dh && xmap->cols == dw && ymap->rows == dh && ymap->cols == dw, ISX_ERR_SIZE,
                  "buildMaps: maps must be %dx%d", dw, dh);
    MapTabs t;
    ISX_TRY(make_tabs(w, roi, &t));
    hipStream_t st = w->stream;
    ISX_TRY(w->st_x.use_out(xmap, st, "buildMaps: xmap"));
    ISX_TRY(w->st_y.use_out(ymap, st, "buildMaps: ymap"));
    dim3 grid(cdiv(dw, 64), cdiv(dh, 4));
    ISX_LAUNCH("build_maps", (double)dw * dh * 8.0, st, k_build_maps, grid, dim3(256), 0, w->proj, t, (float*)w->st_x.d.data, w->st_x.d.step,
               (float*)w->st_y.d.data, w->st_y.d.step, dw, dh);
    ISX_TRY(w->st_x.finish_out(st));
    ISX_TRY(w->st_y.finish_out(st));
    return ISX_OK;
}
}  // namespace

int isx_warper_build_maps(isx_warper* w, int src_w, int src_h, const float K[9], const float R[9], isx_mat* xmap, isx_mat* ymap, int roi[4]) ISX_ENTRY {
    clear_error();
    return build_maps_common(w, src_w, src_h, K, R, xmap, ymap, roi, false);
} ISX_EXIT("isx_warper_build_maps")

int isx_warper_build_maps_roi(isx_warper* w, const float K[9], const float R[9], const int roi[4], isx_mat* xmap, isx_mat* ymap) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(roi != nullptr, ISX_ERR_INVALID, "buildMaps: null roi");
    int r[4] = {roi[0], roi[1], roi[2], roi[3]};
    return build_maps_common(w, 0, 0, K, R, xmap, ymap, r, true);
} ISX_EXIT("isx_warper_build_maps_roi")

int isx_remap(const isx_mat* src, const isx_mat* xmap, const isx_mat* ymap, int interp, int border, isx_mat* dst, int device, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_TRY(check_mat(src, "remap: src"));
    ISX_TRY(check_mat(xmap, "remap: xmap"));
    ISX_TRY(check_mat(ymap, "remap: ymap"));
    ISX_TRY(check_mat(dst, "remap: dst"));
    ISX_CHECK_ARG(xmap->type == ISX_32FC1 && ymap->type == ISX_32FC1, ISX_ERR_TYPE, "remap: maps must be CV_32FC1 (W:128-129)");
    ISX_CHECK_ARG(xmap->rows == ymap->rows && xmap->cols == ymap->cols, ISX_ERR_SIZE, "remap: xmap and ymap differ in size");
    ISX_CHECK_ARG(dst->rows == xmap->rows && dst->cols == xmap->cols && dst->type == src->type, ISX_ERR_SIZE, "remap: dst must have the maps' size and the source's type");
    ISX_CHECK_ARG(src->type == ISX_8UC1 || src->type == ISX_8UC3 || src->type == ISX_32FC1 || src->type == ISX_32FC3, ISX_ERR_TYPE,
                  "remap: CV_8UC1 / CV_8UC3 / CV_32FC1 / CV_32FC3 sources are supported, got %s", type_name(src->type));
    ISX_CHECK_ARG(interp == ISX_INTER_NEAREST || interp == ISX_INTER_LINEAR || interp == (ISX_INTER_LINEAR | ISX_INTER_TIES_EVEN), ISX_ERR_UNSUPPORTED,
                  "remap: INTER_NEAREST and INTER_LINEAR are implemented");
    ISX_CHECK_ARG(border >= ISX_BORDER_CONSTANT && border <= ISX_BORDER_REFLECT_101, ISX_ERR_UNSUPPORTED, "remap: unsupported border mode %d", border);
    ISX_CHECK_ARG(src->cols <= 32767 && src->rows <= 32767, ISX_ERR_UNSUPPORTED, "remap: source larger than 32767 pixels per side (cv::remap's short coordinates)");
    ISX_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    MatStage ss, sx, sy, sd;
    ISX_TRY(ss.use_in(src, st, "remap: src"));
    ISX_TRY(sx.use_in(xmap, st, "remap: xmap"));
    ISX_TRY(sy.use_in(ymap, st, "remap: ymap"));
    ISX_TRY(sd.use_out(dst, st, "remap: dst"));
    SrcView sv{(const unsigned char*)ss.d.data, ss.d.step, src->rows, src->cols};
    const int dw = dst->cols, dh = dst->rows;
    dim3 grid(cdiv(dw, 64), cdiv(dh, 4));
    const double bytes = (double)dw * dh * (8.0 + 2.0 * mat_elem_size(src->type));
#define ISX_REMAP(T, CN)                                                                                                                     \
    ISX_LAUNCH("remap", bytes, st, (k_remap<T, CN>), grid, dim3(256), 0, sv, (const unsigned char*)sx.d.data, sx.d.step, (const unsigned char*)sy.d.data, \
               sy.d.step, (unsigned char*)sd.d.data, sd.d.step, dw, dh, interp, border)
    switch (src->type) {
        case ISX_8UC1: ISX_REMAP(unsigned char, 1); break;
        case ISX_8UC3: ISX_REMAP(unsigned char, 3); break;
        case ISX_32FC1: ISX_REMAP(float, 1); break;
        default: ISX_REMAP(float, 3); break;
    }
#undef ISX_REMAP
    ISX_TRY(sd.finish_out(st));
    if (src->device < 0 || xmap->device < 0 || ymap->device < 0 || dst->device < 0) ISX_HIP(hipStreamSynchronize(st));   // staging buffers are freed on return
    return ISX_OK;
} ISX_EXIT("isx_remap")

int isx_warper_warp(isx_warper* w, const isx_mat* src, const float K[9], const float R[9], int interp, int border, isx_mat* dst, int corner[2]) ISX_ENTRY {
    clear_error();
    return warp_common(w, src, nullptr, K, R, interp, border, dst, nullptr, corner, nullptr, false);
} ISX_EXIT("isx_warper_warp")

int isx_warper_warp_with_mask(isx_warper* w, const isx_mat* src_img, const isx_mat* src_mask, const float K[9], const float R[9],
                              isx_mat* dst_img, isx_mat* dst_mask, int corner[2]) ISX_ENTRY {
    clear_error();
    return warp_common(w, src_img, src_mask, K, R, ISX_INTER_LINEAR, ISX_BORDER_REFLECT, dst_img, dst_mask, corner, nullptr, true);
} ISX_EXIT("isx_warper_warp_with_mask")

int isx_warper_warp_with_mask_planned(isx_warper* w, const isx_mat* src_img, const isx_mat* src_mask, const float K[9], const float R[9],
                                      const int planned_roi[4], isx_mat* dst_img, isx_mat* dst_mask) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(planned_roi != nullptr, ISX_ERR_INVALID, "planned warp: null planned_roi");
    return warp_common(w, src_img, src_mask, K, R, ISX_INTER_LINEAR, ISX_BORDER_REFLECT, dst_img, dst_mask, nullptr, planned_roi, true);
} ISX_EXIT("isx_warper_warp_with_mask_planned")

int isx_warper_warp_roi(isx_warper* w, const isx_mat* src, const float K[9], const float R[9], int interp, int border, const int roi[4], isx_mat* dst) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(roi != nullptr, ISX_ERR_INVALID, "warp_roi: null roi");
    return warp_common(w, src, nullptr, K, R, interp, border, dst, nullptr, nullptr, roi, false, false);
} ISX_EXIT("isx_warper_warp_roi")

int isx_warper_warp_with_mask_roi(isx_warper* w, const isx_mat* src_img, const isx_mat* src_mask, const float K[9], const float R[9],
                                  const int roi[4], isx_mat* dst_img, isx_mat* dst_mask) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(roi != nullptr, ISX_ERR_INVALID, "warp_with_mask_roi: null roi");
    return warp_common(w, src_img, src_mask, K, R, ISX_INTER_LINEAR, ISX_BORDER_REFLECT, dst_img, dst_mask, nullptr, roi, true, false);
} ISX_EXIT("isx_warper_warp_with_mask_roi")

// detectResultRoi computed on the host alone, for the cameras whose ROI the synchronous path takes from the border (every spherical camera;
// a cylindrical one where cyl_extrema_on_border holds): what isx_warper_roi returns there, without a device - the CPU test-suite compares it
// with the oracle's scan of every source pixel.  isa: 0 = the code isx_warper_roi runs, 1 = the scalar form.  ISX_ERR_UNSUPPORTED: a
// cylindrical camera whose extrema are not provably on the border (isx_warper_roi scans every pixel on the device there).
int isx_selftest_roi_host(int kind, float scale, const float K[9], const float R[9], int src_w, int src_h, int isa, int roi[4], float minmax[4]) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(K != nullptr && R != nullptr && roi != nullptr, ISX_ERR_INVALID, "selftest_roi_host: null argument");
    ISX_CHECK_ARG((kind == ISX_WARP_CYLINDRICAL || kind == ISX_WARP_SPHERICAL) && src_w > 0 && src_h > 0, ISX_ERR_INVALID, "selftest_roi_host: bad kind / size");
    Proj p;
    float k[9], rinv[9], kinv[9];
    for (int i = 0; i < 9; ++i) k[i] = K[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rinv[i * 3 + j] = R[j * 3 + i];
    mat3_inv(K, kinv);
    mat3_mul(R, kinv, p.r_kinv);
    mat3_mul(K, rinv, p.k_rinv);
    p.scale = scale; p.kind = kind;
    ISX_CHECK_ARG(kind == ISX_WARP_SPHERICAL || cyl_extrema_on_border(p, k, rinv, src_w, src_h), ISX_ERR_UNSUPPORTED,
                  "selftest_roi_host: the extrema of this cylindrical camera are not provably on the border");
    std::vector<int> cand;
    std::vector<float> scratch;
    int n = 0;
    ISX_TRY(border_scan_host(p.r_kinv, kind == ISX_WARP_SPHERICAL, src_w, src_h, cand, scratch, isa, &n));
    float mm[4];
    roi_from_candidates(p, k, rinv, src_w, src_h, cand.data(), n, roi, mm);
    if (minmax) std::copy(mm, mm + 4, minmax);
    return ISX_OK;
} ISX_EXIT("isx_selftest_roi_host")

int isx_selftest_division(int device, int n, unsigned long long seed, int* mismatches) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(mismatches != nullptr && n > 0, ISX_ERR_INVALID, "selftest_division: bad argument");
    ISX_HIP(hipSetDevice(device));
    unsigned* dm = nullptr;
    ISX_HIP(hipMalloc(&dm, sizeof(unsigned)));
    ISX_HIP(hipMemset(dm, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(k_selftest_division, dim3(cdiv(n, 256)), dim3(256), 0, 0, seed, n, dm);
    unsigned h = 0;
    hipError_t e = hipMemcpy(&h, dm, sizeof(unsigned), hipMemcpyDeviceToHost);
    (void)hipFree(dm);
    ISX_HIP(e);
    *mismatches = (int)h;
    return ISX_OK;
} ISX_EXIT("isx_selftest_division")

int isx_warper_set_deferred_verify(isx_warper* w, int on) ISX_ENTRY {
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_set_deferred_verify: null warper");
    w->defer_verify = on != 0;
    return ISX_OK;
} ISX_EXIT("isx_warper_set_deferred_verify")

int isx_warper_verify_is_light(isx_warper* w, int src_cols, int src_rows, const float K[9], const float R[9], int* light) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr && light != nullptr && src_cols > 0 && src_rows > 0, ISX_ERR_INVALID, "isx_warper_verify_is_light: bad argument");
    ISX_TRY(set_camera(w, K, R));
    *light = (w->kind == ISX_WARP_SPHERICAL || cyl_extrema_on_border(w->proj, w->k, w->rinv, src_cols, src_rows)) ? 1 : 0;
    return ISX_OK;
} ISX_EXIT("isx_warper_verify_is_light")

int isx_warper_verify(isx_warper* w) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_verify: null warper");
    ISX_HIP(hipSetDevice(w->device));
    return flush_verify(w);
} ISX_EXIT("isx_warper_verify")

int isx_warper_verify_after(isx_warper* w, void* hip_event) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr && hip_event != nullptr, ISX_ERR_INVALID, "isx_warper_verify_after: null argument");
    ISX_HIP(hipSetDevice(w->device));
    return flush_verify(w, (hipEvent_t)hip_event);
} ISX_EXIT("isx_warper_verify_after")

// Verification outside a captured step.  A planned warp queues the scan that checks its plan; inside a hipGraph that scan has to be forked
// from the captured stream by an event, and the fork cost a replayed step 12 us (0.206 -> 0.218 ms at 4K; the eager step starts the scan on the
// side stream with no event at all).  A capturing caller therefore drops the queued scans of the captured warps
// (isx_warper_discard_pending) and, after every replay, queues the same verifications from the rig alone (isx_warper_queue_verify: the scan
// reads the projection and the source size, never an image) and starts them beside the graph (isx_warper_verify).
int isx_warper_discard_pending(isx_warper* w) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_discard_pending: null warper");
    w->pending.clear();
    return ISX_OK;
} ISX_EXIT("isx_warper_discard_pending")

int isx_warper_queue_verify(isx_warper* w, int src_w, int src_h, const float K[9], const float R[9], const int planned_roi[4]) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr && K != nullptr && R != nullptr && planned_roi != nullptr, ISX_ERR_INVALID, "isx_warper_queue_verify: null argument");
    ISX_CHECK_ARG(src_w > 0 && src_h > 0, ISX_ERR_INVALID, "isx_warper_queue_verify: empty source %d x %d", src_w, src_h);
    ISX_HIP(hipSetDevice(w->device));
    ISX_TRY(set_camera(w, K, R));
    isx_warper::Pending pd;
    pd.proj = w->proj; pd.sw = src_w; pd.sh = src_h;
    std::copy(planned_roi, planned_roi + 4, pd.planned);
    std::copy(w->k, w->k + 9, pd.k); std::copy(w->rinv, w->rinv + 9, pd.rinv);
    w->pending.push_back(pd);
    return ISX_OK;
} ISX_EXIT("isx_warper_queue_verify")

int isx_warper_join(isx_warper* w) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr, ISX_ERR_INVALID, "isx_warper_join: null warper");
    ISX_HIP(hipSetDevice(w->device));
    ISX_TRY(flush_verify(w));
    if (!w->side) return ISX_OK;
    ISX_HIP(hipStreamWaitEvent(w->stream, w->ev_scan, 0));
    return ISX_OK;
} ISX_EXIT("isx_warper_join")

int isx_warper_plan_status(isx_warper* w, int* mismatches) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(w != nullptr && mismatches != nullptr, ISX_ERR_INVALID, "plan_status: null argument");
    *mismatches = 0;
    ISX_HIP(hipSetDevice(w->device));
    ISX_TRY(flush_verify(w));
    ISX_CHECK_ARG(w->verify_dropped == 0, ISX_ERR_PLAN, "planned warp: %d verification(s) were dropped under ISX_VERIFY_NEVER - this run's plans are unverified", w->verify_dropped);
    if (!w->scan_side.p) return ISX_OK;
    ISX_HIP(hipStreamSynchronize(w->side));
    ISX_HIP(hipStreamSynchronize(w->stream));
    ISX_HIP(hipMemcpy(mismatches, (int*)w->scan_side.p + 5, sizeof(int), hipMemcpyDeviceToHost));
    if (*mismatches) return fail(ISX_ERR_PLAN, "planned warp: %d run(s) produced a ROI that differs from the planned one", *mismatches);
    return ISX_OK;
} ISX_EXIT("isx_warper_plan_status")

}  // extern "C"
