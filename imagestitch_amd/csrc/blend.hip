// blend.hip — multi-band Laplacian blender for MI355X (gfx950): Gaussian/Laplacian pyramid build,
// per-band weighted accumulation, normalise + collapse — and the FeatherBlender.  Replaces OpenCV 3.4.2
// cv::detail::MultiBandBlender / FeatherBlender as the reference calls them (W:271-281,302,313; spec of the
// arithmetic: SURVEY.md §8(a) A9-A12).  HBM-bound stencil work: no MFMA, 16-byte pixel records,
// coalesced row-major access, LDS tiles with halo, wavefront shuffles for the horizontal 5-tap.
//
// Data layout in HBM (per pyramid level, row-major, pitch == cols):
//   F32      : tile Gaussian levels and destination levels are float4 {b,g,r,weight}
//   F16ACC32 : tile Gaussian levels are 4 x f16 {b,g,r,weight} (8 B); destination levels float4
//   I16      : destination levels short4 {b,g,r,0} + a separate float plane for the weight (OpenCV's CV_16SC3 + CV_32F);
//              tile Gaussian levels {int b,g,r; float weight} (16 B, the register record: one load per pixel, and the
//              coarse tiles can be staged by LDS-DMA like the fp32 ones; the values are still OpenCV's shorts)
// Level 0 of a fed tile is never materialised: the level-0 kernels read the caller's image + mask
// through the copyMakeBorder index maps (BORDER_REFLECT image, BORDER_CONSTANT weight).
//
// Kernels:
//   eager cycle (OpenCV's contract, feed() consumes its inputs)
//     k_pyr_down      : G_{k+1} = pyrDown(G_k)           (image + weight in one pass)
//     k_lap_acc(_all) : dst_k[rc] += cast((G_k - pyrUp(G_{k+1})) * W_k), dstW_k[rc] += W_k   (all levels of a feed in one launch)
//     k_top_acc       : dst_L[rc] += cast(G_L * W_L), dstW_L[rc] += W_L
//     k_collapse      : out_{k-1} = sat(pyrUp(out_k) + norm(dst_{k-1})); last level writes the caller's mat
//   deferred cycle (isx_blender_set_deferred_level0; the measured path)
//     k_pyr_down_multi  : the Gaussian chains of all recorded tiles, one launch per level
//     k_collapse_gather : one collapse step that gathers the tiles' Laplacians in registers (the destination pyramid
//                         never exists); the first step also gathers the top level, the last writes the caller's mats
//   FeatherBlender (W:278-281,302,313): k_dt_rows / k_dt_seg_min / k_dt_cols_weight (createWeightMap),
//     k_feather_acc + k_feather_blend (eager), k_feather_gather (deferred)
#include "isx_device.hpp"
#include "isx_internal.hpp"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <memory>
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
    // fast path of load_src0_pair (CV_8UC3 tiles below 2 GiB): 4-byte aligned bases, the misalignment of
    // the real base pointers and the end of the addressable bytes, all as 32-bit offsets (0 = disabled)
    const unsigned char* img_al;
    const unsigned char* mask_al;
    unsigned imis, mmis, iend, mend;
};

// Rectangles of one destination level that earlier feeds have already written.  prepare() does not
// clear the destination pyramid (a 260 MB memset per 4K pair): a pixel outside every rectangle is
// DEFINED to be zero — the first feed that touches it stores instead of accumulating, and blend()
// reads zero for pixels no tile ever covered.  n < 0: the level has been cleared, always load.
constexpr int MAX_COVER = 8;
struct Cover {
    int n;
    int x[MAX_COVER], y[MAX_COVER], w[MAX_COVER], h[MAX_COVER];
};
// (x, y) are level-k coordinates; the rectangles are stored at level 0 when k > 0 is passed
__device__ __forceinline__ bool covered(const Cover& c, int x, int y, int k = 0) {
    if (c.n < 0) return true;
    bool in = false;
#pragma unroll
    for (int i = 0; i < MAX_COVER; ++i)   // static indices: the struct stays in the kernel-argument segment
        if (i < c.n) in = in || ((unsigned)(x - (c.x[i] >> k)) < (unsigned)(c.w[i] >> k) && (unsigned)(y - (c.y[i] >> k)) < (unsigned)(c.h[i] >> k));
    return in;
}

template <int M, bool DST>
__device__ __forceinline__ Px<M> load_px(const LevelBuf& L, int x, int y) {
    // levels hold < 2^31 records and fewer than 2^24 rows / columns (checked by prepare): 24-bit multiply = full-rate VALU
    const unsigned i = __umul24((unsigned)y, (unsigned)L.cols) + (unsigned)x;
    Px<M> p;
    if constexpr (M == M_I16 && !DST) {   // tile Gaussian level: the register record itself, {int b, g, r; float w}
        const int4 v = ((const int4*)L.img)[i];
        p.c0 = v.x; p.c1 = v.y; p.c2 = v.z; p.w = __int_as_float(v.w);
    } else if constexpr (M == M_I16) {
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

// the image channels only (pyrUp sources: the weight of a coarse level is never upsampled) - for the I16 destination
// format this skips the load from the separate weight plane
template <int M, bool DST>
__device__ __forceinline__ Px<M> load_px_rgb(const LevelBuf& L, int x, int y) {
    if constexpr (M == M_I16 && DST) {
        const unsigned i = __umul24((unsigned)y, (unsigned)L.cols) + (unsigned)x;
        const short4 v = ((const short4*)L.img)[i];
        Px<M> p;
        p.c0 = v.x; p.c1 = v.y; p.c2 = v.z; p.w = 0.f;
        return p;
    } else return load_px<M, DST>(L, x, y);
}

// ISX_NT_G (A/B builds): bit 0 - the tiles' level-1 records (store_rgb12 / store_px_planar), bit 1 - every 16-byte level record (store_px) stored
// non-temporally (written through as the kernel runs instead of left dirty in L2 for the end-of-kernel write-back)
#ifndef ISX_NT_G
#define ISX_NT_G 0
#endif
template <class T>
__device__ __forceinline__ void st_g(T* p, const T& v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
template <int M, bool DST>
__device__ __forceinline__ void store_px(const LevelBuf& L, int x, int y, const Px<M>& p) {
    const unsigned i = __umul24((unsigned)y, (unsigned)L.cols) + (unsigned)x;
    if constexpr (M == M_I16 && !DST) {
        ((int4*)L.img)[i] = make_int4(p.c0, p.c1, p.c2, __float_as_int(p.w));
    } else if constexpr (M == M_I16) {
        ((short4*)L.img)[i] = make_short4((short)p.c0, (short)p.c1, (short)p.c2, 0);
        L.wgt[i] = p.w;
    } else if constexpr (M == M_F16 && !DST) {
        ((ushort4*)L.img)[i] = make_ushort4(f2h_bits(p.c0), f2h_bits(p.c1), f2h_bits(p.c2), f2h_bits(p.w));
    } else {
        if (ISX_NT_G & 2) { typedef float f4v __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(f4v{p.c0, p.c1, p.c2, p.w}, (f4v*)L.img + i); }
        else ((float4*)L.img)[i] = make_float4(p.c0, p.c1, p.c2, p.w);
    }
}

// PLANAR tile level (round 4, level 1 of the deferred cycle when k_collapse_roll runs the last step): the record's image channels as dense
// 12-byte records at L.img and its weight in a float plane at L.wgt (inside the same allocation: 12 n + 4 n bytes).  The last collapse
// step reads the image channels of level 1 only (pyrUp never reads a coarse weight): with 16-byte records it fetched the weights' 4 bytes
// per pixel for nothing (14.8 MB per 4K pair); pyrDown of level 1 and the level-1 collapse step read both parts, as many bytes as before.
// Round 5: in OpenCV's int16 arithmetic the dense records are what OpenCV's own levels are - three SHORTS, 6 bytes (every Gaussian level of a
// CV_16S pyramid and every collapsed level fits a short: pyrDown's weights sum to 256, restoreImageFromLaplacePyr saturates) - instead of the three
// ints of the register record: the last collapse step and the level-1 step move 6 bytes less per level-1 pixel and source.  A record is read as
// one 8-byte load at a 2-byte aligned address (two bytes into its neighbour / the weight plane behind the image plane: inside the allocation).
typedef unsigned u32x3_rec __attribute__((ext_vector_type(3), aligned(4)));
typedef unsigned u32x2_rec2 __attribute__((ext_vector_type(2), aligned(2)));
typedef unsigned u32_rec2 __attribute__((aligned(2)));
template <int M> constexpr unsigned dense_rec() { return M == M_I16 ? 6u : 12u; }
// Q8 (round 5, fp32 pyramids over CV_8UC3 tiles): level 1 of such a tile is pyrDown of integers 0..255 - (sum of 25 products with weights that add
// up to 256) / 256, every step exact in fp32 - so each channel is k / 256 with k <= 65280: the dense record holds the three k as unsigned SHORTS,
// 6 bytes instead of 12, and the float is rebuilt exactly ((float)k * 2^-8).  Level 1 is written once and read three times per step (pyrDown to
// level 2, the level-1 collapse step, the last step's pyrUp source): 24 bytes less per level-1 pixel and tile, 89 MB of a 4K pair's 716.  The weight
// plane stays fp32 (a mask may hold any value / 255).  Chosen by run_blend_deferred_t when the chain itself produces level 1 (ISX_G1Q8=0: off).
// where the weight plane of a planar level starts (bytes from L.img): behind the dense image records, dword aligned
inline __host__ __device__ size_t planar_wgt_offset(int prec, int rows, int cols, bool q8 = false) {
    return ((size_t)rows * cols * ((prec == M_I16 || q8) ? 6u : 12u) + 3u) & ~(size_t)3u;
}
__device__ __forceinline__ float q8_f(unsigned k) { return (float)k * (1.f / 256.f); }
template <int M, bool Q8 = false>
__device__ __forceinline__ Px<M> load_px_planar(const LevelBuf& L, int x, int y) {
    static_assert(M == M_F32 || M == M_I16, "planar tile levels: 16-byte register records only");
    static_assert(!Q8 || M == M_F32, "Q8 records: fp32 pyramids");
    const unsigned i = __umul24((unsigned)y, (unsigned)L.cols) + (unsigned)x;
    Px<M> p;
    if constexpr (M == M_I16) {
        const u32x2_rec2 v = *(const u32x2_rec2*)((const char*)L.img + (size_t)i * 6u);
        p.c0 = (int)(short)(v.x & 0xffffu); p.c1 = (int)(short)(v.x >> 16); p.c2 = (int)(short)(v.y & 0xffffu);
    } else if constexpr (Q8) {
        const u32x2_rec2 v = *(const u32x2_rec2*)((const char*)L.img + (size_t)i * 6u);
        p.c0 = q8_f(v.x & 0xffffu); p.c1 = q8_f(v.x >> 16); p.c2 = q8_f(v.y & 0xffffu);
    } else {
        const u32x3_rec v = *(const u32x3_rec*)((const char*)L.img + (size_t)i * 12u);
        p.c0 = __uint_as_float(v.x); p.c1 = __uint_as_float(v.y); p.c2 = __uint_as_float(v.z);
    }
    p.w = L.wgt[i];
    return p;
}
// the image channels of a register record as a dense record (planar tile levels; out_1 in dense records, OutMat::rec12): 12 bytes, int16: 6
template <int M, bool Q8 = false>
__device__ __forceinline__ void store_rgb12(const LevelBuf& L, unsigned i, const Px<M>& p) {
    if constexpr (M == M_I16) {
        char* q = (char*)L.img + (size_t)i * 6u;
        *(u32_rec2*)q = ((unsigned)p.c0 & 0xffffu) | ((unsigned)p.c1 << 16);
        *(unsigned short*)(q + 4) = (unsigned short)p.c2;
    } else if constexpr (Q8) {      // (exact: see Q8 above)
        const unsigned k0 = (unsigned)(p.c0 * 256.f), k1 = (unsigned)(p.c1 * 256.f), k2 = (unsigned)(p.c2 * 256.f);
        char* q = (char*)L.img + (size_t)i * 6u;
        st_g((u32_rec2*)q, (u32_rec2)(k0 | (k1 << 16)), (ISX_NT_G & 1) != 0);
        st_g((unsigned short*)(q + 4), (unsigned short)k2, (ISX_NT_G & 1) != 0);
    } else {
        u32x3_rec v;
        v.x = __float_as_uint(p.c0); v.y = __float_as_uint(p.c1); v.z = __float_as_uint(p.c2);
        st_g((u32x3_rec*)((char*)L.img + (size_t)i * 12u), v, (ISX_NT_G & 1) != 0);
    }
}
// A tile level that is planar or not - known only when the kernel runs (L.wgt != nullptr, uniform) - read WITHOUT a branch: the image channels as one
// 12-byte load at the record's address (a 2-byte aligned address for int16's 6-byte dense records, whose other 6 bytes belong to the neighbour or the
// plane behind) and the weight as one dword from wherever it lies.  A branch between the two forms, even a uniform one, makes the compiler drain
// the loads in front of it: the level-1 step waited for eight round trips per round, one after the other (round 5).
typedef unsigned u32x3_rec2 __attribute__((ext_vector_type(3), aligned(2)));
template <int M>
__device__ __forceinline__ Px<M> load_px_tile(const LevelBuf& L, int x, int y, bool q8 = false) {      // q8 (uniform): a planar level in Q8 records
    static_assert(M == M_F32 || M == M_I16, "16-byte register records or their planar form");
    const unsigned i = __umul24((unsigned)y, (unsigned)L.cols) + (unsigned)x;
    const bool pl = L.wgt != nullptr;
    const char* ip = (const char*)L.img + (size_t)i * (pl ? (q8 ? 6u : dense_rec<M>()) : 16u);
    const float* wp = pl ? L.wgt + i : (const float*)(ip + 12);
    const u32x3_rec2 v = *(const u32x3_rec2*)ip;
    Px<M> p;
    if constexpr (M == M_I16) {
        const int d0 = (int)(short)(v.x & 0xffffu), d1 = (int)(short)(v.x >> 16), d2 = (int)(short)(v.y & 0xffffu);
        p.c0 = pl ? d0 : (int)v.x; p.c1 = pl ? d1 : (int)v.y; p.c2 = pl ? d2 : (int)v.z;
    } else {
        p.c0 = q8 ? q8_f(v.x & 0xffffu) : __uint_as_float(v.x); p.c1 = q8 ? q8_f(v.x >> 16) : __uint_as_float(v.y); p.c2 = q8 ? q8_f(v.y & 0xffffu) : __uint_as_float(v.z);
    }
    p.w = *wp;
    return p;
}
template <int M, bool Q8 = false>
__device__ __forceinline__ void store_px_planar(const LevelBuf& L, int x, int y, const Px<M>& p) {
    static_assert(M == M_F32 || M == M_I16, "planar tile levels: 16-byte register records only");
    const unsigned i = __umul24((unsigned)y, (unsigned)L.cols) + (unsigned)x;
    store_rgb12<M, Q8>(L, i, p);
    st_g(&L.wgt[i], p.w, (ISX_NT_G & 1) != 0);
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

// Two horizontally adjacent level-0 pixels (x, x + 1) of row y.  For CV_8UC3 tiles whose two pixels
// lie inside the image (no reflected border in between) the 6 image bytes come from ONE 12-byte
// window and the 2 mask bytes from ONE 8-byte window, realigned with v_alignbyte — instead of eight
// byte loads (the level-0 kernels were bound by load-instruction issue, not by HBM).
struct U3 { unsigned x, y, z; };
struct U2 { unsigned x, y; };

template <int M, int SK>
__device__ __forceinline__ void load_src0_pair(const Src0& s, int x, int y, Px<M>& a, Px<M>& b) {
    if constexpr (SK == SK_U8) {
        const int yr = y - s.top, xr = x - s.left;
        if ((unsigned)yr < (unsigned)s.rows && xr >= 0 && xr + 1 < s.cols) {
            const unsigned io = __umul24((unsigned)yr, (unsigned)s.img_step) + __umul24((unsigned)xr, 3u) + s.imis;    // offset from img_al (steps < 2^24)
            const unsigned mo = __umul24((unsigned)yr, (unsigned)s.mask_step) + (unsigned)xr + s.mmis;       // offset from mask_al
            if ((io & ~3u) + 12u <= s.iend && (mo & ~3u) + 8u <= s.mend) {
                const U3 v = *(const U3*)(s.img_al + (io & ~3u));
                const U2 q = *(const U2*)(s.mask_al + (mo & ~3u));
                const unsigned lo = __builtin_amdgcn_alignbyte(v.y, v.x, io & 3u), hi = __builtin_amdgcn_alignbyte(v.z, v.y, io & 3u);
                const unsigned mk = __builtin_amdgcn_alignbyte(q.y, q.x, mo & 3u);
                const float inv255 = (float)(1. / 255.);
                const int a0 = lo & 255, a1 = (lo >> 8) & 255, a2 = (lo >> 16) & 255, b0 = lo >> 24, b1 = hi & 255, b2 = (hi >> 8) & 255;
                if constexpr (M == M_I16) { a.c0 = a0; a.c1 = a1; a.c2 = a2; b.c0 = b0; b.c1 = b1; b.c2 = b2; }
                else { a.c0 = (float)a0; a.c1 = (float)a1; a.c2 = (float)a2; b.c0 = (float)b0; b.c1 = (float)b1; b.c2 = (float)b2; }
                a.w = (float)(mk & 255) * inv255;
                b.w = (float)((mk >> 8) & 255) * inv255;
                return;
            }
        }
    }
    a = load_src0<M, SK>(s, x, y);
    b = load_src0<M, SK>(s, x + 1, y);
}

// The same pair in two steps, for kernels that want several rows' windows (and other loads) in flight before the first
// one is decoded: src0_pair_issue only issues the two window loads (raw.fast says whether the fast path applies),
// src0_pair_finish decodes them - or takes load_src0's path when it does not.
struct RawPair { U3 v; U2 q; unsigned sh; bool fast; };   // sh = (io & 3) | (mo & 3) << 2

template <int SK>
__device__ __forceinline__ RawPair src0_pair_issue(const Src0& s, int x, int y) {
    RawPair r;
    r.v = U3{0u, 0u, 0u}; r.q = U2{0u, 0u}; r.sh = 0u; r.fast = false;
    if constexpr (SK == SK_U8) {
        const int yr = y - s.top, xr = x - s.left;
        if ((unsigned)yr < (unsigned)s.rows && xr >= 0 && xr + 1 < s.cols) {
            const unsigned io = __umul24((unsigned)yr, (unsigned)s.img_step) + __umul24((unsigned)xr, 3u) + s.imis;
            const unsigned mo = __umul24((unsigned)yr, (unsigned)s.mask_step) + (unsigned)xr + s.mmis;
            if ((io & ~3u) + 12u <= s.iend && (mo & ~3u) + 8u <= s.mend) {
                r.v = *(const U3*)(s.img_al + (io & ~3u));
                r.q = *(const U2*)(s.mask_al + (mo & ~3u));
                r.sh = (io & 3u) | ((mo & 3u) << 2);
                r.fast = true;
            }
        }
    }
    return r;
}

// two adjacent CV_8UC3 pixels + their mask bytes out of the loaded windows
template <int M>
__device__ __forceinline__ void raw_pair_decode(const RawPair& r, Px<M>& a, Px<M>& b) {
    const unsigned lo = __builtin_amdgcn_alignbyte(r.v.y, r.v.x, r.sh & 3u), hi = __builtin_amdgcn_alignbyte(r.v.z, r.v.y, r.sh & 3u);
    const unsigned mk = __builtin_amdgcn_alignbyte(r.q.y, r.q.x, r.sh >> 2);
    const float inv255 = (float)(1. / 255.);
    const int a0 = lo & 255, a1 = (lo >> 8) & 255, a2 = (lo >> 16) & 255, b0 = lo >> 24, b1 = hi & 255, b2 = (hi >> 8) & 255;
    if constexpr (M == M_I16) { a.c0 = a0; a.c1 = a1; a.c2 = a2; b.c0 = b0; b.c1 = b1; b.c2 = b2; }
    else { a.c0 = (float)a0; a.c1 = (float)a1; a.c2 = (float)a2; b.c0 = (float)b0; b.c1 = (float)b1; b.c2 = (float)b2; }
    a.w = (float)(mk & 255) * inv255;
    b.w = (float)((mk >> 8) & 255) * inv255;
}

// Wave-level test for the window path (one ballot): the pair (x, x + 1) of row y lies inside the tile and both windows inside its
// buffers for EVERY active lane; the offsets come back for the loads.  `pre`: conditions the caller already has per lane.
__device__ __forceinline__ bool wave_pair_fast(const Src0& s, int x, int y, bool pre, unsigned& io, unsigned& mo) {
    const int yr = y - s.top, xr = x - s.left;
    io = __umul24((unsigned)yr, (unsigned)s.img_step) + __umul24((unsigned)xr, 3u) + s.imis;
    mo = __umul24((unsigned)yr, (unsigned)s.mask_step) + (unsigned)xr + s.mmis;
    const bool ok = pre & ((unsigned)yr < (unsigned)s.rows) & (xr >= 0) & (xr + 1 < s.cols) & ((io & ~3u) + 12u <= s.iend) & ((mo & ~3u) + 8u <= s.mend);
    return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}
__device__ __forceinline__ RawPair raw_pair_load(const Src0& s, unsigned io, unsigned mo) {
    RawPair r;
    r.v = *(const U3*)(s.img_al + (io & ~3u));
    r.q = *(const U2*)(s.mask_al + (mo & ~3u));
    r.sh = (io & 3u) | ((mo & 3u) << 2);
    r.fast = true;
    return r;
}

template <int M, int SK>
__device__ __forceinline__ void src0_pair_finish(const Src0& s, int x, int y, const RawPair& r, Px<M>& a, Px<M>& b) {
    if (r.fast) {
        raw_pair_decode<M>(r, a, b);
    } else {
        a = load_src0<M, SK>(s, x, y);
        b = load_src0<M, SK>(s, x + 1, y);
    }
}

template <int M, int SK>
__device__ __forceinline__ Px<M> load_any(const Src0& s0, const LevelBuf& L, int x, int y) {
    if constexpr (SK == SK_LEVEL) return load_px<M, false>(L, x, y);
    else return load_src0<M, SK>(s0, x, y);
}

// Neighbour-lane exchange for the horizontal 5-tap: whole-wavefront shifts by one lane done as DPP moves
// (wave_shr:1 / wave_shl:1, plain VALU instructions) instead of __shfl_up/__shfl_down, which compile to
// ds_bpermute_b32 and made the LDS pipe the bottleneck of k_pyr_down (12 per input row; PMC:
// SQ_ACTIVE_INST_LDS ~ 85 % of the kernel).  Lanes 0 / 63 receive 0 (they are halo lanes, unused).
__device__ __forceinline__ int dpp_from_lower_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }   // wave_shr:1
__device__ __forceinline__ int dpp_from_upper_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }   // wave_shl:1
__device__ __forceinline__ float dpp_from_lower_lane(float v) { return __int_as_float(dpp_from_lower_lane(__float_as_int(v))); }
__device__ __forceinline__ float dpp_from_upper_lane(float v) { return __int_as_float(dpp_from_upper_lane(__float_as_int(v))); }

template <int M>
__device__ __forceinline__ Px<M> shfl_up1(const Px<M>& p) {   // lane i <- lane i - 1
    Px<M> r;
    r.c0 = dpp_from_lower_lane(p.c0); r.c1 = dpp_from_lower_lane(p.c1); r.c2 = dpp_from_lower_lane(p.c2); r.w = dpp_from_lower_lane(p.w);
    return r;
}
template <int M>
__device__ __forceinline__ Px<M> shfl_down1(const Px<M>& p) {   // lane i <- lane i + 1
    Px<M> r;
    r.c0 = dpp_from_upper_lane(p.c0); r.c1 = dpp_from_upper_lane(p.c1); r.c2 = dpp_from_upper_lane(p.c2); r.w = dpp_from_upper_lane(p.w);
    return r;
}

// LDS-DMA: 16 bytes per lane from each lane's own global address to lds_wave_base + lane * 16 (gfx950
// global_load_lds_dwordx4; counted by vmcnt, the compiler drains it before the next barrier)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the [1 4 6 4 1] tap in the association pyramids.cpp uses: c*6 + (l1 + r1)*4 + l2 + r2
template <class T>
__device__ __forceinline__ T tap5(T c, T l1, T r1, T l2, T r2) { return c * 6 + (l1 + r1) * 4 + l2 + r2; }

// ------------------------------------------------------------------------------------------------
// k_pyr_down: one block (8 waves) = 62 x 16 outputs.  Lane j of every wave owns output column
// ox0 - 1 + j and loads the two input pixels (2x, 2x+1) of that column for each of its input rows;
// the other three taps of the horizontal [1 4 6 4 1] come from the neighbour lanes by wavefront
// shuffle.  Lanes 0 and 63 are halo lanes (they only feed their neighbours), so the wave needs no
// divergent edge loads; every column / row index goes through REFLECT_101, which makes the shuffled
// values correct at the image borders too.  All global loads of a wave are issued before the first
// use (5 rows x 2 records in flight per lane).  The row-filtered tile (35 x 62 records, 2-row halo
// above, 1 below... = 2*16+3 rows) is staged in LDS; the column filter runs out of LDS.
// ------------------------------------------------------------------------------------------------
constexpr int PD_TY = 16;
constexpr int PD_NR = 2 * PD_TY + 3;
constexpr int PD_OW = WAVE - 2;
constexpr int PD_WAVES = 8;

template <int M, int PLD = 0> __device__ __forceinline__ void pyr_down_columns(Px<M> (*hb)[WAVE], const LevelBuf& dst, int ox, int oy0, int lane, int wv);

// pyrDown's row filter at one output column: c = pixel 2x, l1 / r1 = pixels 2x - 1 / 2x + 1, l2 / r2 = pixels 2x - 2 / 2x + 2;
// tap5's association, the float precisions on (b, g) / (r, w) register pairs (packed fp32, each half rounded on its own)
template <int M>
__device__ __forceinline__ Px<M> pyr_down_row5(const Px<M>& c, const Px<M>& l1, const Px<M>& r1, const Px<M>& l2, const Px<M>& r2) {
    using WT = typename WorkT<M>::t;
    Px<M> h;
    if constexpr (M != M_I16) {
        const f32x2 a0 = {c.c0, c.c1}, a1 = {c.c2, c.w}, b0 = {r1.c0, r1.c1}, b1 = {r1.c2, r1.w};
        const f32x2 am0 = {l2.c0, l2.c1}, am1 = {l2.c2, l2.w}, bm0 = {l1.c0, l1.c1}, bm1 = {l1.c2, l1.w}, ap0 = {r2.c0, r2.c1}, ap1 = {r2.c2, r2.w};
        const f32x2 h0 = ((a0 * splat2(6.f) + (bm0 + b0) * splat2(4.f)) + am0) + ap0;    // tap5: c * 6 + (l1 + r1) * 4 + l2 + r2
        const f32x2 h1 = ((a1 * splat2(6.f) + (bm1 + b1) * splat2(4.f)) + am1) + ap1;
        h.c0 = h0.x; h.c1 = h0.y; h.c2 = h1.x; h.w = h1.y;
    } else {
        h.c0 = tap5<WT>(c.c0, l1.c0, r1.c0, l2.c0, r2.c0);
        h.c1 = tap5<WT>(c.c1, l1.c1, r1.c1, l2.c1, r2.c1);
        h.c2 = tap5<WT>(c.c2, l1.c2, r1.c2, l2.c2, r2.c2);
        h.w = tap5<float>(c.w, l1.w, r1.w, l2.w, r2.w);
    }
    return h;
}

// The row phase of a block: NR input rows starting at row `row0` of the source level (each through REFLECT_101), row-filtered for the
// output column `ox` of this lane, into hb[0 .. NR).  (pyr_down_block: NR = NR rows from 2 oy0 - 2; the fused level-0 + level-1 kernel
// of pyrdown_l0.inc: 41 rows.)
template <int M, int SK, int NR, int PLS = 0>
__device__ __forceinline__ void pyr_down_rows(const Src0& s0, const LevelBuf& src, int ox, int row0, Px<M> (*hb)[WAVE]) {
    constexpr int RPW = (NR + PD_WAVES - 1) / PD_WAVES;
    const int sw = (SK == SK_LEVEL) ? src.cols : s0.width;
    const int sh = (SK == SK_LEVEL) ? src.rows : s0.height;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cA = reflect101(2 * ox, sw), cB = reflect101(2 * ox + 1, sw);
    // Level 0: every row's two windows are issued before the first one is decoded (RawPair) - decoding inside the
    // loading loop made each of the wave's rows wait for its own loads, five memory latencies in a row.
    Px<M> A[RPW], B[RPW];
    RawPair rw[SK != SK_LEVEL ? RPW : 1];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        int r = wv + PD_WAVES * i;
        if (r < NR) {
            int iy = reflect101(row0 + r, sh);
            if constexpr (SK != SK_LEVEL) {
                if (cB == cA + 1) rw[i] = src0_pair_issue<SK>(s0, cA, iy);
                else rw[i].fast = false;
            } else if constexpr (PLS) {
                A[i] = load_px_planar<M, PLS == 2>(src, cA, iy);
                B[i] = load_px_planar<M, PLS == 2>(src, cB, iy);
            } else {
                A[i] = load_px<M, false>(src, cA, iy);
                B[i] = load_px<M, false>(src, cB, iy);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        int r = wv + PD_WAVES * i;
        if (r < NR) {
            if constexpr (SK != SK_LEVEL) {
                int iy = reflect101(row0 + r, sh);
                if (rw[i].fast) src0_pair_finish<M, SK>(s0, cA, iy, rw[i], A[i], B[i]);
                else { A[i] = load_src0<M, SK>(s0, cA, iy); B[i] = load_src0<M, SK>(s0, cB, iy); }
            }
            const Px<M> h = pyr_down_row5<M>(A[i], shfl_up1<M>(B[i]), B[i], shfl_up1<M>(A[i]), shfl_down1<M>(A[i]));
            hb[r][lane] = h;
        }
    }
}

template <int M, int SK, int PLS = 0, int PLD = 0>
__device__ __forceinline__ void pyr_down_block(const Src0& s0, const LevelBuf& src, const LevelBuf& dst, int bx, int by, Px<M> (*hb)[WAVE]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ox = bx * PD_OW + lane - 1, oy0 = by * PD_TY;
    pyr_down_rows<M, SK, PD_NR, PLS>(s0, src, ox, 2 * oy0 - 2, hb);
    __syncthreads();
    pyr_down_columns<M, PLD>(hb, dst, ox, oy0, lane, wv);
}

// pyrDown's column filter over five row-filtered rows + the 1/256 (the (v + 128) >> 8 of the 16-bit pyramid)
template <int M>
__device__ __forceinline__ Px<M> pyr_down_col5(const Px<M>& r0, const Px<M>& r1, const Px<M>& r2, const Px<M>& r3, const Px<M>& r4) {
    using WT = typename WorkT<M>::t;
    Px<M> o;
    if constexpr (M != M_I16) {   // packed the same way as the row filter
        const f32x2 p0[5] = {{r0.c0, r0.c1}, {r1.c0, r1.c1}, {r2.c0, r2.c1}, {r3.c0, r3.c1}, {r4.c0, r4.c1}};
        const f32x2 p1[5] = {{r0.c2, r0.w}, {r1.c2, r1.w}, {r2.c2, r2.w}, {r3.c2, r3.w}, {r4.c2, r4.w}};
        const f32x2 v0 = (((p0[2] * splat2(6.f) + (p0[1] + p0[3]) * splat2(4.f)) + p0[0]) + p0[4]) * splat2(1.f / 256.f);
        const f32x2 v1 = (((p1[2] * splat2(6.f) + (p1[1] + p1[3]) * splat2(4.f)) + p1[0]) + p1[4]) * splat2(1.f / 256.f);
        o.c0 = v0.x; o.c1 = v0.y; o.c2 = v1.x; o.w = v1.y;
    } else {
        const WT a0 = tap5<WT>(r2.c0, r1.c0, r3.c0, r0.c0, r4.c0);
        const WT a1 = tap5<WT>(r2.c1, r1.c1, r3.c1, r0.c1, r4.c1);
        const WT a2 = tap5<WT>(r2.c2, r1.c2, r3.c2, r0.c2, r4.c2);
        const float aw = tap5<float>(r2.w, r1.w, r3.w, r0.w, r4.w);
        o.c0 = (a0 + 128) >> 8; o.c1 = (a1 + 128) >> 8; o.c2 = (a2 + 128) >> 8;   // weights sum to 256: no clamp can act
        o.w = aw * (1.f / 256.f);
    }
    return o;
}

// the column filter of a block out of the row-filtered rows in LDS (after the block's barrier)
template <int M, int PLD>
__device__ __forceinline__ void pyr_down_columns(Px<M> (*hb)[WAVE], const LevelBuf& dst, int ox, int oy0, int lane, int wv) {
    const int dw = dst.cols, dh = dst.rows;
    if (lane == 0 || lane == 63 || ox >= dw) return;
#pragma unroll
    for (int i = 0; i < PD_TY / PD_WAVES; ++i) {
        int ty = wv + PD_WAVES * i, oy = oy0 + ty;
        if (oy >= dh) break;
        const Px<M> o = pyr_down_col5<M>(hb[2 * ty][lane], hb[2 * ty + 1][lane], hb[2 * ty + 2][lane], hb[2 * ty + 3][lane], hb[2 * ty + 4][lane]);
        if constexpr (PLD != 0) store_px_planar<M, PLD == 2>(dst, ox, oy, o);
        else store_px<M, false>(dst, ox, oy, o);
    }
}

template <int M, int SK>
__global__ __launch_bounds__(512) void k_pyr_down(Src0 s0, LevelBuf src, LevelBuf dst) {
    __shared__ Px<M> hb[PD_NR][WAVE];
    pyr_down_block<M, SK>(s0, src, dst, blockIdx.x, blockIdx.y, hb);
}

// ------------------------------------------------------------------------------------------------
// pyrUp of a coarse tile held in LDS.  Thread (lane, wv) owns coarse pixel (cx, cy) and produces
// the 2x2 fine block.  ct rows are coarse rows cy0-1 .. cy0+4 through the row map
// (row -1 := row 1, row h := row h-1), columns cx0-1 .. cx0+64 (clamped; edge columns use
// OpenCV's explicit edge formulas so the clamped value is never used).
// ------------------------------------------------------------------------------------------------
constexpr int UP_TY = 4;
static_assert(ISX_WINDOW_GRANULE == 2 * WAVE, "a window starts on a block column of the last collapse step");

template <int M>
struct Up4 { typename WorkT<M>::t v[2][2][3]; };  // [dy][dx][channel]

template <int M>
__device__ __forceinline__ int up_row_map(int y, int h) {
    // borderInterpolate(2*y, 2*h, REFLECT_101) / 2 for y in [-1, h]: -1 -> 1 (0 when h == 1), h -> h - 1.
    // Rows further out only feed threads that own no pixel: any valid row will do.
    return y < 0 ? (h > 1 ? 1 : 0) : min(y, h - 1);
}

template <int M> __device__ __forceinline__ void normalise(Px<M>& d);

// NORMC: the coarse level is the (not yet normalised) top level of the destination pyramid: apply
// normalizeUsingWeightMap while staging (uncovered pixels are zero by definition, see Cover)
template <int M, bool DST, bool NORMC = false>
__device__ __forceinline__ void stage_coarse(Px<M> (*ct)[WAVE + 2], const LevelBuf& coarse, int cx0, int cy0, const Cover* ccov = nullptr) {
    for (int i = threadIdx.x; i < (UP_TY + 2) * (WAVE + 2); i += 256) {
        int ry = i / (WAVE + 2), rx = i - ry * (WAVE + 2);
        int gy = up_row_map<M>(cy0 - 1 + ry, coarse.rows);
        int gx = min(max(cx0 - 1 + rx, 0), coarse.cols - 1);
        if constexpr (NORMC) {
            Px<M> d;
            if (covered(*ccov, gx, gy)) d = load_px<M, DST>(coarse, gx, gy);
            else { d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f; }
            normalise<M>(d);
            ct[ry][rx] = d;
        } else {
            ct[ry][rx] = load_px_rgb<M, DST>(coarse, gx, gy);
        }
    }
}

template <int M>
__device__ __forceinline__ Up4<M> pyr_up_2x2(Px<M> (*ct)[WAVE + 2], int lane, int wv, int cx, int cw) {
    using WT = typename WorkT<M>::t;
    WT t0[3][3], t1[3][3];  // [row][channel]
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const Px<M> sm = ct[wv + rr][lane], sc = ct[wv + rr][lane + 1], sp = ct[wv + rr][lane + 2];
        t0[rr][0] = sm.c0 + sc.c0 * 6 + sp.c0; t1[rr][0] = (sc.c0 + sp.c0) * 4;
        t0[rr][1] = sm.c1 + sc.c1 * 6 + sp.c1; t1[rr][1] = (sc.c1 + sp.c1) * 4;
        t0[rr][2] = sm.c2 + sc.c2 * 6 + sp.c2; t1[rr][2] = (sc.c2 + sp.c2) * 4;
    }
    if (cx == 0 || cx == cw - 1) {   // OpenCV's explicit edge formulas (a different fp32 association): two lanes per row
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const Px<M> sm = ct[wv + rr][lane], sc = ct[wv + rr][lane + 1], sp = ct[wv + rr][lane + 2];
            const WT m[3] = {sm.c0, sm.c1, sm.c2}, c[3] = {sc.c0, sc.c1, sc.c2}, p[3] = {sp.c0, sp.c1, sp.c2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (cw == 1) { t0[rr][k] = c[k] * 8; t1[rr][k] = c[k] * 8; }
                else if (cx == 0) { t0[rr][k] = c[k] * 6 + p[k] * 2; t1[rr][k] = (c[k] + p[k]) * 4; }
                else { t0[rr][k] = m[k] + c[k] * 7; t1[rr][k] = c[k] * 8; }
            }
        }
    }
    Up4<M> u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        WT e0 = t0[0][k] + t0[1][k] * 6 + t0[2][k], e1 = t1[0][k] + t1[1][k] * 6 + t1[2][k];
        WT o0 = (t0[1][k] + t0[2][k]) * 4, o1 = (t1[1][k] + t1[2][k]) * 4;
        if constexpr (M == M_I16) {
            // saturate_cast<short>((v + 32) >> 6): the kernel weights of every output sum to 64 (edge formulas included) and the inputs are
            // shorts, so the result lies in [-32768, 32767] by itself - the clamp OpenCV applies can never act and is not issued
            u.v[0][0][k] = (e0 + 32) >> 6; u.v[0][1][k] = (e1 + 32) >> 6;
            u.v[1][0][k] = (o0 + 32) >> 6; u.v[1][1][k] = (o1 + 32) >> 6;
        } else {
            u.v[0][0][k] = e0 * (1.f / 64.f); u.v[0][1][k] = e1 * (1.f / 64.f);
            u.v[1][0][k] = o0 * (1.f / 64.f); u.v[1][1][k] = o1 * (1.f / 64.f);
        }
    }
    return u;
}

// The same pyrUp with the fp32 arithmetic packed (v_pk_mul_f32 / v_pk_add_f32 round each half on its own: the bits of pyr_up_2x2).
// Pairs are chosen so that nothing is shuffled and no half is wasted: (b, g) of a record is a register pair as loaded; the r channel
// pairs the even-column and odd-column results (t0, t1) of a row, which is exactly what the vertical pass combines alike.
struct UpPk {
    f32x2 a[2][2];   // [dy][dx] = (b, g) of fine pixel (dy, dx)
    f32x2 c[2];      // [dy]     = r of fine pixels (dy, 0) and (dy, 1)
};

template <int M>
__device__ __forceinline__ UpPk pyr_up_2x2_pk(Px<M> (*ct)[WAVE + 2], int lane, int wv, int cx, int cw) {
    static_assert(M != M_I16, "packed pyrUp: float work types only");
    f32x2 t0a[3], t1a[3], tc[3];      // tc[rr] = (t0, t1) of the r channel
    const bool edge = cx == 0 || cx == cw - 1;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const Px<M> sm = ct[wv + rr][lane], sc = ct[wv + rr][lane + 1], sp = ct[wv + rr][lane + 2];
        const f32x2 ma = {sm.c0, sm.c1}, ca = {sc.c0, sc.c1}, pa = {sp.c0, sp.c1};
        t0a[rr] = (ma + ca * splat2(6.f)) + pa; t1a[rr] = (ca + pa) * splat2(4.f);
        tc[rr].x = (sm.c2 + sc.c2 * 6.f) + sp.c2; tc[rr].y = (sc.c2 + sp.c2) * 4.f;
        if (edge) {   // OpenCV's explicit edge formulas (a different fp32 association): two lanes per row
            if (cw == 1) { t0a[rr] = ca * splat2(8.f); t1a[rr] = t0a[rr]; tc[rr].x = sc.c2 * 8.f; tc[rr].y = tc[rr].x; }
            else if (cx == 0) { t0a[rr] = ca * splat2(6.f) + pa * splat2(2.f); tc[rr].x = sc.c2 * 6.f + sp.c2 * 2.f; }
            else { t0a[rr] = ma + ca * splat2(7.f); t1a[rr] = ca * splat2(8.f); tc[rr].x = sm.c2 + sc.c2 * 7.f; tc[rr].y = sc.c2 * 8.f; }
        }
    }
    UpPk u;
    const f32x2 s64 = splat2(1.f / 64.f);
    u.a[0][0] = ((t0a[0] + t0a[1] * splat2(6.f)) + t0a[2]) * s64; u.a[0][1] = ((t1a[0] + t1a[1] * splat2(6.f)) + t1a[2]) * s64;
    u.a[1][0] = ((t0a[1] + t0a[2]) * splat2(4.f)) * s64;          u.a[1][1] = ((t1a[1] + t1a[2]) * splat2(4.f)) * s64;
    u.c[0] = ((tc[0] + tc[1] * splat2(6.f)) + tc[2]) * s64;
    u.c[1] = ((tc[1] + tc[2]) * splat2(4.f)) * s64;
    return u;
}

// dst += cast(lap * w), dstW += w   (MultiBandBlender::feed accumulate loop)
template <int M>
__device__ __forceinline__ void accumulate(const LevelBuf& dst, int x, int y, typename WorkT<M>::t l0,
                                           typename WorkT<M>::t l1, typename WorkT<M>::t l2, float w, bool have) {
    Px<M> d;
    if (have) d = load_px<M, true>(dst, x, y);
    else { d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f; }
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
__device__ __forceinline__ void lap_acc_block(Px<M> (*ct)[WAVE + 2], const Src0& s0, const LevelBuf& fine, const LevelBuf& coarse,
                                              const LevelBuf& dst, int x_tl, int y_tl, const Cover& cov, int bx, int by, int ck = 0) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cx0 = bx * WAVE, cy0 = by * UP_TY;
    const int cx = cx0 + lane, cy = cy0 + wv;
    const bool mine = cx < coarse.cols && cy < coarse.rows;
    // rectangle corners are even at every level below the top, so the 2x2 block is covered as a whole
    const bool have = mine && covered(cov, x_tl + 2 * cx, y_tl + 2 * cy, ck);
    // every load of the block is issued before its one barrier (the order used to be: stage the coarse tile, barrier,
    // then each fine row loaded and decoded in turn, then the destination records read one by one inside the
    // accumulate - four memory latencies in a row per wave): the thread's fine pixels (level 0: raw windows), its four
    // destination records, then the coarse tile
    Px<M> gg[2][2], dd[2][2];
    RawPair rw[SK != SK_LEVEL ? 2 : 1];
    if (mine) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            if constexpr (SK != SK_LEVEL) rw[dy] = src0_pair_issue<SK>(s0, 2 * cx, 2 * cy + dy);
            else { gg[dy][0] = load_px<M, false>(fine, 2 * cx, 2 * cy + dy); gg[dy][1] = load_px<M, false>(fine, 2 * cx + 1, 2 * cy + dy); }
        }
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            if (have) dd[dy][dx] = load_px<M, true>(dst, x_tl + 2 * cx + dx, y_tl + 2 * cy + dy);
            else { dd[dy][dx].c0 = 0; dd[dy][dx].c1 = 0; dd[dy][dx].c2 = 0; dd[dy][dx].w = 0.f; }
        }
    stage_coarse<M, false>(ct, coarse, cx0, cy0);
    __syncthreads();
    if (!mine) return;
    if constexpr (SK != SK_LEVEL) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) src0_pair_finish<M, SK>(s0, 2 * cx, 2 * cy + dy, rw[dy], gg[dy][0], gg[dy][1]);
    }
    Up4<M> u = pyr_up_2x2<M>(ct, lane, wv, cx, coarse.cols);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const Px<M> g = gg[dy][dx];
            Px<M> d = dd[dy][dx];
            if constexpr (M == M_I16) {  // cv::subtract saturates; dst += static_cast<short>(lap * w), wrapping
                d.c0 = wrap_s16(d.c0 + f2s_x86((float)sat_s16(g.c0 - u.v[dy][dx][0]) * g.w));
                d.c1 = wrap_s16(d.c1 + f2s_x86((float)sat_s16(g.c1 - u.v[dy][dx][1]) * g.w));
                d.c2 = wrap_s16(d.c2 + f2s_x86((float)sat_s16(g.c2 - u.v[dy][dx][2]) * g.w));
            } else {
                d.c0 = d.c0 + (g.c0 - u.v[dy][dx][0]) * g.w; d.c1 = d.c1 + (g.c1 - u.v[dy][dx][1]) * g.w; d.c2 = d.c2 + (g.c2 - u.v[dy][dx][2]) * g.w;
            }
            d.w = d.w + g.w;
            store_px<M, true>(dst, x_tl + 2 * cx + dx, y_tl + 2 * cy + dy, d);
        }
}

template <int M, int SK>
__global__ __launch_bounds__(256) void k_lap_acc(Src0 s0, LevelBuf fine, LevelBuf coarse, LevelBuf dst, int x_tl, int y_tl, Cover cov) {
    __shared__ Px<M> ct[UP_TY + 2][WAVE + 2];
    lap_acc_block<M, SK>(ct, s0, fine, coarse, dst, x_tl, y_tl, cov, blockIdx.x, blockIdx.y);
}

// All levels of one feed in ONE launch: the per-level accumulations are independent of each other
// once the Gaussian chain exists, so a 1-D grid is cut into per-level block ranges (level 0 first,
// the small levels fill the tail of the launch instead of paying a kernel boundary each).
constexpr int ACC_MAXL = 8;
struct AccArgs {
    Src0 s0;
    LevelBuf g[ACC_MAXL + 1];
    LevelBuf dst[ACC_MAXL + 1];
    int blk_start[ACC_MAXL + 2];   // first block of level k; [L + 1] = grid size
    int gw[ACC_MAXL + 1];          // blocks per row of level k
    int L, x_tl, y_tl;
    Cover cov0;                    // level-0 rectangles of the earlier feeds
};

template <int M, int SK>
__global__ __launch_bounds__(256) void k_lap_acc_all(AccArgs a) {
    __shared__ Px<M> ct[UP_TY + 2][WAVE + 2];
    const int bid = blockIdx.x;
    int k = 0;
#pragma unroll
    for (int i = 1; i <= ACC_MAXL; ++i) k += (i <= a.L && bid >= a.blk_start[i]) ? 1 : 0;
    const int local = bid - a.blk_start[k];
    const int by = local / a.gw[k], bx = local - by * a.gw[k];
    const int xt = a.x_tl >> k, yt = a.y_tl >> k;
    if (k == a.L) {   // top level: Laplacian == Gaussian
        const int x = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
        if (x >= a.g[k].cols || y >= a.g[k].rows) return;
        Px<M> g;
        if (k == 0) g = load_any<M, SK>(a.s0, a.g[0], x, y); else g = load_px<M, false>(a.g[k], x, y);
        accumulate<M>(a.dst[k], xt + x, yt + y, g.c0, g.c1, g.c2, g.w, covered(a.cov0, xt + x, yt + y, k));
    } else if (k == 0) {
        lap_acc_block<M, SK>(ct, a.s0, a.g[0], a.g[1], a.dst[0], xt, yt, a.cov0, bx, by, 0);
    } else {
        lap_acc_block<M, SK_LEVEL>(ct, a.s0, a.g[k], a.g[k + 1], a.dst[k], xt, yt, a.cov0, bx, by, k);
    }
}

// top level: the Laplacian pyramid's last level is the Gaussian level itself
template <int M, int SK>
__global__ __launch_bounds__(256) void k_top_acc(Src0 s0, LevelBuf top, LevelBuf dst, int x_tl, int y_tl, int rows, int cols, Cover cov) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    Px<M> g = load_any<M, SK>(s0, top, x, y);
    accumulate<M>(dst, x_tl + x, y_tl + y, g.c0, g.c1, g.c2, g.w, covered(cov, x_tl + x, y_tl + y));
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
    unsigned char* img; size_t img_step; int img_f32;   // img_f32: 0 = CV_16SC3, 1 = CV_32FC3, 2 = CV_8UC3 (blend + convertTo(CV_8U))
    unsigned char* mask; size_t mask_step;
    int rows, cols;  // dst_roi_final_ size
    int vec;         // image rows 4-byte aligned and mask rows 2-byte aligned: pair stores allowed
    int bx0;         // column window (isx_blender_set_window): first block column of this launch, 0 without a window
    int grp, gx, gy; // grp > 0: a 1-D launch in the XCD-aware block order of isx_device.hpp (xcd_block) over gx x gy blocks
    unsigned xmagic; // xcd_magic(grp, gx)
    int band;        // > 0: xcd_band_block's mapping, block rows per XCD band
    int rec12;       // deferred cycle, k_collapse_roll runs the last step: out_1 holds dense 12-byte image records (no fourth dword nobody reads):
                     // the level-1 step (k_collapse_gather, not FINE0) writes them, k_collapse_roll reads them
};

// saturate_cast<short / uchar>(float) = sat(cvRound(v)).  BOUNDED: the caller guarantees |v| < 2^31 (blends of CV_8UC3 / CV_16SC3 tiles
// stay below 2^21), so the x86 "integer indefinite" case of cvRound cannot occur and clamping first gives the same integer in three
// instructions (clamp, round-half-even, convert; NaN clamps to the lower bound as cvRound's INT_MIN saturates to it).
template <bool BOUNDED>
__device__ __forceinline__ int f2s16_sat(float v) {
    if constexpr (BOUNDED) return (int)__builtin_rintf(__builtin_amdgcn_fmed3f(v, -32768.f, 32767.f));
    else return sat_s16(cvround_x86(v));
}
template <bool BOUNDED>
__device__ __forceinline__ int f2u8_sat(float v) {
    if constexpr (BOUNDED) return (int)__builtin_rintf(__builtin_amdgcn_fmed3f(v, 0.f, 255.f));
    else return sat_u8(cvround_x86(v));
}

template <int M, bool BOUNDED = false>
__device__ __forceinline__ void write_final(const OutMat& o, int x, int y, const Px<M>& d) {
    if (x >= o.cols || y >= o.rows) return;      // crop to dst_roi_final_
    bool on = d.w > WEIGHT_EPS;                  // compare(w0, WEIGHT_EPS, CMP_GT)
    // blend() checks rows * step < 2^32 and step < 2^24 for both mats: 32-bit offsets from full-rate 24-bit multiplies
    if (o.mask) o.mask[__umul24((unsigned)y, (unsigned)o.mask_step) + (unsigned)x] = on ? 255 : 0;
    if (o.img_f32 == 2) {   // result.convertTo(CV_8U): saturate_cast<uchar>(short) / saturate_cast<uchar>(cvRound(float))
        unsigned char* q = o.img + (__umul24((unsigned)y, (unsigned)o.img_step) + (unsigned)x * 3u);
        if constexpr (M == M_I16) {
            q[0] = on ? (unsigned char)sat_u8(d.c0) : 0; q[1] = on ? (unsigned char)sat_u8(d.c1) : 0; q[2] = on ? (unsigned char)sat_u8(d.c2) : 0;
        } else {
            q[0] = on ? (unsigned char)f2u8_sat<BOUNDED>(d.c0) : 0;
            q[1] = on ? (unsigned char)f2u8_sat<BOUNDED>(d.c1) : 0;
            q[2] = on ? (unsigned char)f2u8_sat<BOUNDED>(d.c2) : 0;
        }
    } else if (o.img_f32) {
        float* q = (float*)(o.img + (__umul24((unsigned)y, (unsigned)o.img_step) + (unsigned)x * 12u));
        q[0] = on ? (float)d.c0 : 0.f; q[1] = on ? (float)d.c1 : 0.f; q[2] = on ? (float)d.c2 : 0.f;
    } else {
        short* q = (short*)(o.img + (__umul24((unsigned)y, (unsigned)o.img_step) + (unsigned)x * 6u));
        if constexpr (M == M_I16) {
            q[0] = on ? (short)d.c0 : 0; q[1] = on ? (short)d.c1 : 0; q[2] = on ? (short)d.c2 : 0;
        } else {  // saturate_cast<short>(float)
            q[0] = on ? (short)f2s16_sat<BOUNDED>(d.c0) : 0;
            q[1] = on ? (short)f2s16_sat<BOUNDED>(d.c1) : 0;
            q[2] = on ? (short)f2s16_sat<BOUNDED>(d.c2) : 0;
        }
    }
}

// two horizontally adjacent result pixels (x even): CV_16SC3 = 12 contiguous bytes -> three dword
// stores when the row is 4-byte aligned (OutMat::vec), mask = one 16-bit store
// ALLIN: the caller has established for the whole wave (one ballot) that the vector path applies and both pixels are inside
template <int M, bool BOUNDED = false, bool ALLIN = false>
__device__ __forceinline__ void write_final_pair(const OutMat& o, int x, int y, const Px<M>& d0, const Px<M>& d1) {
    if constexpr (!ALLIN) {
        if (!o.vec || o.img_f32 || x + 1 >= o.cols) {
            write_final<M, BOUNDED>(o, x, y, d0);
            write_final<M, BOUNDED>(o, x + 1, y, d1);
            return;
        }
        if (y >= o.rows) return;
    }
    const bool on0 = d0.w > WEIGHT_EPS, on1 = d1.w > WEIGHT_EPS;
    unsigned* q = (unsigned*)(o.img + (__umul24((unsigned)y, (unsigned)o.img_step) + (unsigned)x * 6u));
    if constexpr (BOUNDED && M != M_I16) {
        // |v| < 2^31 and never NaN: round-half-even, convert, and let v_cvt_pk_i16_i32 saturate and pack two values at once
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const float f[6] = {on0 ? (float)d0.c0 : 0.f, on0 ? (float)d0.c1 : 0.f, on0 ? (float)d0.c2 : 0.f, on1 ? (float)d1.c0 : 0.f, on1 ? (float)d1.c1 : 0.f, on1 ? (float)d1.c2 : 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const s16x2 pk = __builtin_amdgcn_cvt_pk_i16((int)__builtin_rintf(f[2 * i]), (int)__builtin_rintf(f[2 * i + 1]));
            q[i] = __builtin_bit_cast(unsigned, pk);
        }
    } else {
        int v[6];
        if constexpr (M == M_I16) { v[0] = d0.c0; v[1] = d0.c1; v[2] = d0.c2; v[3] = d1.c0; v[4] = d1.c1; v[5] = d1.c2; }
        else {
            v[0] = f2s16_sat<BOUNDED>(d0.c0); v[1] = f2s16_sat<BOUNDED>(d0.c1); v[2] = f2s16_sat<BOUNDED>(d0.c2);
            v[3] = f2s16_sat<BOUNDED>(d1.c0); v[4] = f2s16_sat<BOUNDED>(d1.c1); v[5] = f2s16_sat<BOUNDED>(d1.c2);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { if (!on0) v[i] = 0; if (!on1) v[3 + i] = 0; }
        q[0] = (unsigned)(v[0] & 0xffff) | ((unsigned)v[1] << 16);
        q[1] = (unsigned)(v[2] & 0xffff) | ((unsigned)v[3] << 16);
        q[2] = (unsigned)(v[4] & 0xffff) | ((unsigned)v[5] << 16);
    }
    if (o.mask) *(unsigned short*)(o.mask + (__umul24((unsigned)y, (unsigned)o.mask_step) + (unsigned)x)) = (unsigned short)((on0 ? 255u : 0u) | (on1 ? 0xff00u : 0u));
}

// top level of blend(): normalise in place (or straight to the caller's mat when num_bands == 0)
template <int M, bool FINAL>
__global__ __launch_bounds__(256) void k_norm_top(LevelBuf lv, OutMat out, Cover cov) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= lv.cols || y >= lv.rows) return;
    Px<M> d;
    if (covered(cov, x, y)) d = load_px<M, true>(lv, x, y);
    else { d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f; }
    normalise<M>(d);
    if constexpr (FINAL) write_final<M>(out, x, y, d);
    else store_px<M, true>(lv, x, y, d);
}

// out_{k-1} = sat(pyrUp(out_k) + normalise(dst_{k-1}))   (restoreImageFromLaplacePyr, fused normalise)
template <int M, bool FINAL, bool NORMC>
__global__ __launch_bounds__(256) void k_collapse(LevelBuf coarse, LevelBuf fine, OutMat out, Cover cov, Cover ccov) {
    __shared__ Px<M> ct[UP_TY + 2][WAVE + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cx0 = blockIdx.x * WAVE, cy0 = blockIdx.y * UP_TY;
    stage_coarse<M, true, NORMC>(ct, coarse, cx0, cy0, &ccov);
    __syncthreads();
    const int cx = cx0 + lane, cy = cy0 + wv;
    if (cx >= coarse.cols || cy >= coarse.rows) return;
    Up4<M> u = pyr_up_2x2<M>(ct, lane, wv, cx, coarse.cols);
    const bool have = covered(cov, 2 * cx, 2 * cy);   // fine-level rectangles have even corners
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int fy = 2 * cy + dy;
        if constexpr (FINAL) { if (fy >= out.rows) continue; }
        Px<M> dd[2];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int fx = 2 * cx + dx;
            Px<M> d;
            if (have && (!FINAL || fx < out.cols)) d = load_px<M, true>(fine, fx, fy);
            else { d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f; }
            normalise<M>(d);
            if constexpr (M == M_I16) {  // cv::add saturates
                d.c0 = sat_s16(u.v[dy][dx][0] + d.c0); d.c1 = sat_s16(u.v[dy][dx][1] + d.c1); d.c2 = sat_s16(u.v[dy][dx][2] + d.c2);
            } else {
                d.c0 = u.v[dy][dx][0] + d.c0; d.c1 = u.v[dy][dx][1] + d.c1; d.c2 = u.v[dy][dx][2] + d.c2;
            }
            dd[dx] = d;
        }
        if constexpr (FINAL) write_final_pair<M>(out, 2 * cx, fy, dd[0], dd[1]);
        else { store_px<M, true>(fine, 2 * cx, fy, dd[0]); store_px<M, true>(fine, 2 * cx + 1, fy, dd[1]); }
    }
}

// ------------------------------------------------------------------------------------------------
// Deferred mode (opt-in, isx_blender_set_deferred_level0): feed() only records the tile; blend()
//   1. builds every tile's Gaussian chain, all tiles of a level in one launch   (k_pyr_down_multi)
//   2. gathers + normalises the top level inside the first collapse step      (top_px, TOP)
//   3. collapses: out_{k-1} = sat(pyrUp(out_k) + norm(SUM_t cast(lap_{k-1,t} * w_{k-1,t})))
//      where the sum over the tiles covering a pixel runs IN REGISTERS, in feed order, from
//      lap = G_{k-1,t} - pyrUp(G_{k,t})                                         (k_collapse_gather)
// The destination Laplacian / weight pyramid (32 B/px of read-modify-write per feed + 16 B/px
// read back by blend) never exists in HBM.  Operations and their order per pixel are those of the
// eager path (0 + a == a exactly), so the results are identical.
// ------------------------------------------------------------------------------------------------
constexpr int DEF_MAX = 20;     // kernel arguments hold one TileSet: 20 tiles keep a launch's argument block below the 4 KB limit
constexpr int DEF_REC_MAX = 4096;   // tiles a deferred cycle records; with more than DEF_MAX of them blend() works in column strips (run_blend_deferred_strips)
struct TileSet {                 // per-tile views of one pyramid level pair, indexed by the (uniform) tile id
    int n;
    int q8;                      // the tiles' planar level 1 holds Q8 records (load_px_planar; read by the level-1 collapse step and the last step)
    Src0 s0[DEF_MAX];            // level-0 view (used where the fine / source level is level 0)
    LevelBuf fine[DEF_MAX];      // G_{k-1,t}  (unused when the fine level is level 0)
    LevelBuf coarse[DEF_MAX];    // G_{k,t}
    int x_tl[DEF_MAX], y_tl[DEF_MAX], w[DEF_MAX], h[DEF_MAX];   // tile rectangle at the FINE level, dst_roi_ coordinates
    int bx_lo[DEF_MAX], bx_hi[DEF_MAX];                          // k_pyr_down_multi: block columns of the tile's destination level to produce (column window)
};

static_assert(sizeof(TileSet) + 2 * sizeof(LevelBuf) + sizeof(OutMat) <= 4096, "k_collapse_gather's arguments exceed the kernel-argument limit");

// ---- more than DEF_MAX tiles in one chain (round 5): the same per-tile views as a TABLE IN DEVICE MEMORY -------------------------------
// A cycle of more than 20 tiles used to be cut into column strips of at most 20 (run_blend_deferred_strips), and tiles stacked more than 20 deep
// over one strip fell back to the eager destination pyramid.  TileTab is TileSet with its arrays behind a pointer: one 160-byte TileDesc per
// tile, written by the host once per launch and geometry (DevTable: only what changed since the previous blend is uploaded - nothing in the
// steady state of a fixed rig), read by the kernels with SCALAR loads exactly as the kernel-argument segment is (a uniform index into constant
// address space: s_load_dword*).  The kernel bodies are templates over the view type and index it with the same expressions (ts.x_tl[t],
// ts.coarse[t].cols ...), so both forms run the same code on the same values: identical bits.
// A wave of a collapse step must not look at every tile of a long panorama to find the two or three that reach it: the table carries, per
// 2^gshift columns of the step's fine level, the index range [first, last) of the tiles whose rectangles meet those columns (feed order is
// kept inside the range; tiles of the range that do not reach the wave are skipped by the same tests as before).
struct TileDesc { Src0 s0; LevelBuf fine, coarse; int x_tl, y_tl, w, h, bx_lo, bx_hi; };
static_assert(sizeof(TileDesc) % 16 == 0, "TileDesc records are copied and loaded in 16-byte pieces");
#define ISX_AS4 __attribute__((address_space(4)))
// a field of a record in constant address space, dword by dword (what is not used is never loaded; neighbours merge into s_load_dwordx2/4/8)
template <class T>
__device__ __forceinline__ T ld_const(const void* p) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    T v;
    const ISX_AS4 unsigned* s = (const ISX_AS4 unsigned*)p;
    unsigned* o = reinterpret_cast<unsigned*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) o[i] = s[i];
    return v;
}
// A pointer that was LOADED (from the table) is a generic pointer to the compiler, and generic pointers make FLAT loads - counted on lgkmcnt as
// well as vmcnt, so every LDS wait of a kernel drains them too (collapse_roll.inc, gcp / ROLL_G, tells the same story about pointers that went
// through an asm statement).  Pointers that arrive in the kernel arguments (TileSet) are known to be global.  The table's pointers are said to
// be global where they are loaded: a round trip through address space 1 that the optimiser cannot fold away (a plain cast pair is folded, and
// __builtin_assume(!is_shared && !is_private) is not picked up: both measured) - the two halves go through v_readfirstlane, which is free for a
// value that already sits in scalar registers (a tile index is wave-uniform by construction and its descriptor comes from s_load; where the
// compiler cannot see that, inside a per-lane branch, it is two instructions) and from whose result the address-space inference types every
// access behind it (round 6: 10 - 54 flat_load per *_tab kernel before, none after; tools/isa_flat.py, tests/test_isa_flat.py).
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (T*)(__attribute__((address_space(1))) T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void mark_global(int&) {}
__device__ __forceinline__ void mark_global(Src0& s) { s.img = as_global(s.img); s.mask = as_global(s.mask); s.img_al = as_global(s.img_al); s.mask_al = as_global(s.mask_al); }
__device__ __forceinline__ void mark_global(LevelBuf& l) { l.img = as_global(l.img); l.wgt = as_global(l.wgt); }
template <class T, size_t OFF>
struct TabField {        // ts.field[t] of a TileTab, by value
    const char* base;
    __device__ __forceinline__ T operator[](int t) const { T v = ld_const<T>(base + (size_t)(unsigned)t * sizeof(TileDesc) + OFF); mark_global(v); return v; }
};
struct TileTab {
    int n, gshift, nrng, q8;      // q8: as TileSet::q8
    const int2* rng;         // per 2^gshift fine-level columns: [first, last) tile indices
    TabField<Src0, offsetof(TileDesc, s0)> s0;
    TabField<LevelBuf, offsetof(TileDesc, fine)> fine;
    TabField<LevelBuf, offsetof(TileDesc, coarse)> coarse;
    TabField<int, offsetof(TileDesc, x_tl)> x_tl;
    TabField<int, offsetof(TileDesc, y_tl)> y_tl;
    TabField<int, offsetof(TileDesc, w)> w;
    TabField<int, offsetof(TileDesc, h)> h;
    TabField<int, offsetof(TileDesc, bx_lo)> bx_lo;
    TabField<int, offsetof(TileDesc, bx_hi)> bx_hi;
};
// the tiles a block / wave that covers the fine-level columns [x0, x1) has to look at: [tb, te) narrowed (TileSet: all of them, as always)
__device__ __forceinline__ void tile_range(const TileSet&, int, int, int&, int&) {}
__device__ __forceinline__ void tile_range(const TileTab& ts, int x0, int x1, int& tb, int& te) {
    if (ts.nrng <= 0) return;
    const int b0 = min(max(x0, 0) >> ts.gshift, ts.nrng - 1), b1 = min(max(x1 - 1, 0) >> ts.gshift, ts.nrng - 1);
    int lo = 1 << 30, hi = 0;
    for (int b = b0; b <= b1; ++b) {
        const int2 r = ld_const<int2>(ts.rng + b);
        lo = min(lo, r.x); hi = max(hi, r.y);
    }
    tb = max(tb, min(lo, hi)); te = min(te, hi);
    if (te < tb) te = tb;
}

// PLS: the source level is PLANAR (level 1 of the deferred cycle, see load_px_planar); 2: in Q8 records
// FeedPub: blend() of a cycle with narrowed tiles (pyrdown_l0.inc) hands their violation words to the host with the FIRST launch of its chain - this
// one, when level 1 came from feed() - instead of a launch of its own (k_feed_publish: 4 us of a 0.32 ms step); pin == nullptr: nothing to publish
struct FeedPub { unsigned* state; int n; int* pin; int seq; };
// run by the whole first wave of block 0 (lanes stride the words; a table cycle may hold 4096 tiles: one thread did n dependent atomic loads
// and stores while the host polled - ADVICE r5)
__device__ __forceinline__ void feed_publish_words(const FeedPub& fp) {
    const int lane = threadIdx.x & 63;
    unsigned v = 0u;
    for (int i = lane; i < fp.n; i += 64) {
        v |= __hip_atomic_load(&fp.state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&fp.state[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool any = __ballot(v != 0u) != 0ull;
    if (lane == 0) {
        __hip_atomic_store(&fp.pin[1], any ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&fp.pin[0], fp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <int M, int SK, int PLS, class TS>
__device__ __forceinline__ void pyr_down_multi_body(const TS& ts, const FeedPub& fp) {
    if (fp.pin != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64) feed_publish_words(fp);
    // blockIdx.z = tile; ts.fine = source level (or s0 when SK != SK_LEVEL), ts.coarse = destination level
    const int t = blockIdx.z;
    const LevelBuf dst = ts.coarse[t];
    if ((int)blockIdx.x * PD_OW >= dst.cols || (int)blockIdx.y * PD_TY >= dst.rows || (int)blockIdx.x < ts.bx_lo[t] || (int)blockIdx.x >= ts.bx_hi[t]) return;
    __shared__ Px<M> hb[PD_NR][WAVE];
    pyr_down_block<M, SK, PLS>(ts.s0[t], ts.fine[t], dst, blockIdx.x, blockIdx.y, hb);
}
template <int M, int SK, int PLS = 0>
__global__ __launch_bounds__(512) void k_pyr_down_multi(TileSet ts, FeedPub fp) { pyr_down_multi_body<M, SK, PLS>(ts, fp); }
template <int M, int SK, int PLS = 0>
__global__ __launch_bounds__(512) void k_pyr_down_multi_tab(TileTab ts, FeedPub fp) { pyr_down_multi_body<M, SK, PLS>(ts, fp); }

// top level of the pyramid: out_L = norm(SUM_t cast(G_{L,t} * W_{L,t})) at level-L pixel (x, y); the tile rectangles of
// ts are those of level L - sh (the first collapse step passes its fine level, sh = 1).
template <int M, class TS>
__device__ __forceinline__ Px<M> top_px(const TS& ts, int tb, int te, int x, int y, int sh) {
    Px<M> d; d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f;
    for (int t = tb; t < te; ++t) {
        const int lx = x - (ts.x_tl[t] >> sh), ly = y - (ts.y_tl[t] >> sh);
        const LevelBuf cl = ts.coarse[t];
        if ((unsigned)lx < (unsigned)cl.cols && (unsigned)ly < (unsigned)cl.rows) {
            Px<M> g = load_px<M, false>(cl, lx, ly);
            if constexpr (M == M_I16) {
                d.c0 = wrap_s16(d.c0 + f2s_x86((float)g.c0 * g.w)); d.c1 = wrap_s16(d.c1 + f2s_x86((float)g.c1 * g.w)); d.c2 = wrap_s16(d.c2 + f2s_x86((float)g.c2 * g.w));
            } else { d.c0 = d.c0 + g.c0 * g.w; d.c1 = d.c1 + g.c1 * g.w; d.c2 = d.c2 + g.c2 * g.w; }
            d.w = d.w + g.w;
        }
    }
    normalise<M>(d);
    return d;
}

// FINE0: the fine level is level 0 = the caller's tiles (read through the copyMakeBorder maps) and the
// result goes to the caller's mats (crop, mask, zero fill); otherwise fine = G_{k-1,t} and the result is
// stored as level k-1 of the collapsed pyramid.
// TOP: this is the first collapse step (k = L): out_L is not read from memory but gathered while its coarse tile is
// staged (top_px), so the top level of the collapsed pyramid is never materialised and its launch disappears.
// Optional instrumentation (-DISX_PHASE_TIMING, tools/phase_probe.sh): s_memtime stamps at the phase boundaries of the last
// collapse step, summed per wave in registers and flushed with one set of atomics at exit (spread over 1024 slots) - the
// measurement that showed a wave spending half its life getting its loads issued.  Compiles to nothing otherwise.
#ifdef ISX_PHASE_TIMING
__device__ unsigned long long g_phase[1024][12];
#define PT_DECL unsigned long long ph__[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tprev__ = __builtin_amdgcn_s_memtime()
#define PT(k) do { if (FINE0) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); ph__[k] += now__ - tprev__; tprev__ = now__; } } while (0)
#define PT_FLUSH do { if (FINE0 && (threadIdx.x & 63) == 0) { const int slot__ = (blockIdx.x * 4 + (threadIdx.x >> 6) + blockIdx.y * 37) & 1023; \
        for (int k__ = 0; k__ < 11; ++k__) atomicAdd(&g_phase[slot__][k__], ph__[k__]); atomicAdd(&g_phase[slot__][11], 1ull); } } while (0)
#else
#define PT_DECL do { } while (0)
#define PT(k) do { } while (0)
#define PT_FLUSH do { } while (0)
#endif
// The tiles [tb, te) of ts are those of ONE mosaic (all of them in a single blend; one mosaic's share in a batched launch, see
// BatchOut): the body of a collapse step for the block (blockIdx.x, blockIdx.y) of that mosaic.
template <int M, int SK, bool FINE0, bool TOP, class TS>
__device__ __forceinline__ void collapse_gather_body(const TS& ts, int tb, int te, const LevelBuf& coarse_out, const LevelBuf& fine_out, const OutMat& out) {
    using WT = typename WorkT<M>::t;
    // Tiles are taken G at a time: their coarse tiles (and, in the last round, that of out_k) are staged in ONE phase —
    // every global load of the round in flight together, the fine-level pixels included, one barrier pair per round
    // instead of one per source.  G = 2 (a pair of tiles = a single round) for the upper steps; the last step keeps
    // G = 1: its level-0 decode needs the registers (152 VGPRs = 3 waves per SIMD with G = 2, measured slower).
    constexpr int G = FINE0 ? 1 : 2;
    // fp32 pyramids hold 16-byte records: their coarse tiles go from HBM to LDS by LDS-DMA (global_load_lds_dwordx4: lane i
    // of a wave fills entry base + i, which is exactly the flat staging index below) - no staging registers, no ds_write
    // pass.  The DMA lands asynchronously, so a buffer must not be re-staged while a slower wave still reads it: the tiles'
    // buffers alternate between rounds (a wave that issues round r has passed the barrier of round r - 1, which every wave
    // reaches only after its reads of round r - 2).
    // DMA_T: the tiles' Gaussian levels are 16-byte register records (fp32 and I16; the F16 tile levels are 8-byte records:
    // register staging); DMA_O: so is out_k.
    constexpr bool OUT_DST = M != M_I16;     // format of the out_k levels: the destination record, except I16 (see run_blend_deferred_t)
    constexpr bool DMA_T = (M == M_F32 || M == M_I16) && !TOP, DMA_O = !TOP;   // out_k is a 16-byte record in every precision
    constexpr int NB = DMA_T ? 2 * G + 1 : G + 1;     // the last buffer is out_k's
    __shared__ Px<M> ct[NB][UP_TY + 2][WAVE + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bxi = blockIdx.x, byi = blockIdx.y;
    if (out.grp > 0 && !xcd_block(blockIdx.x, out.grp, out.gx, out.gy, out.xmagic, bxi, byi)) return;
    const int cx0 = (bxi + out.bx0) * WAVE, cy0 = byi * UP_TY;
    // (a TileTab: the tiles whose columns meet the block's, widened by the one coarse column the staged halo of the first step's gathered top
    // level reaches - top_px looks tiles up on the block's coarse columns cx0 - 1 .. cx0 + WAVE; a TileSet: all tiles, as given)
    tile_range(ts, 2 * cx0 - (TOP ? 2 : 0), 2 * cx0 + 2 * WAVE + (TOP ? 2 : 0), tb, te);
    PT_DECL;
    // float work types: the accumulators and every stencil operation on (b, g) / (r, -) register pairs (packed fp32, see pyr_up_2x2_pk)
    constexpr bool PK = M != M_I16;
    WT acc[2][2][3];
    f32x2 accA[2][2], accC[2], accW[2];     // (b, g) per pixel; r and the weight sum of the two pixels of a fine row
    float accw[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j][0] = 0; acc[i][j][1] = 0; acc[i][j][2] = 0; accw[i][j] = 0.f; accA[i][j] = splat2(0.f); accC[i] = splat2(0.f); accW[i] = splat2(0.f); }
    constexpr int NCT = (UP_TY + 2) * (WAVE + 2);
    int rnd = 0;                                           // rounds executed so far
    for (int t0 = tb; t0 < max(te, tb + 1); t0 += G) {
        const bool with_out = t0 + G >= te;
        bool touch[G], mine[G];
        int lx0[G], ly0[G], ccl[G], crw[G];
        const void* cimg[G];
        Px<M> gg[G][2][2];
        RawPair rw[FINE0 ? G : 1][2];     // FINE0: the level-0 windows, decoded after the barrier
        // The order below is the order of the wave's latency chain (measured with s_memtime per phase: two thirds of a
        // wave's life used to be spent getting its loads issued): 1. the round's tile descriptors, every scalar load
        // up front in one batch (they used to be fetched one dependent s_load at a time, ten round trips per tile);
        // 2. the coarse tiles' staging, in flight under 3. the fine pixels' loads, all rows' windows issued before any is
        // decoded (level 0: decoded after the barrier, so that the round waits for memory once).
#pragma unroll
        for (int s = 0; s < G; ++s) {
            const int t = min(t0 + s, te - 1);
            touch[s] = false; mine[s] = false; lx0[s] = 0; ly0[s] = 0; ccl[s] = 1; crw[s] = 1; cimg[s] = nullptr;
            if (te > tb) {
                // block-uniform: does the block's fine region touch the tile's rectangle?
                const int tx = ts.x_tl[t], ty = ts.y_tl[t], tw = ts.w[t], th = ts.h[t], ccols = ts.coarse[t].cols, crows = ts.coarse[t].rows;
                cimg[s] = ts.coarse[t].img; ccl[s] = ccols; crw[s] = crows;
                asm volatile("" ::"s"(tx), "s"(ty), "s"(tw), "s"(th), "s"(ccols), "s"(crows), "s"(cimg[s]));   // one batch of s_loads, one wait
                touch[s] = (t0 + s < te) & !((2 * cx0 >= tx + tw) | (2 * cx0 + 2 * WAVE <= tx) | (2 * cy0 >= ty + th) | (2 * cy0 + 2 * UP_TY <= ty));
                lx0[s] = cx0 - (tx >> 1); ly0[s] = cy0 - (ty >> 1);      // block origin in the tile's coarse coordinates
                const int lcx = lx0[s] + lane, lcy = ly0[s] + wv;
                mine[s] = touch[s] & ((unsigned)lcx < (unsigned)ccols) & ((unsigned)lcy < (unsigned)crows);
            }
        }
        PT(t0 == tb ? 0 : 4);    // descriptors
        // A round none of whose tiles reaches this block is skipped altogether (block-uniform: no staging, no barrier) unless it is the
        // one that stages out_k - with two tiles side by side that is the first round of every block right of the overlap.
        {
            bool any = false;
#pragma unroll
            for (int s = 0; s < G; ++s) any = any | touch[s];
            if (!any && !with_out) continue;
        }
        const int b0 = DMA_T ? (rnd & 1) * G : 0;          // this round's tile buffers: they alternate between EXECUTED rounds
        const bool later_round = rnd > 0;
        ++rnd;
        Px<M> sv[G + 1][2];                                // staging registers of whatever is not staged by DMA
        // register-staged coarse tiles of a level step (fp16 tile levels): no branch around the loads either (see the fine pixels below) - a tile
        // that does not reach the block, and the threads past the tile's last entry, read one of its pixels that nothing will use
        constexpr bool SV_FLAT = !DMA_T && !FINE0;
        if constexpr (SV_FLAT) {
            if (te > tb) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int i = min((int)threadIdx.x + 256 * it, NCT - 1);
                    const int ry = i / (WAVE + 2), rx = i - ry * (WAVE + 2);
#pragma unroll
                    for (int s = 0; s < G; ++s) {
                        const int gx = touch[s] ? min(max(lx0[s] - 1 + rx, 0), ccl[s] - 1) : 0, gy = touch[s] ? up_row_map<M>(ly0[s] - 1 + ry, crw[s]) : 0;
                        sv[s][it] = load_px_rgb<M, false>(ts.coarse[min(t0 + s, te - 1)], gx, gy);
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = threadIdx.x + 256 * it;
            if (i < NCT) {
                const int ry = i / (WAVE + 2), rx = i - ry * (WAVE + 2);
#pragma unroll
                for (int s = 0; s < G; ++s)
                    if (!SV_FLAT && touch[s]) {
                        const int gx = min(max(lx0[s] - 1 + rx, 0), ccl[s] - 1), gy = up_row_map<M>(ly0[s] - 1 + ry, crw[s]);
                        if constexpr (DMA_T) glds16((const float4*)cimg[s] + (__umul24((unsigned)gy, (unsigned)ccl[s]) + (unsigned)gx), &ct[b0 + s][0][0] + (i - lane));
                        else sv[s][it] = load_px_rgb<M, false>(ts.coarse[t0 + s], gx, gy);
                    }
                if (with_out) {
                    const int gx = min(max(cx0 - 1 + rx, 0), coarse_out.cols - 1), gy = up_row_map<M>(cy0 - 1 + ry, coarse_out.rows);
                    if constexpr (DMA_O) glds16((const float4*)coarse_out.img + (__umul24((unsigned)gy, (unsigned)coarse_out.cols) + (unsigned)gx), &ct[NB - 1][0][0] + (i - lane));
                    else if constexpr (TOP) sv[G][it] = top_px<M>(ts, tb, te, gx, gy, 1);
                    else sv[G][it] = load_px_rgb<M, OUT_DST>(coarse_out, gx, gy);
                }
            }
        }
        PT(t0 == tb ? 1 : 5);    // coarse tiles issued
        // Wave-level tier (FINE0, CV_8UC3 tiles): when EVERY lane of the wave owns pixels of the tile and both of its rows' windows lie
        // inside the tile's buffer - true for all waves but those on a tile's rim - the wave takes a path without a single per-lane
        // branch: the per-pixel form below spends as many scalar instructions on exec-mask bookkeeping (343 per wave against 547
        // vector ones, profiles/round2_final_kernel_sq.txt) as the arithmetic is worth.
        bool wfast[G];
#pragma unroll
        for (int s = 0; s < G; ++s) {
            wfast[s] = false;
            if constexpr (FINE0 && SK == SK_U8) {
                if (touch[s]) {
                    const Src0& q = ts.s0[t0 + s];
                    const int lcx = lx0[s] + lane, lcy = ly0[s] + wv;
                    unsigned io[2], mo[2];
                    const bool f0 = wave_pair_fast(q, 2 * lcx, 2 * lcy, mine[s], io[0], mo[0]), f1 = wave_pair_fast(q, 2 * lcx, 2 * lcy + 1, mine[s], io[1], mo[1]);
                    wfast[s] = f0 & f1;
                    if (wfast[s]) { rw[s][0] = raw_pair_load(q, io[0], mo[0]); rw[s][1] = raw_pair_load(q, io[1], mo[1]); }
                }
            }
        }
        if constexpr (!FINE0) {
            // The thread's 2 x 2 fine pixels of each tile of the round, loaded with NO branch around them (round 5): a lane that owns no pixel of
            // the tile - and every lane of a block the tile does not reach - reads the tile's pixels (0..1, 0..1) instead (one cache line per load
            // for the whole wave) and never uses them.  Behind `if (mine[s])` each tile's eight loads were drained before the next tile's were issued.
            if (te > tb) {
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const int t = min(t0 + s, te - 1);
                    const int fx = mine[s] ? 2 * (lx0[s] + lane) : 0, fy = mine[s] ? 2 * (ly0[s] + wv) : 0;
                    const LevelBuf fl = ts.fine[t];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        if constexpr (M == M_F32 || M == M_I16) {      // planar level 1 or 16-byte records (block-uniform): one branch-free form
                            gg[s][dy][0] = load_px_tile<M>(fl, fx, fy + dy, ts.q8 != 0); gg[s][dy][1] = load_px_tile<M>(fl, fx + 1, fy + dy, ts.q8 != 0);
                        } else {
                            gg[s][dy][0] = load_px<M, false>(fl, fx, fy + dy); gg[s][dy][1] = load_px<M, false>(fl, fx + 1, fy + dy);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < G; ++s)
            if (FINE0 && !wfast[s] && mine[s]) {   // the thread's 2x2 level-0 pixels of this tile
                const int t = t0 + s, lcx = lx0[s] + lane, lcy = ly0[s] + wv;
                Src0 s0;
                if constexpr (FINE0) {
                    s0 = ts.s0[t];
                    asm volatile("" ::"s"(s0.top), "s"(s0.left), "s"(s0.rows), "s"(s0.cols), "s"((unsigned)s0.img_step), "s"((unsigned)s0.mask_step),
                                 "s"(s0.imis), "s"(s0.mmis), "s"(s0.iend), "s"(s0.mend), "s"(s0.img_al), "s"(s0.mask_al));   // the descriptor in one batch
                }
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    if constexpr (FINE0) rw[s][dy] = src0_pair_issue<SK>(s0, 2 * lcx, 2 * lcy + dy);
                }
            }
        if constexpr (!DMA_T || !DMA_O) {
            if constexpr (!DMA_T) { if (later_round) __syncthreads(); }     // the previous round's readers are done with the (single) tile buffers
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = threadIdx.x + 256 * it;
                if (i < NCT) {
                    const int ry = i / (WAVE + 2), rx = i - ry * (WAVE + 2);
                    if constexpr (!DMA_T) {
#pragma unroll
                        for (int s = 0; s < G; ++s) if (touch[s]) ct[s][ry][rx] = sv[s][it];
                    }
                    if constexpr (!DMA_O) { if (with_out) ct[NB - 1][ry][rx] = sv[G][it]; }   // staged once, in the last round
                }
            }
        }
        PT(t0 == tb ? 2 : 6);    // fine pixels issued
        __syncthreads();
        PT(t0 == tb ? 3 : 7);    // memory + barrier wait
#pragma unroll
        for (int s = 0; s < G; ++s) {
            const int t = t0 + s;
            if (wfast[s]) {          // every lane: both windows loaded
                if constexpr (FINE0 && SK == SK_U8) {
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) raw_pair_decode<M>(rw[s][dy], gg[s][dy][0], gg[s][dy][1]);
                }
            } else {
            if (!mine[s]) continue;
            if constexpr (FINE0) {
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
                    src0_pair_finish<M, SK>(ts.s0[t], 2 * (lx0[s] + lane), 2 * (ly0[s] + wv) + dy, rw[s][dy], gg[s][dy][0], gg[s][dy][1]);
            }
            }
            if constexpr (PK) {
                const UpPk u = pyr_up_2x2_pk<M>(ct[b0 + s], lane, wv, lx0[s] + lane, ccl[s]);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {   // acc = acc + (g - pyrUp) * w, accw = accw + w: two values per instruction
                    const Px<M> g0 = gg[s][dy][0], g1 = gg[s][dy][1];
                    const f32x2 a0 = {(float)g0.c0, (float)g0.c1}, a1 = {(float)g1.c0, (float)g1.c1}, gc = {(float)g0.c2, (float)g1.c2}, gw = {g0.w, g1.w};
                    accA[dy][0] = accA[dy][0] + (a0 - u.a[dy][0]) * splat2(g0.w);
                    accA[dy][1] = accA[dy][1] + (a1 - u.a[dy][1]) * splat2(g1.w);
                    accC[dy] = accC[dy] + (gc - u.c[dy]) * gw;
                    accW[dy] = accW[dy] + gw;
                }
                continue;
            }
            Up4<M> u = pyr_up_2x2<M>(ct[b0 + s], lane, wv, lx0[s] + lane, ccl[s]);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const Px<M> g = gg[s][dy][dx];
                    if constexpr (M == M_I16 && FINE0 && SK == SK_U8) {
                        // CV_8UC3 tiles: g in [0, 255], pyrUp of their Gaussian level in [0, 255], w in [0, 1] - cv::subtract cannot saturate,
                        // static_cast<short>(lap * w) is a plain truncation of a value within +-255, and the sum over at most DEF_MAX = 20 tiles (+-5100) cannot
                        // wrap a short: the three guards of the general form below never act and are not issued
                        acc[dy][dx][0] = acc[dy][dx][0] + (int)((float)(g.c0 - u.v[dy][dx][0]) * g.w);
                        acc[dy][dx][1] = acc[dy][dx][1] + (int)((float)(g.c1 - u.v[dy][dx][1]) * g.w);
                        acc[dy][dx][2] = acc[dy][dx][2] + (int)((float)(g.c2 - u.v[dy][dx][2]) * g.w);
                    } else if constexpr (M == M_I16) {
                        acc[dy][dx][0] = wrap_s16(acc[dy][dx][0] + f2s_x86((float)sat_s16(g.c0 - u.v[dy][dx][0]) * g.w));
                        acc[dy][dx][1] = wrap_s16(acc[dy][dx][1] + f2s_x86((float)sat_s16(g.c1 - u.v[dy][dx][1]) * g.w));
                        acc[dy][dx][2] = wrap_s16(acc[dy][dx][2] + f2s_x86((float)sat_s16(g.c2 - u.v[dy][dx][2]) * g.w));
                    } else {
                        acc[dy][dx][0] = acc[dy][dx][0] + (g.c0 - u.v[dy][dx][0]) * g.w;
                        acc[dy][dx][1] = acc[dy][dx][1] + (g.c1 - u.v[dy][dx][1]) * g.w;
                        acc[dy][dx][2] = acc[dy][dx][2] + (g.c2 - u.v[dy][dx][2]) * g.w;
                    }
                    accw[dy][dx] = accw[dy][dx] + g.w;
                }
        }
        PT(t0 == tb ? 8 : 9);    // decode + pyrUp + accumulate
    }
    const int cx = cx0 + lane, cy = cy0 + wv;
    if (cx >= coarse_out.cols || cy >= coarse_out.rows) { PT_FLUSH; return; }
    // Blends of CV_8UC3 / CV_16SC3 tiles in the last step: every Laplacian is an integer minus a multiple of 2^-30 (a pyrUp of fp32 or
    // fp16 levels of integer images), every weight a multiple of 2^-8 / 255 or zero, so a normalised numerator is zero or far above
    // 2^-103 and far below 2^31: the shared-reciprocal division (isx_device.hpp) and the clamp-first conversions are exact.
    constexpr bool BOUNDED = PK && FINE0 && (SK == SK_U8 || SK == SK_S16);
    // wave-level tier of the stores: every (remaining) lane's 2 x 2 pixels inside the result and the mats fit the vector path
    bool allin = false;
    if constexpr (FINE0) allin = out.vec && !out.img_f32 && __builtin_amdgcn_ballot_w64(!((2 * cx + 1 < out.cols) & (2 * cy + 1 < out.rows))) == 0ull;
    if constexpr (PK) {
        const UpPk u = pyr_up_2x2_pk<M>(ct[NB - 1], lane, wv, cx, coarse_out.cols);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int fy = 2 * cy + dy;
            const f32x2 den = accW[dy] + splat2(WEIGHT_EPS);                 // normalizeUsingWeightMap: c / (w + 1e-5f), both pixels of the row
            f32x2 n0, n1, nc;
            if constexpr (BOUNDED) {
                const f32x2 r0 = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                const f32x2 r1 = refine_rcp(den, r0);
                n0 = div_by_refined(accA[dy][0], splat2(den.x), splat2(r1.x));
                n1 = div_by_refined(accA[dy][1], splat2(den.y), splat2(r1.y));
                nc = div_by_refined(accC[dy], den, r1);
            } else {
                n0.x = accA[dy][0].x / den.x; n0.y = accA[dy][0].y / den.x; n1.x = accA[dy][1].x / den.y; n1.y = accA[dy][1].y / den.y;
                nc.x = accC[dy].x / den.x; nc.y = accC[dy].y / den.y;
            }
            const f32x2 ra0 = u.a[dy][0] + n0, ra1 = u.a[dy][1] + n1, rc = u.c[dy] + nc;    // restoreImageFromLaplacePyr: pyrUp(out_k) + level
            Px<M> dd[2];
            dd[0].c0 = ra0.x; dd[0].c1 = ra0.y; dd[0].c2 = rc.x; dd[0].w = accW[dy].x;
            dd[1].c0 = ra1.x; dd[1].c1 = ra1.y; dd[1].c2 = rc.y; dd[1].w = accW[dy].y;
            if constexpr (FINE0) {
                if (allin) write_final_pair<M, BOUNDED, true>(out, 2 * cx, fy, dd[0], dd[1]);
                else write_final_pair<M, BOUNDED>(out, 2 * cx, fy, dd[0], dd[1]);
            } else if (out.rec12) {
                const unsigned i12 = __umul24((unsigned)fy, (unsigned)fine_out.cols) + 2u * (unsigned)cx;
                store_rgb12<M>(fine_out, i12, dd[0]); store_rgb12<M>(fine_out, i12 + 1u, dd[1]);
            } else { store_px<M, OUT_DST>(fine_out, 2 * cx, fy, dd[0]); store_px<M, OUT_DST>(fine_out, 2 * cx + 1, fy, dd[1]); }
        }
    } else {
    Up4<M> u = pyr_up_2x2<M>(ct[NB - 1], lane, wv, cx, coarse_out.cols);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int fy = 2 * cy + dy;
        Px<M> dd[2];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            Px<M> d;
            d.c0 = acc[dy][dx][0]; d.c1 = acc[dy][dx][1]; d.c2 = acc[dy][dx][2]; d.w = accw[dy][dx];
            if constexpr (M == M_I16 && FINE0 && SK == SK_U8) {
                // normalizeUsingWeightMap on integers within +-5100 (DEF_MAX = 20 tiles) over a denominator in [1e-5, 20 + 1e-5]: the three divisions share one
                // reciprocal (isx_device.hpp: the hardware's own recurrence; the numerators are integers, the quotients below 2^30) and
                // static_cast<short> is truncation + the low 16 bits
                const float den = d.w + WEIGHT_EPS;
                const f32x2 z = splat2(den), r1 = refine_rcp(z, splat2(__builtin_amdgcn_rcpf(den)));
                const f32x2 n01 = div_by_refined(f32x2{(float)d.c0, (float)d.c1}, z, r1), n2 = div_by_refined(f32x2{(float)d.c2, 0.f}, z, r1);
                d.c0 = wrap_s16((int)n01.x); d.c1 = wrap_s16((int)n01.y); d.c2 = wrap_s16((int)n2.x);
            } else normalise<M>(d);
            if constexpr (M == M_I16) {
                d.c0 = sat_s16(u.v[dy][dx][0] + d.c0); d.c1 = sat_s16(u.v[dy][dx][1] + d.c1); d.c2 = sat_s16(u.v[dy][dx][2] + d.c2);
            } else {
                d.c0 = u.v[dy][dx][0] + d.c0; d.c1 = u.v[dy][dx][1] + d.c1; d.c2 = u.v[dy][dx][2] + d.c2;
            }
            dd[dx] = d;
        }
        if constexpr (FINE0) {
            if (allin) write_final_pair<M, false, true>(out, 2 * cx, fy, dd[0], dd[1]);
            else write_final_pair<M>(out, 2 * cx, fy, dd[0], dd[1]);
        } else if (out.rec12) {
                const unsigned i12 = __umul24((unsigned)fy, (unsigned)fine_out.cols) + 2u * (unsigned)cx;
                store_rgb12<M>(fine_out, i12, dd[0]); store_rgb12<M>(fine_out, i12 + 1u, dd[1]);
            } else { store_px<M, OUT_DST>(fine_out, 2 * cx, fy, dd[0]); store_px<M, OUT_DST>(fine_out, 2 * cx + 1, fy, dd[1]); }
    }
    }
    PT(10);                     // epilogue: normalise, pyrUp of out, convert, stores issued
    PT_FLUSH;
}


#ifndef ISX_GATHER_LEVEL_WPE
#define ISX_GATHER_LEVEL_WPE 3      // waves per SIMD the level steps of k_collapse_gather are compiled for (re-measured in round 6: tools/probes/retune_round6.sh)
#endif
// One mosaic per launch: its tiles are all of ts.
template <int M, int SK, bool FINE0, bool TOP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FINE0 ? (M == M_F16 ? 5 : 4) : ISX_GATHER_LEVEL_WPE))) void k_collapse_gather(TileSet ts, LevelBuf coarse_out, LevelBuf fine_out, OutMat out) {
    collapse_gather_body<M, SK, FINE0, TOP>(ts, 0, ts.n, coarse_out, fine_out, out);
}
// ... its tiles in a device-resident table (more than DEF_MAX of them)
template <int M, int SK, bool FINE0, bool TOP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FINE0 ? (M == M_F16 ? 5 : 4) : ISX_GATHER_LEVEL_WPE))) void k_collapse_gather_tab(TileTab ts, LevelBuf coarse_out, LevelBuf fine_out, OutMat out) {
    collapse_gather_body<M, SK, FINE0, TOP>(ts, 0, ts.n, coarse_out, fine_out, out);
}

// Several mosaics per launch (isx_blender_blend_batch: independent pairs of one rig - BASELINE configs 3 and 4 - share every launch of
// the chain, so that its launch-latency-bound small levels run at P times the waves): blockIdx.z is the mosaic, ts holds the tiles of
// all of them grouped by mosaic (first[m] .. first[m + 1]), every mosaic has its own collapsed levels and result mats.
constexpr int BATCH_MAX = 6;        // kernel arguments: TileSet (3.2 KB) + 6 x 128 bytes stay below the 4 KB limit
struct BatchOut {
    int first[BATCH_MAX + 1];
    LevelBuf coarse_out[BATCH_MAX], fine_out[BATCH_MAX];
    OutMat out[BATCH_MAX];
};
static_assert(sizeof(TileSet) + sizeof(BatchOut) <= 4096, "a batched collapse step's arguments exceed the kernel-argument limit");
template <int M, int SK, bool FINE0, bool TOP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FINE0 ? (M == M_F16 ? 5 : 4) : ISX_GATHER_LEVEL_WPE))) void k_collapse_gather_batch(TileSet ts, BatchOut bo) {
    const int m = blockIdx.z;
    collapse_gather_body<M, SK, FINE0, TOP>(ts, bo.first[m], bo.first[m + 1], bo.coarse_out[m], bo.fine_out[m], bo.out[m]);
}

#include "collapse_roll.inc"
#include "pyrdown_l0.inc"
#include "collapse_top.inc"
#include "collapse_top2.inc"

template <int M, int SK>
int launch_pyr_down0(const TileSet& ts, dim3 grid, double bytes, hipStream_t st, int planar = 0);

// ---- host side of the tile tables (TileTab / TopTab) ---------------------------------------------------------------------------------------
// DevTable: a device buffer that mirrors host-built tables.  Uploads travel in KERNEL ARGUMENTS (k_tab_write: 3.5 KB per launch) - in stream
// order behind the kernels that still read the old contents, with no pinned staging buffer whose lifetime would have to be tracked and no host
// synchronisation - and only for the chunks that differ from what the device already holds: a fixed rig re-uses the tables of its previous
// blend() as they are (same geometry, same buffers), so the steady state uploads nothing.
constexpr size_t TAB_CH = 3584;
struct TabChunk { unsigned char b[TAB_CH]; };
__global__ __launch_bounds__(256) void k_tab_write(TabChunk c, unsigned char* dst, int n) {
    const int i = (int)threadIdx.x * 16;
    if (i < n) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(c.b + i);
}
struct DevTable {
    DevBuf buf;
    std::vector<unsigned char> mirror;      // what the device holds (after everything enqueued on `st` so far)
    std::vector<unsigned> known;            // per TAB_CH chunk: the bytes from its start that the mirror vouches for
    std::vector<unsigned char> baked;       // per chunk: a hipGraph holds a write of it (sticky): a replay rewrites the device's copy at a time the
                                            // host does not see, so the mirror never vouches for that chunk again (capture A, eager B,
                                            // replay A, eager B - the second eager blend compared equal to the mirror and skipped its upload)
    hipStream_t st = nullptr;
    long long uploads = 0;                  // chunks uploaded so far (introspection: the steady state adds none)
    // room for a whole chain's slots before its first put(): growing in the middle of a chain would work (hipFree waits for the kernels that read
    // the old table) but stall the chain once per growth
    int ensure(hipStream_t s, size_t total) {
        if (total > buf.cap || s != st) {
            if (total > buf.cap) ISX_TRY(buf.reserve(std::max(total + total / 2, (size_t)64 * TAB_CH)));
            mirror.assign(buf.cap, 0); known.assign(buf.cap / TAB_CH + 1, 0u); baked.assign(buf.cap / TAB_CH + 1, 0); st = s;
        }
        return ISX_OK;
    }
    // `bytes` (a multiple of 16) at offset `off` (a multiple of TAB_CH); *dev = where they lie
    int put(hipStream_t s, size_t off, const unsigned char* src, size_t bytes, const unsigned char** dev) {
        ISX_TRY(ensure(s, off + (bytes + TAB_CH - 1) / TAB_CH * TAB_CH));
        // A stream that is being captured into a hipGraph: the graph must hold the table's writes itself (a replay may come after another blend of
        // this blender changed the table), and what a capture enqueues has not happened: nothing is skipped and nothing is vouched for afterwards.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (s != nullptr && hipStreamIsCapturing(s, &cap) != hipSuccess) { cap = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
        const bool capturing = cap == hipStreamCaptureStatusActive;
        for (size_t o = 0; o < bytes; o += TAB_CH) {
            const size_t len = std::min(TAB_CH, bytes - o), c = (off + o) / TAB_CH;
            if (capturing) baked[c] = 1;
            if (!capturing && !baked[c] && known[c] >= len && memcmp(mirror.data() + off + o, src + o, len) == 0) continue;
            TabChunk ch;
            memcpy(ch.b, src + o, len);
            memcpy(mirror.data() + off + o, src + o, len);
            known[c] = (capturing || baked[c]) ? 0u : std::max(known[c], (unsigned)len);
            hipLaunchKernelGGL(k_tab_write, dim3(1), dim3(256), 0, s, ch, (unsigned char*)buf.p + off + o, (int)len);
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail(ISX_ERR_HIP, "launch of tab_write failed: %s", hipGetErrorString(le));
            ++uploads;
        }
        *dev = (const unsigned char*)buf.p + off;
        return ISX_OK;
    }
};
// One launch's tiles: in the kernel arguments up to DEF_MAX (TileSet), as a device table beyond (TileTab).  `td` = one TileDesc per tile;
// gshift / range_cols: the index ranges per 2^gshift columns of the launch's fine level (range_cols wide); range_cols = 0: none (kernels that
// take their tile from the grid).  `slot` numbers the launches of a chain: each has its own stretch of the table.
struct TileViews { bool tab = false; TileSet ts; TileTab tt; };
inline void tileset_of(const std::vector<TileDesc>& td, TileSet* ts) {
    memset(ts, 0, sizeof(*ts));
    ts->n = (int)td.size();
    for (int t = 0; t < ts->n; ++t) {
        const TileDesc& e = td[(size_t)t];
        ts->s0[t] = e.s0; ts->fine[t] = e.fine; ts->coarse[t] = e.coarse;
        ts->x_tl[t] = e.x_tl; ts->y_tl[t] = e.y_tl; ts->w[t] = e.w; ts->h[t] = e.h; ts->bx_lo[t] = e.bx_lo; ts->bx_hi[t] = e.bx_hi;
    }
}
// index ranges [first, last) of the tiles whose columns [x, x + w) meet each stretch of 2^gshift columns; *deepest = the longest range
inline void tile_ranges(const int* xs, const int* ws, int n, int gshift, int range_cols, std::vector<int2>* rng, int* deepest) {
    const int nr = std::max(1, (range_cols + (1 << gshift) - 1) >> gshift);
    rng->assign((size_t)nr, make_int2(1 << 30, 0));
    for (int t = 0; t < n; ++t) {
        if (ws[t] <= 0 || xs[t] + ws[t] <= 0) continue;
        const int b0 = std::min(std::max(xs[t], 0) >> gshift, nr - 1), b1 = std::min((xs[t] + ws[t] - 1) >> gshift, nr - 1);
        for (int q = b0; q <= b1; ++q) { int2& r = (*rng)[(size_t)q]; r.x = std::min(r.x, t); r.y = std::max(r.y, t + 1); }
    }
    int deep = 0;
    for (auto& r : *rng) { if (r.y <= r.x) r = make_int2(0, 0); deep = std::max(deep, r.y - r.x); }
    if (deepest) *deepest = deep;
}
struct TabScratch { std::vector<unsigned char> img; std::vector<int2> rng; std::vector<int> xs, ws; };
inline size_t tab_slot_stride(int n, int range_cols_max) {
    const size_t bytes = (size_t)n * std::max(sizeof(TileDesc), sizeof(TopDesc)) + ((size_t)range_cols_max / 32 + 4) * sizeof(int2) + 64;
    return (bytes + TAB_CH - 1) / TAB_CH * TAB_CH;
}
inline int make_views(DevTable* tab, TabScratch* sc, hipStream_t st, int slot, size_t stride, const std::vector<TileDesc>& td, int gshift, int range_cols, TileViews* v) {
    const int n = (int)td.size();
    v->tab = n > DEF_MAX;
    if (!v->tab) { tileset_of(td, &v->ts); return ISX_OK; }
    const size_t dbytes = (size_t)n * sizeof(TileDesc);
    size_t rbytes = 0;
    if (range_cols > 0) {
        sc->xs.resize((size_t)n); sc->ws.resize((size_t)n);
        for (int t = 0; t < n; ++t) { sc->xs[(size_t)t] = td[(size_t)t].x_tl; sc->ws[(size_t)t] = td[(size_t)t].w; }
        tile_ranges(sc->xs.data(), sc->ws.data(), n, gshift, range_cols, &sc->rng, nullptr);
        rbytes = (sc->rng.size() * sizeof(int2) + 15) & ~(size_t)15;
    }
    ISX_CHECK_ARG(dbytes + rbytes <= stride, ISX_ERR_INTERNAL, "tile table: %zu bytes exceed the slot's %zu", dbytes + rbytes, stride);
    sc->img.assign(dbytes + rbytes, 0);
    memcpy(sc->img.data(), td.data(), dbytes);
    if (rbytes) memcpy(sc->img.data() + dbytes, sc->rng.data(), sc->rng.size() * sizeof(int2));
    const unsigned char* dev = nullptr;
    ISX_TRY(tab->put(st, (size_t)slot * stride, sc->img.data(), dbytes + rbytes, &dev));
    TileTab& tt = v->tt;
    memset(&tt, 0, sizeof(tt));
    tt.n = n; tt.gshift = gshift; tt.nrng = range_cols > 0 ? (int)sc->rng.size() : 0;
    tt.rng = range_cols > 0 ? (const int2*)(dev + dbytes) : nullptr;
    const char* base = (const char*)dev;
    tt.s0.base = base; tt.fine.base = base; tt.coarse.base = base; tt.x_tl.base = base; tt.y_tl.base = base; tt.w.base = base; tt.h.base = base;
    tt.bx_lo.base = base; tt.bx_hi.base = base;
    return ISX_OK;
}

// level 0 -> 1 of every recorded tile: CV_8UC3 and CV_16SC3 tiles through k_pyr_down0 (ISX_PD0=0: the general kernel, for A/B runs)
// planar: level 1 is written as 12-byte image records + a weight plane (ts.coarse[t].wgt set by the caller; k_pyr_down0 only)
template <int M, int SK>
int launch_pyr_down0(const TileViews& v, dim3 grid, double bytes, hipStream_t st, int planar = 0) {      // planar: 0 records, 1 planar, 2 planar in Q8 records
    if (!v.tab) return launch_pyr_down0<M, SK>(v.ts, grid, bytes, st, planar);
    static const bool fast = [] { const char* e = getenv("ISX_PD0"); return !(e && e[0] == '0'); }();
    if constexpr (SK == SK_U8 || SK == SK_S16) {
        if constexpr (M == M_F32 && SK == SK_U8) {
            if (planar == 2) { ISX_LAUNCH("pyr_down0", bytes, st, (k_pyr_down0_tab<M, SK, 2>), grid, dim3(512), 0, v.tt); return ISX_OK; }
        }
        ISX_CHECK_ARG(planar != 2, ISX_ERR_INTERNAL, "pyr_down0: Q8 records asked of a kernel that has none");
        if constexpr (M == M_F32 || M == M_I16) {
            if (planar) { ISX_LAUNCH("pyr_down0", bytes, st, (k_pyr_down0_tab<M, SK, 1>), grid, dim3(512), 0, v.tt); return ISX_OK; }
        }
        if (fast) { ISX_LAUNCH("pyr_down0", bytes, st, (k_pyr_down0_tab<M, SK>), grid, dim3(512), 0, v.tt); return ISX_OK; }
    }
    ISX_LAUNCH("pyr_down_l0", bytes, st, (k_pyr_down_multi_tab<M, SK>), grid, dim3(512), 0, v.tt, FeedPub{nullptr, 0, nullptr, 0});
    return ISX_OK;
}
template <int M, int SK>
int launch_pyr_down0(const TileSet& ts, dim3 grid, double bytes, hipStream_t st, int planar) {
    static const bool fast = [] { const char* e = getenv("ISX_PD0"); return !(e && e[0] == '0'); }();
    if constexpr (SK == SK_U8 || SK == SK_S16) {
        if constexpr (M == M_F32 && SK == SK_U8) {
            if (planar == 2) { ISX_LAUNCH("pyr_down0", bytes, st, (k_pyr_down0<M, SK, 2>), grid, dim3(512), 0, ts); return ISX_OK; }
        }
        ISX_CHECK_ARG(planar != 2, ISX_ERR_INTERNAL, "pyr_down0: Q8 records asked of a kernel that has none");
        if constexpr (M == M_F32 || M == M_I16) {
            if (planar) { ISX_LAUNCH("pyr_down0", bytes, st, (k_pyr_down0<M, SK, 1>), grid, dim3(512), 0, ts); return ISX_OK; }
        }
        if (fast) { ISX_LAUNCH("pyr_down0", bytes, st, (k_pyr_down0<M, SK>), grid, dim3(512), 0, ts); return ISX_OK; }
    }
    ISX_LAUNCH("pyr_down_l0", bytes, st, (k_pyr_down_multi<M, SK>), grid, dim3(512), 0, ts, FeedPub{nullptr, 0, nullptr, 0});
    return ISX_OK;
}

// ------------------------------------------------------------------------------------------------
// N2  FeatherBlender (W:278-281,302,313; OpenCV 3.4.2 blenders.cpp createWeightMap / feed / blend)
//   weight = min(distanceTransform(mask, DIST_L1, 3) * sharpness, 1)
//   dst.c += short(img.c * weight) (wrapping), dst_w += weight;  blend: c / (w + 1e-5), mask = w > 1e-5
// distanceTransform_3x3 with metrics {1, 2} in 16.16 fixed point is the exact city-block distance to the
// nearest zero pixel, min'ed with the chamfer distance from the INIT_DIST0 border (only reachable when no
// zero pixel is near): computed separably — nearest zero along the row (block prefix-max / suffix-min of
// zero positions), then a forward / backward "+1" sweep down the columns (lanes = adjacent columns).
// ------------------------------------------------------------------------------------------------
constexpr int DT_BIG = 1 << 28;

// Row pass, one block per row: rowd[y][x] = |x - nearest zero of row y| (DT_BIG when the row has no zero).
// Each thread owns 16 consecutive pixels (the row is staged through LDS as aligned dwords, so any mask pointer /
// step works), finds the last / first zero inside them, the block combines those with an exclusive prefix-max /
// suffix-min (wave shuffles + 4 LDS words), and the thread writes its 16 distances as four int4 stores (the row
// pitch of the map is a multiple of 4).  Rows wider than 4096 take a forward sweep over their chunks followed by
// a backward sweep that min-merges in place.
constexpr int DTR_PX = 16;
constexpr int DTR_CHUNK = 256 * DTR_PX;

__global__ __launch_bounds__(256) void k_dt_rows(const unsigned char* mask, size_t mstep, int rows, int cols, int* rowd, int pitch) {
    __shared__ unsigned s_row[DTR_CHUNK / 4 + 4];
    __shared__ int s_wl[4], s_wf[4];
    const int y = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned char* mrow = mask + (size_t)y * mstep;
    int* out = rowd + (size_t)y * pitch;
    const int nch = (cols + DTR_CHUNK - 1) / DTR_CHUNK;
    int carry_last = -DT_BIG, carry_first = DT_BIG;
    for (int pass = 0; pass < (nch > 1 ? 2 : 1); ++pass) {
        const bool fwd = pass == 0, bwd = pass == 1 || nch == 1;
        for (int ci = 0; ci < nch; ++ci) {
            const int c = pass == 0 ? ci : nch - 1 - ci;
            const int xb = c * DTR_CHUNK, n = min(DTR_CHUNK, cols - xb);
            const unsigned char* p = mrow + xb;
            const unsigned mis = (unsigned)((uintptr_t)p & 3);
            const unsigned* pa = (const unsigned*)(p - mis);   // every dword read holds at least one byte of the row
            const int ndw = (n + (int)mis + 3) >> 2;
            __syncthreads();
            for (int i = t; i < ndw; i += 256) s_row[i] = pa[i];
            __syncthreads();
            const uint4 q = ((const uint4*)s_row)[t];
            const unsigned q4 = s_row[4 * t + 4];
            const unsigned w[4] = {__builtin_amdgcn_alignbyte(q.y, q.x, mis), __builtin_amdgcn_alignbyte(q.z, q.y, mis),
                                   __builtin_amdgcn_alignbyte(q.w, q.z, mis), __builtin_amdgcn_alignbyte(q4, q.w, mis)};
            const int x0 = DTR_PX * t;
            int f[DTR_PX], g[DTR_PX];
            int last = -DT_BIG, first = DT_BIG;
#pragma unroll
            for (int i = 0; i < DTR_PX; ++i) {
                const bool z = ((w[i >> 2] >> (8 * (i & 3))) & 255u) == 0u && x0 + i < n;
                last = z ? xb + x0 + i : last;
                f[i] = last;
            }
#pragma unroll
            for (int i = DTR_PX - 1; i >= 0; --i) {
                const bool z = ((w[i >> 2] >> (8 * (i & 3))) & 255u) == 0u && x0 + i < n;
                first = z ? xb + x0 + i : first;
                g[i] = first;
            }
            int pl = -DT_BIG, nn = DT_BIG;
            if (fwd) {
                int v = last;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v = max(v, u); }
                pl = __shfl_up(v, 1);
                if (lane == 0) pl = -DT_BIG;
                if (lane == 63) s_wl[wv] = v;
            }
            if (bwd) {
                int v = first;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_down(v, o); if (lane + o < 64) v = min(v, u); }
                nn = __shfl_down(v, 1);
                if (lane == 63) nn = DT_BIG;
                if (lane == 0) s_wf[wv] = v;
            }
            __syncthreads();
            if (fwd) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < wv) pl = max(pl, s_wl[i]);
                pl = max(pl, carry_last);
                carry_last = max(max(carry_last, max(s_wl[0], s_wl[1])), max(s_wl[2], s_wl[3]));
            }
            if (bwd) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i > wv) nn = min(nn, s_wf[i]);
                nn = min(nn, carry_first);
                carry_first = min(min(carry_first, min(s_wf[0], s_wf[1])), min(s_wf[2], s_wf[3]));
            }
#pragma unroll
            for (int k = 0; k < DTR_PX / 4; ++k) {
                const int gx0 = xb + x0 + 4 * k;
                if (gx0 >= pitch || x0 + 4 * k >= n) continue;
                int d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * k + j, gx = gx0 + j;
                    int v = DT_BIG;
                    if (fwd) v = gx - max(f[i], pl);
                    if (bwd) v = min(v, min(g[i], nn) - gx);
                    d[j] = v >= DT_BIG / 2 ? DT_BIG : v;
                }
                int4* o4 = (int4*)(out + gx0);
                if (!fwd) { const int4 e = *o4; d[0] = min(d[0], e.x); d[1] = min(d[1], e.y); d[2] = min(d[2], e.z); d[3] = min(d[3], e.w); }
                *o4 = make_int4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

// Column pass.  The "+1" sweeps are min-plus scans: forward d_f[y] = y + min_{k<=y}(r[k] - k), backward
// d_b[y] = -y + min_{k>=y}(r[k] + k), result min(d_f, d_b) — so they parallelise as prefix / suffix minima over
// DT_SEG-row segments: k_dt_seg_min reduces each segment, k_dt_cols_weight combines the segments before / after its
// own, finishes the scans in registers and writes the weight map (in place: int -> float).
constexpr int DT_SEG = 32;

__global__ __launch_bounds__(64) void k_dt_seg_min(const int* rowd, int pitch, int rows, int cols, int* seg_f, int* seg_b) {
    const int x = blockIdx.x * 64 + threadIdx.x, s = blockIdx.y;
    if (x >= cols) return;
    const int y0 = s * DT_SEG;
    int mf = 2 * DT_BIG, mb = 2 * DT_BIG;
#pragma unroll 8
    for (int j = 0; j < DT_SEG; ++j) {
        const int y = y0 + j;
        if (y < rows) { const int r = rowd[(size_t)y * pitch + x]; mf = min(mf, r - y); mb = min(mb, r + y); }
    }
    seg_f[(size_t)s * cols + x] = mf;
    seg_b[(size_t)s * cols + x] = mb;
}

__global__ __launch_bounds__(64) void k_dt_cols_weight(int* rowd, int pitch, int rows, int cols, const int* seg_f, const int* seg_b, int nseg,
                                                       float sharpness) {
    const int x = blockIdx.x * 64 + threadIdx.x, s = blockIdx.y;
    if (x >= cols) return;
    const int y0 = s * DT_SEG;
    int run_f = 2 * DT_BIG, run_b = 2 * DT_BIG;
#pragma unroll 4
    for (int i = 0; i < s; ++i) run_f = min(run_f, seg_f[(size_t)i * cols + x]);
#pragma unroll 4
    for (int i = nseg - 1; i > s; --i) run_b = min(run_b, seg_b[(size_t)i * cols + x]);
    int r[DT_SEG], db[DT_SEG];
#pragma unroll
    for (int j = 0; j < DT_SEG; ++j) r[j] = y0 + j < rows ? rowd[(size_t)(y0 + j) * pitch + x] : DT_BIG;
#pragma unroll
    for (int j = DT_SEG - 1; j >= 0; --j) { run_b = min(run_b, r[j] + (y0 + j)); db[j] = run_b - (y0 + j); }
    const unsigned INIT = (unsigned)(INT_MAX >> 2);
#pragma unroll
    for (int j = 0; j < DT_SEG; ++j) {
        const int y = y0 + j;
        run_f = min(run_f, r[j] - y);
        const int d = min(run_f + y, db[j]);
        // chamfer value in 16.16: city-block distance to a zero pixel, or INIT_DIST0 + distance to the border ring
        const unsigned border = INIT + ((unsigned)(1 + min(min(x, cols - 1 - x), min(y, rows - 1 - y))) << 16);
        const unsigned t0 = d >= DT_BIG / 2 ? border : min((unsigned)d << 16, border);
        float w = ((float)t0 * (1.f / 65536.f)) * sharpness;       // multiply(weight, sharpness)
        w = w > 1.f ? 1.f : w;                                      // threshold(1.f, THRESH_TRUNC)
        if (y < rows) *(float*)(rowd + (size_t)y * pitch + x) = w;
    }
}

template <int SK>
__global__ __launch_bounds__(256) void k_feather_acc(const unsigned char* img, size_t istep, const float* weight, int wpitch, int rows, int cols,
                                                     LevelBuf dst, int dx, int dy, Cover cov) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    int c0, c1, c2;
    if constexpr (SK == SK_U8) { const unsigned char* q = img + (size_t)y * istep + (size_t)x * 3; c0 = q[0]; c1 = q[1]; c2 = q[2]; }
    else { const short* q = (const short*)(img + (size_t)y * istep) + (size_t)x * 3; c0 = q[0]; c1 = q[1]; c2 = q[2]; }
    const float w = weight[(size_t)y * wpitch + x];
    Px<M_I16> d;   // dst_ / dst_weight_map_ are not cleared by prepare(): a pixel no earlier feed covered is zero by definition
    if (covered(cov, dx + x, dy + y)) d = load_px<M_I16, true>(dst, dx + x, dy + y);
    else { d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f; }
    d.c0 = wrap_s16(d.c0 + f2s_x86((float)c0 * w));
    d.c1 = wrap_s16(d.c1 + f2s_x86((float)c1 * w));
    d.c2 = wrap_s16(d.c2 + f2s_x86((float)c2 * w));
    d.w = d.w + w;
    store_px<M_I16, true>(dst, dx + x, dy + y, d);
}

__global__ __launch_bounds__(256) void k_feather_blend(LevelBuf dst, OutMat out, Cover cov) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dst.cols || y >= dst.rows) return;
    Px<M_I16> d;
    if (covered(cov, x, y)) d = load_px<M_I16, true>(dst, x, y);
    else { d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f; }
    normalise<M_I16>(d);
    write_final<M_I16>(out, x, y, d);
}

// Deferred FeatherBlender cycle (isx_blender_set_deferred_level0): feed() only builds the tile's weight map, blend()
// gathers dst(x, y) = SUM_t short(img_t * w_t) (wrapping, feed order) and SUM_t w_t over the tiles covering the pixel,
// normalises and writes the caller's mats — dst_ / dst_weight_map_ are never materialised.  Adding the tiles that do
// not cover a pixel would add exact zeros, so the result is the eager one bit for bit.
struct FeatherSet {
    int n;
    const unsigned char* img[DEF_MAX];
    size_t istep[DEF_MAX];
    const float* wgt[DEF_MAX];
    int wpitch[DEF_MAX];
    int x[DEF_MAX], y[DEF_MAX], w[DEF_MAX], h[DEF_MAX];
};

template <int SK>
__global__ __launch_bounds__(256) void k_feather_gather(FeatherSet fs, OutMat out) {
    const int x = (blockIdx.x + out.bx0) * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= out.cols || y >= out.rows) return;
    Px<M_I16> d; d.c0 = 0; d.c1 = 0; d.c2 = 0; d.w = 0.f;
    for (int t = 0; t < fs.n; ++t) {
        const int lx = x - fs.x[t], ly = y - fs.y[t];
        if ((unsigned)lx < (unsigned)fs.w[t] && (unsigned)ly < (unsigned)fs.h[t]) {
            int c0, c1, c2;
            if constexpr (SK == SK_U8) { const unsigned char* q = fs.img[t] + (size_t)ly * fs.istep[t] + (size_t)lx * 3; c0 = q[0]; c1 = q[1]; c2 = q[2]; }
            else { const short* q = (const short*)(fs.img[t] + (size_t)ly * fs.istep[t]) + (size_t)lx * 3; c0 = q[0]; c1 = q[1]; c2 = q[2]; }
            const float w = fs.wgt[t][(size_t)ly * fs.wpitch[t] + lx];
            d.c0 = wrap_s16(d.c0 + f2s_x86((float)c0 * w));
            d.c1 = wrap_s16(d.c1 + f2s_x86((float)c1 * w));
            d.c2 = wrap_s16(d.c2 + f2s_x86((float)c2 * w));
            d.w = d.w + w;
        }
    }
    normalise<M_I16>(d);
    write_final<M_I16>(out, x, y, d);
}

// ------------------------------------------------------------------------------------------------
// Blender::NO (W:276 `Blender::createDefault(Blender::NO, false)`; OpenCV 3.4.2 blenders.cpp, the base class):
//   prepare: dst_ (CV_16SC3) and dst_mask_ (CV_8U) of dst_roi's size, both zeroed
//   feed   : where mask != 0: dst_(dy + y, dx + x) = img(y, x); everywhere: dst_mask_(dy + y, dx + x) |= mask(y, x)
//   blend  : dst_.setTo(0, dst_mask_ == 0); dst = dst_; dst_mask = dst_mask_
// dst_ is dense short3, dst_mask_ dense bytes, pitch = the ROI's width.
// ------------------------------------------------------------------------------------------------
template <int SK>
__global__ __launch_bounds__(256) void k_no_feed(const unsigned char* img, size_t istep, const unsigned char* mask, size_t mstep, int rows, int cols,
                                                 short* dst, unsigned char* dmask, int dpitch, int dx, int dy) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const unsigned m = mask[(size_t)y * mstep + x];
    const size_t o = (size_t)(dy + y) * dpitch + (dx + x);
    if (m) {
        int c0, c1, c2;
        if constexpr (SK == SK_U8) { const unsigned char* q = img + (size_t)y * istep + (size_t)x * 3; c0 = q[0]; c1 = q[1]; c2 = q[2]; }
        else { const short* q = (const short*)(img + (size_t)y * istep) + (size_t)x * 3; c0 = q[0]; c1 = q[1]; c2 = q[2]; }
        short* d = dst + o * 3;
        d[0] = (short)c0; d[1] = (short)c1; d[2] = (short)c2;
        dmask[o] |= (unsigned char)m;
    }
}
__global__ __launch_bounds__(256) void k_no_blend(const short* dst, const unsigned char* dmask, int dpitch, OutMat out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= out.cols || y >= out.rows) return;
    const size_t o = (size_t)y * dpitch + x;
    const unsigned m = dmask[o];
    const short* d = dst + o * 3;
    int c0 = d[0], c1 = d[1], c2 = d[2];
    if (m == 0) { c0 = 0; c1 = 0; c2 = 0; }      // dst_.setTo(Scalar::all(0), dst_mask_ == 0)
    if (out.mask) out.mask[(size_t)y * out.mask_step + x] = (unsigned char)m;
    if (out.img_f32 == 2) {                       // + result.convertTo(CV_8U)
        unsigned char* q = out.img + (size_t)y * out.img_step + (size_t)x * 3;
        q[0] = (unsigned char)sat_u8(c0); q[1] = (unsigned char)sat_u8(c1); q[2] = (unsigned char)sat_u8(c2);
    } else {
        short* q = (short*)(out.img + (size_t)y * out.img_step) + (size_t)x * 3;
        q[0] = (short)c0; q[1] = (short)c1; q[2] = (short)c2;
    }
}

// zero every pixel of a level that no fed tile covers (only needed when more than MAX_COVER tiles are
// fed, and by the level-introspection entry point)
template <int M>
__global__ __launch_bounds__(256) void k_fill_uncovered(LevelBuf lv, Cover cov) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= lv.cols || y >= lv.rows || covered(cov, x, y)) return;
    Px<M> z; z.c0 = 0; z.c1 = 0; z.c2 = 0; z.w = 0.f;
    store_px<M, true>(lv, x, y, z);
}

// stage_coarse reads coarse pixels through the same definition (uncovered == 0)
// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
constexpr int MAX_LEVELS = 24;

size_t g_px_bytes(int prec) { return prec == M_F16 ? 8 : 16; }           // tile Gaussian record (I16: {int b, g, r; float w}, see the header comment)
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
    int win_x0 = 0, win_x1 = 0;          // isx_blender_set_window: blend() produces columns [win_x0, win_x1) only (0, 0 = everything)
    LevelBuf dst[MAX_LEVELS];
    DevBuf dst_arena, tile_arena;
    DevBuf out_arena;                    // I16, deferred cycle: the collapsed levels out_k as 16-byte register records
    DevTable tab;                        // deferred cycle of more than DEF_MAX tiles: the launches' tile tables in device memory (TileTab)
    TabScratch tab_sc;
    std::vector<TileDesc> td;            // one launch's tiles, host side (reused)
    std::vector<TopDesc> tpd;
    MatStage st_img, st_mask, st_out, st_outmask;
    MatStage st_other;                   // isx_blender_feed_dilated: the warped mask that is AND-ed in (W:299)
    DevBuf dil_mask;                     // ... the dilated & AND-ed mask of an eager / Feather / NO feed (recorded tiles keep theirs per tile)
    std::vector<unsigned char> host_tmp;
    // level-0 rectangles (x, y, w, h in dst_roi_ coordinates) written by the feeds so far; `cleared`
    // means every level has been zero-filled outside them (only when > MAX_COVER tiles are fed)
    std::vector<int4> fed;
    bool cleared = false;
    // deferred level 0: tiles recorded by feed(), consumed by blend()
    // fused feed (k_feed_pd0, deferred mode 2): g1 = 0 level 1 not produced yet, 1 produced by feed() as 16-byte records, 2 produced PLANAR, 4 PLANAR in Q8 records, 3 produced by
    // feed() but since rewritten on a window's columns in another layout (to be produced again);
    // narrow != 0: a CV_16SC3 tile whose private copy was written as CV_8UC3 (sk = SK_U8, fed_sk = SK_S16) with escape segments in `wide`
    struct TileRec { Src0 s0; int sk; LevelBuf g[MAX_LEVELS]; int x_tl, y_tl, width, height;
                     int fed_sk = 0, g1 = 0, narrow = 0; unsigned char* wide = nullptr; size_t wide_step = 0; unsigned char* chunk = nullptr; int nbx = 0; };
    bool deferred = false;          // requested by the caller
    bool deferred_copy = false;     // ... with private copies of the fed device mats (OpenCV's contract kept)
    std::vector<std::unique_ptr<DevBuf>> tile_copy_img, tile_copy_mask;
    std::vector<std::unique_ptr<DevBuf>> tile_copy_wide, tile_chunk;     // narrowed tiles: the CV_16SC3 escape buffer and the segment map
    DevBuf feed_state;              // one device word per tile slot: "a segment of the narrowed copy escaped" (k_feed_pd0 sets, k_feed_publish clears)
    int* feed_pin = nullptr;        // pinned host words {sequence number, violation seen} that k_feed_publish writes
    int feed_seq = 0;
    int narrow_pending = 0;         // recorded tiles of this cycle whose narrowed copies have not been confirmed (0 .. tiles.size())
    bool narrow_published = false;  // k_feed_publish of this cycle has been enqueued
    bool narrow_off = false;        // a cycle was violated once: this blender keeps CV_16SC3 copies from now on
    bool fused_cycle = false;       // some tile of this cycle came through k_feed_pd0 (level 1 produced by feed())
    bool level0_pending = false;    // recorded tiles have not been accumulated into dst[0] yet
    std::vector<TileRec> tiles;
    std::vector<std::unique_ptr<DevBuf>> tile_arenas;      // one per recorded tile (their pyramids must outlive feed)
    std::vector<std::unique_ptr<MatStage>> tile_img, tile_mask;   // staging of host mats, one per recorded tile
    // overlap mode (isx_blender_set_overlap): a recorded tile's Gaussian chain starts right away on its own
    // side stream (it only needs the tile), so the memory-bound chain of tile t runs under the VALU-bound
    // warp of tile t+1 that the caller enqueues next on the main stream; blend() joins the side streams
    float sharpness = 0.02f;        // FeatherBlender(float sharpness = 0.02f)
    DevBuf feather_w;               // weight map of the tile being fed (int row distances, then float weights)
    struct FeatherRec { const unsigned char* img; size_t istep; int sk; float* wgt; int wpitch, dx, dy, rows, cols; };
    std::vector<FeatherRec> ftiles; // deferred FeatherBlender cycle: tiles recorded by feed(), gathered by blend()
    hipEvent_t mark_event = nullptr;   // isx_blender_set_mark_event: recorded inside a deferred blend()
    int mark_level = 0;
    bool overlap = false;
    // which code path the last blend() took (isx_blender_last_path): the fast kernels have limits, and a caller / a bench line should be
    // able to say which side of them it ran on.  cycle: 0 eager, 1 deferred, 2 deferred as part of a batched chain; last: the kernel of
    // the last collapse step - 0 none (a 0-band blend, Feather, NO), 1 k_collapse, 2 k_collapse_gather, 3 k_collapse_roll
    int path_cycle = 0, path_last = 0, path_g1 = 0;      // path_g1: layout of the tiles' level 1 in the last blend (isx_blender_level1_format)
    int path_fused = 0, path_narrow = 0;   // isx_blender_feed_path: tiles of the last blend() that came through k_feed_pd0; 0 none narrowed, 1 narrowed copies confirmed, 2 widened
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> ev_ready, ev_done;
    std::vector<char> chain_on_side;   // per recorded tile: its chain was launched by feed()
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
        if (prec == M_I16 && is_dst) {
            lv[i].wgt = base ? (float*)(base + off) : nullptr;
            off += (n * 4 + 255) & ~(size_t)255;
        }
        rows = (rows + 1) / 2; cols = (cols + 1) / 2;
    }
    *total = off;
    return ISX_OK;
}

Cover make_cover(const isx_blender* b, int level) {
    Cover c;
    memset(&c, 0, sizeof(c));
    if (b->cleared) { c.n = -1; return c; }
    c.n = (int)std::min<size_t>(b->fed.size(), MAX_COVER);
    for (int i = 0; i < c.n; ++i) {
        c.x[i] = b->fed[i].x >> level; c.y[i] = b->fed[i].y >> level;
        c.w[i] = b->fed[i].z >> level; c.h[i] = b->fed[i].w >> level;
    }
    return c;
}

int fill_uncovered(isx_blender* b, int level) {
    const LevelBuf& d = b->dst[level];
    dim3 grid(cdiv(d.cols, 64), cdiv(d.rows, 4));
    Cover c = make_cover(b, level);
    switch (b->prec) {
        case M_I16: ISX_LAUNCH("fill_uncovered", 0.0, b->stream, (k_fill_uncovered<M_I16>), grid, dim3(256), 0, d, c); break;
        case M_F32: ISX_LAUNCH("fill_uncovered", 0.0, b->stream, (k_fill_uncovered<M_F32>), grid, dim3(256), 0, d, c); break;
        default: ISX_LAUNCH("fill_uncovered", 0.0, b->stream, (k_fill_uncovered<M_F16>), grid, dim3(256), 0, d, c); break;
    }
    return ISX_OK;
}

// Level-0 pixels of the rectangle (x, y, w, h) that an earlier feed of this cycle has already written (the union of its intersections
// with the Cover rectangles): only those are READ by the accumulate kernels - a pixel no earlier feed covered is stored, not
// read-modified-written - so only those count as read traffic in the kernels' algorithmic bytes.
double covered_px(const isx_blender* b, int x, int y, int w, int h) {
    if (b->cleared) return (double)w * h;
    std::vector<int> xs{x, x + w}, ys{y, y + h};
    std::vector<int4> r;
    for (const int4& f : b->fed) {
        const int x0 = std::max(x, f.x), y0 = std::max(y, f.y), x1 = std::min(x + w, f.x + f.z), y1 = std::min(y + h, f.y + f.w);
        if (x0 < x1 && y0 < y1) { r.push_back(make_int4(x0, y0, x1, y1)); xs.push_back(x0); xs.push_back(x1); ys.push_back(y0); ys.push_back(y1); }
    }
    std::sort(xs.begin(), xs.end()); std::sort(ys.begin(), ys.end());
    double area = 0.0;
    for (size_t i = 0; i + 1 < xs.size(); ++i)
        for (size_t j = 0; j + 1 < ys.size(); ++j) {
            if (xs[i] == xs[i + 1] || ys[j] == ys[j + 1]) continue;
            for (const int4& q : r)
                if (xs[i] >= q.x && xs[i + 1] <= q.z && ys[j] >= q.y && ys[j + 1] <= q.w) { area += (double)(xs[i + 1] - xs[i]) * (ys[j + 1] - ys[j]); break; }
        }
    return area;
}

// k_feed_pd0 / k_feed_strip for (precision, tile type, level-1 layout, copy format)
template <int M, int SK, int PLD, int CF>
int launch_feed_pd0_v(const Src0& s0, const LevelBuf& g1, FeedCopy fc, dim3 grid, double bytes, hipStream_t st) {
    // single-wave strips for the whole level (k_feed_strip) unless the tile is too small for its window scheme; ISX_FEED_STRIP=0: the block kernel
    // (A/B runs), 4: four output rows per strip instead of two
    static const int strip_no = [] { const char* e = getenv("ISX_FEED_STRIP"); return e ? atoi(e) : 2; }();
    if (strip_no && s0.iend != 0u && s0.cols >= 2 && s0.rows >= 2 && s0.width >= 4 && s0.height >= 4) {
        const int no = strip_no == 4 ? 4 : 2;
        const unsigned n = (unsigned)cdiv(g1.cols, PD_OW) * (unsigned)cdiv(g1.rows, no);
        const unsigned nblk = (n + 7u) / 8u * 8u;         // (the XCD dealing walks whole groups of 8)
        if (no == 4) ISX_LAUNCH("feed_strip", bytes, st, (k_feed_strip<M, SK, PLD, CF, 4>), dim3(nblk), dim3(64), 0, s0, g1, fc);
        else ISX_LAUNCH("feed_strip", bytes, st, (k_feed_strip<M, SK, PLD, CF, 2>), dim3(nblk), dim3(64), 0, s0, g1, fc);
        return ISX_OK;
    }
    ISX_LAUNCH("feed_pd0", bytes, st, (k_feed_pd0<M, SK, PLD, CF>), grid, dim3(512), 0, s0, g1, fc);
    return ISX_OK;
}
template <int M, int SK>
int launch_feed_pd0_t(int planar, bool narrow, const Src0& s0, const LevelBuf& g1, const FeedCopy& fc, dim3 grid, double bytes, hipStream_t st) {
    // planar = 2: level 1 in Q8 records (load_px_planar) - fp32 pyramids over CV_8UC3 tiles and over CV_16SC3 tiles that are being narrowed (if those
    // turn out not to be bytes the cycle is widened and its level 1 produced again, run_blend_deferred_t)
    ISX_CHECK_ARG(planar != 2 || (M == M_F32 && (SK == SK_U8 || narrow)), ISX_ERR_INTERNAL, "feed: Q8 records asked of a kernel that has none");
    if constexpr (SK == SK_S16) {
        if (narrow) {
            if constexpr (M == M_F32) {
                if (planar == 2) return launch_feed_pd0_v<M, SK, 2, CF_NARROW>(s0, g1, fc, grid, bytes, st);
            }
            if constexpr (M == M_F32 || M == M_I16) {
                if (planar) return launch_feed_pd0_v<M, SK, 1, CF_NARROW>(s0, g1, fc, grid, bytes, st);
            }
            return launch_feed_pd0_v<M, SK, 0, CF_NARROW>(s0, g1, fc, grid, bytes, st);
        }
    }
    if constexpr (M == M_F32 && SK == SK_U8) {
        if (planar == 2) return launch_feed_pd0_v<M, SK, 2, CF_SAME>(s0, g1, fc, grid, bytes, st);
    }
    if constexpr (M == M_F32 || M == M_I16) {
        if (planar) return launch_feed_pd0_v<M, SK, 1, CF_SAME>(s0, g1, fc, grid, bytes, st);
    }
    return launch_feed_pd0_v<M, SK, 0, CF_SAME>(s0, g1, fc, grid, bytes, st);
}
int launch_feed_pd0(int prec, int sk, int planar, bool narrow, const Src0& s0, const LevelBuf& g1, const FeedCopy& fc, dim3 grid, double bytes, hipStream_t st) {
    switch (prec) {
        case M_I16: return sk == SK_U8 ? launch_feed_pd0_t<M_I16, SK_U8>(planar, narrow, s0, g1, fc, grid, bytes, st) : launch_feed_pd0_t<M_I16, SK_S16>(planar, narrow, s0, g1, fc, grid, bytes, st);
        case M_F32: return sk == SK_U8 ? launch_feed_pd0_t<M_F32, SK_U8>(planar, narrow, s0, g1, fc, grid, bytes, st) : launch_feed_pd0_t<M_F32, SK_S16>(planar, narrow, s0, g1, fc, grid, bytes, st);
        default: return sk == SK_U8 ? launch_feed_pd0_t<M_F16, SK_U8>(planar, narrow, s0, g1, fc, grid, bytes, st) : launch_feed_pd0_t<M_F16, SK_S16>(planar, narrow, s0, g1, fc, grid, bytes, st);
    }
}

// Narrowed tiles (k_feed_pd0<.., CF_NARROW>): the violation words of the cycle's tiles go to the host in one small launch ...
int narrow_publish(isx_blender* b) {
    if (b->narrow_pending == 0 || b->narrow_published) return ISX_OK;
    const int seq = ++b->feed_seq;
    ISX_LAUNCH("feed_publish", 0.0, b->stream, k_feed_publish, dim3(1), dim3(64), 0, (unsigned*)b->feed_state.p, (int)b->tiles.size(), b->feed_pin, seq);
    b->narrow_published = true;
    return ISX_OK;
}
// ... and are read here (the caller has enqueued whatever does not depend on the answer).  No violation (always, after W:294): the tiles are
// CV_8UC3 tiles from here on.  Violation: every narrowed segment is widened into the escape buffer (k_feed_widen), which then is the tile's
// CV_16SC3 private copy, the tiles become CV_16SC3 tiles again (*widened = true) and this blender stops narrowing.
int narrow_resolve(isx_blender* b, bool* widened) {
    *widened = false;
    if (b->narrow_pending == 0) return ISX_OK;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        ISX_CHECK_ARG(hipStreamIsCapturing(b->stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone, ISX_ERR_STATE,
                      "blend: tiles fed before the capture began cannot be blended inside it (their narrowed copies are confirmed on the host)");
    }
    ISX_TRY(narrow_publish(b));
    const int seq = b->feed_seq;
    // (a spin, not hipStreamSynchronize: the stream already holds the launches that follow the publish, and waiting for those would leave the GPU
    // idle while the last step is enqueued; when the GPU is behind the host this is where the host waits for it - one step ahead at most)
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 1; __atomic_load_n(&b->feed_pin[0], __ATOMIC_ACQUIRE) != seq; ++spin) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spin & 4095) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            ISX_HIP(hipStreamSynchronize(b->stream));
            ISX_CHECK_ARG(__atomic_load_n(&b->feed_pin[0], __ATOMIC_ACQUIRE) == seq, ISX_ERR_HIP, "feed: the narrowed copies' check finished without publishing its result");
        }
    }
    const bool bad = __atomic_load_n(&b->feed_pin[1], __ATOMIC_RELAXED) != 0;
    b->narrow_pending = 0;
    b->path_narrow = bad ? 2 : 1;
    if (!bad) { for (auto& r : b->tiles) r.narrow = 0; return ISX_OK; }
    b->narrow_off = true;
    for (auto& r : b->tiles) {
        if (!r.narrow) continue;
        ISX_LAUNCH("feed_widen", (double)r.s0.rows * r.s0.cols * 9.0, b->stream, k_feed_widen, dim3(cdiv(r.s0.cols, 256), r.s0.rows), dim3(256), 0,
                   r.s0.img, (unsigned)r.s0.img_step, r.wide, (unsigned)r.wide_step, (const unsigned char*)r.chunk, r.nbx, r.s0.rows, r.s0.cols, r.s0.left);
        r.s0.img = r.wide; r.s0.img_step = r.wide_step;
        r.s0.imis = (unsigned)((uintptr_t)r.s0.img & 3); r.s0.img_al = r.s0.img - r.s0.imis;
        r.s0.iend = 0;
        if ((unsigned long long)r.wide_step * r.s0.rows < (1ull << 31) && r.wide_step < (1u << 24) && r.s0.mend != 0u)
            r.s0.iend = (unsigned)((size_t)(r.s0.rows - 1) * r.wide_step + (size_t)r.s0.cols * 6) + r.s0.imis;
        r.sk = SK_S16; r.narrow = 0;
    }
    *widened = true;
    return ISX_OK;
}

int src_kind_of(int type) { return type == ISX_8UC3 ? SK_U8 : (type == ISX_16SC3 ? SK_S16 : SK_F32); }
double src_px_bytes(int sk) { return sk == SK_U8 ? 3.0 : (sk == SK_S16 ? 6.0 : 12.0); }

template <int M, int SK>
int run_feed(isx_blender* b, const Src0& s0, LevelBuf* g, int L, int x_tl, int y_tl, bool skip_level0) {
    hipStream_t st = b->stream;
    const int prec = M;
    // Gaussian chain (image + weight): G_{k+1} = pyrDown(G_k)
    for (int k = 0; k < L; ++k) {
        dim3 grid(cdiv(g[k + 1].cols, PD_OW), cdiv(g[k + 1].rows, PD_TY));
        double in_px = (double)g[k].rows * g[k].cols, out_px = (double)g[k + 1].rows * g[k + 1].cols;
        double bytes = in_px * (k == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec)) + out_px * alg_g(prec);
        if (k == 0) ISX_LAUNCH("pyr_down_l0", bytes, st, (k_pyr_down<M, SK>), grid, dim3(512), 0, s0, g[0], g[1]);
        else ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down<M, SK_LEVEL>), grid, dim3(512), 0, s0, g[k], g[k + 1]);
    }
    // Laplacian + weighted accumulate: every level in one launch
    if (L <= ACC_MAXL) {
        AccArgs a;
        memset(&a, 0, sizeof(a));
        a.s0 = s0; a.L = L; a.x_tl = x_tl; a.y_tl = y_tl; a.cov0 = make_cover(b, 0);
        double bytes = 0.0;
        int nb = 0;
        // destination records: every one is written, only those an earlier feed covered are read first
        const double rd_frac = covered_px(b, x_tl, y_tl, g[0].cols, g[0].rows) / ((double)g[0].rows * g[0].cols);
        for (int k = 0; k <= L; ++k) {
            a.g[k] = g[k]; a.dst[k] = b->dst[k];
            a.blk_start[k] = nb;
            double px = (double)g[k].rows * g[k].cols;
            double gin = (k == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec));
            if (k == 0 && skip_level0) {
                a.gw[k] = 1;   // deferred level 0: no blocks, blend() accumulates it in registers
            } else if (k < L) {
                a.gw[k] = cdiv(g[k + 1].cols, WAVE);
                nb += a.gw[k] * cdiv(g[k + 1].rows, UP_TY);
                bytes += px * (gin + (1.0 + rd_frac) * alg_d(prec)) + (double)g[k + 1].rows * g[k + 1].cols * alg_g_rgb(prec);
            } else {
                a.gw[k] = cdiv(g[k].cols, 64);
                nb += a.gw[k] * cdiv(g[k].rows, 4);
                bytes += px * (gin + (1.0 + rd_frac) * alg_d(prec));
            }
        }
        a.blk_start[L + 1] = nb;
        ISX_LAUNCH("lap_acc_all", bytes, st, (k_lap_acc_all<M, SK>), dim3(nb), dim3(256), 0, a);
        return ISX_OK;
    }
    int xt = x_tl, yt = y_tl;
    const double rd_frac2 = covered_px(b, x_tl, y_tl, g[0].cols, g[0].rows) / ((double)g[0].rows * g[0].cols);
    for (int k = 0; k < L; ++k) {
        dim3 grid(cdiv(g[k + 1].cols, WAVE), cdiv(g[k + 1].rows, UP_TY));
        double fine_px = (double)g[k].rows * g[k].cols, coarse_px = (double)g[k + 1].rows * g[k + 1].cols;
        double bytes = fine_px * ((k == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec)) + (1.0 + rd_frac2) * alg_d(prec)) + coarse_px * alg_g_rgb(prec);
        Cover cov = make_cover(b, k);
        if (k == 0) ISX_LAUNCH("lap_acc_l0", bytes, st, (k_lap_acc<M, SK>), grid, dim3(256), 0, s0, g[0], g[1], b->dst[0], xt, yt, cov);
        else ISX_LAUNCH("lap_acc", bytes, st, (k_lap_acc<M, SK_LEVEL>), grid, dim3(256), 0, s0, g[k], g[k + 1], b->dst[k], xt, yt, cov);
        xt /= 2; yt /= 2;
    }
    {
        dim3 grid(cdiv(g[L].cols, 64), cdiv(g[L].rows, 4));
        double px = (double)g[L].rows * g[L].cols;
        double bytes = px * ((L == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec)) + (1.0 + rd_frac2) * alg_d(prec));
        Cover cov = make_cover(b, L);
        if (L == 0) ISX_LAUNCH("top_acc", bytes, st, (k_top_acc<M, SK>), grid, dim3(256), 0, s0, g[0], b->dst[0], xt, yt, g[0].rows, g[0].cols, cov);
        else ISX_LAUNCH("top_acc", bytes, st, (k_top_acc<M, SK_LEVEL>), grid, dim3(256), 0, s0, g[L], b->dst[L], xt, yt, g[L].rows, g[L].cols, cov);
    }
    return ISX_OK;
}

template <int M>
int run_feed_kind(isx_blender* b, int sk, const Src0& s0, LevelBuf* g, int L, int x_tl, int y_tl, bool skip_level0) {
    switch (sk) {
        case SK_U8: return run_feed<M, SK_U8>(b, s0, g, L, x_tl, y_tl, skip_level0);
        case SK_S16: return run_feed<M, SK_S16>(b, s0, g, L, x_tl, y_tl, skip_level0);
        default: return run_feed<M, SK_F32>(b, s0, g, L, x_tl, y_tl, skip_level0);
    }
}

Cover make_cover_n(const isx_blender* b, int level, int n) {
    Cover c = make_cover(b, level);
    if (c.n > n) c.n = n;
    return c;
}

template <int M, int SK>
int launch_down_chain_t(isx_blender* b, const isx_blender::TileRec& r, hipStream_t st) {
    const int L = b->num_bands, prec = M;
    for (int k = 0; k < L; ++k) {
        dim3 grid(cdiv(r.g[k + 1].cols, PD_OW), cdiv(r.g[k + 1].rows, PD_TY));
        double bytes = (double)r.g[k].rows * r.g[k].cols * (k == 0 ? src_px_bytes(SK) + 1.0 : alg_g(prec)) + (double)r.g[k + 1].rows * r.g[k + 1].cols * alg_g(prec);
        if (k == 0) ISX_LAUNCH("pyr_down_l0", bytes, st, (k_pyr_down<M, SK>), grid, dim3(512), 0, r.s0, r.g[0], r.g[1]);
        else ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down<M, SK_LEVEL>), grid, dim3(512), 0, r.s0, r.g[k], r.g[k + 1]);
    }
    return ISX_OK;
}
template <int M>
int launch_down_chain_k(isx_blender* b, const isx_blender::TileRec& r, hipStream_t st) {
    switch (r.sk) {
        case SK_U8: return launch_down_chain_t<M, SK_U8>(b, r, st);
        case SK_S16: return launch_down_chain_t<M, SK_S16>(b, r, st);
        default: return launch_down_chain_t<M, SK_F32>(b, r, st);
    }
}
int launch_down_chain(isx_blender* b, const isx_blender::TileRec& r, hipStream_t st) {
    switch (b->prec) {
        case M_I16: return launch_down_chain_k<M_I16>(b, r, st);
        case M_F32: return launch_down_chain_k<M_F32>(b, r, st);
        default: return launch_down_chain_k<M_F16>(b, r, st);
    }
}
// make the main stream wait for the recorded tiles' side-stream chains
int join_side_streams(isx_blender* b) {
    for (size_t t = 0; t < b->tiles.size(); ++t)
        if (t < b->chain_on_side.size() && b->chain_on_side[t]) ISX_HIP(hipStreamWaitEvent(b->stream, b->ev_done[t], 0));
    return ISX_OK;
}

// Deferred tiles have had no accumulation run for them.  When the cycle cannot stay deferred (level
// introspection, a 9th tile, a tile of another type) they are replayed through the eager feed, in order.
int flush_deferred(isx_blender* b) {
    if (!b->level0_pending) return ISX_OK;
    const int L = b->num_bands;
    { bool widened; ISX_TRY(narrow_resolve(b, &widened)); }   // the replay reads the private copies: in which type?
    ISX_TRY(join_side_streams(b));   // the replay rewrites the tiles' pyramid levels
    for (size_t t = 0; t < b->tiles.size(); ++t) {
        isx_blender::TileRec& r = b->tiles[t];
        int rc;
        if (!b->cleared && b->fed.size() >= (size_t)MAX_COVER) {
            for (int k = 0; k <= L; ++k) ISX_TRY(fill_uncovered(b, k));
            b->cleared = true;
        }
        switch (b->prec) {
            case M_I16: rc = run_feed_kind<M_I16>(b, r.sk, r.s0, r.g, L, r.x_tl, r.y_tl, false); break;
            case M_F32: rc = run_feed_kind<M_F32>(b, r.sk, r.s0, r.g, L, r.x_tl, r.y_tl, false); break;
            default: rc = run_feed_kind<M_F16>(b, r.sk, r.s0, r.g, L, r.x_tl, r.y_tl, false); break;
        }
        ISX_TRY(rc);
        if (!b->cleared) b->fed.push_back(make_int4(r.x_tl, r.y_tl, r.width, r.height));
    }
    b->level0_pending = false;   // the tiles stay recorded (their buffers are in use); the cycle continues eagerly
    return ISX_OK;
}

// The last collapse step as the rolling kernel (collapse_roll.inc) when its limits hold: coarse columns [cx_lo, cx_hi) of level 1 in
// strips of RL_CW columns x R rows, one wave each, dealt to the XCDs in groups of two strip rows (xcd_block).  *done = false: the
// caller launches k_collapse_gather instead.  ISX_ROLL=0 forces that (A/B runs); ISX_ROLL_R picks the strip height.
// most tiles that reach any one strip of RL_CW x R coarse pixels: every tile covers a rectangle of strip indices; the deepest overlap of
// those rectangles is found on the grid of their corners (at most 2 n x 2 n cells)
inline int roll_max_tiles(const TileSet& ts, const LevelBuf& coarse, int cx_lo, int cx_hi, int R) {
    const int nsx = cdiv(cx_hi - cx_lo, RL_CW), nsy = cdiv(coarse.rows, R);
    int x0[DEF_MAX], x1[DEF_MAX], y0[DEF_MAX], y1[DEF_MAX], m = 0;
    for (int t = 0; t < ts.n; ++t) {
        // strip sx covers fine columns [2 (cx_lo + sx RL_CW), + 2 RL_CW); it is reached when that interval meets [x_tl, x_tl + w)
        const int fx0 = ts.x_tl[t] - 2 * cx_lo, fx1 = fx0 + ts.w[t], fy0 = ts.y_tl[t], fy1 = fy0 + ts.h[t];
        if (fx1 <= 0 || fy1 <= 0) continue;
        const int a = fx0 > 0 ? fx0 / (2 * RL_CW) : 0, b = std::min(nsx - 1, (fx1 - 1) / (2 * RL_CW));
        const int c = fy0 > 0 ? fy0 / (2 * R) : 0, d = std::min(nsy - 1, (fy1 - 1) / (2 * R));
        if (a > b || c > d) continue;
        x0[m] = a; x1[m] = b; y0[m] = c; y1[m] = d; ++m;
    }
    int most = 0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {         // deepest point lies at (a left edge, a top edge)
            const int px = x0[i], py = y0[j];
            int cnt = 0;
            for (int t = 0; t < m; ++t) cnt += (x0[t] <= px) & (px <= x1[t]) & (y0[t] <= py) & (py <= y1[t]);
            most = std::max(most, cnt);
        }
    return most;
}
// the same for any number of tiles (TileDesc records at the fine level): per left edge of a tile's strip rectangle, the tiles over that strip
// column, and among them the deepest overlap of their strip-row intervals; stops as soon as more than `cap` tiles meet (the callers ask "at
// most 2?", "at most 3?")
inline int roll_max_tiles(const std::vector<TileDesc>& td, const LevelBuf& coarse, int cx_lo, int cx_hi, int R, int cap) {
    const int nsx = cdiv(cx_hi - cx_lo, RL_CW), nsy = cdiv(coarse.rows, R), n = (int)td.size();
    std::vector<int4> r;
    r.reserve((size_t)n);
    for (int t = 0; t < n; ++t) {
        const TileDesc& e = td[(size_t)t];
        const int fx0 = e.x_tl - 2 * cx_lo, fx1 = fx0 + e.w, fy0 = e.y_tl, fy1 = fy0 + e.h;
        if (fx1 <= 0 || fy1 <= 0) continue;
        const int a = fx0 > 0 ? fx0 / (2 * RL_CW) : 0, b = std::min(nsx - 1, (fx1 - 1) / (2 * RL_CW));
        const int c = fy0 > 0 ? fy0 / (2 * R) : 0, d = std::min(nsy - 1, (fy1 - 1) / (2 * R));
        if (a > b || c > d) continue;
        r.push_back(make_int4(a, b, c, d));
    }
    std::sort(r.begin(), r.end(), [](const int4& p, const int4& q) { return p.x < q.x; });
    int most = 0;
    std::vector<int2> over;
    for (size_t i = 0; i < r.size(); ++i) {
        if (i > 0 && r[i].x == r[i - 1].x) continue;
        const int px = r[i].x;
        over.clear();
        for (size_t j = 0; j < r.size() && r[j].x <= px; ++j) if (r[j].y >= px) over.push_back(make_int2(r[j].z, r[j].w));
        for (size_t j = 0; j < over.size(); ++j) {       // deepest point of the row intervals lies at a top edge
            int cnt = 0;
            for (size_t q = 0; q < over.size(); ++q) cnt += (over[q].x <= over[j].x && over[j].x <= over[q].y) ? 1 : 0;
            most = std::max(most, cnt);
            if (most > cap) return most;
        }
    }
    return most;
}
template <int M, int SK, int R, int MAXT>
int launch_collapse_roll_r(hipStream_t st, const TileViews& v, const LevelBuf& coarse, OutMat o, int cx_lo, int cx_hi, double bytes, bool* done) {
    const int nsx = cdiv(cx_hi - cx_lo, RL_CW), nsy = cdiv(coarse.rows, R), nby = cdiv(nsy, ROLL_WAVES);
    if (nsx <= 0 || nsy <= 0) return ISX_OK;
    static const int band_mode = [] { const char* e = getenv("ISX_ROLL_BAND"); return e ? atoi(e) : 1; }();
    static const int grp_sel = [] { const char* e = getenv("ISX_ROLL_GRP"); return e ? atoi(e) : 2; }();
    const int grp = std::max(grp_sel, nsx == 1 ? 2 : 1);      // (xcd_magic needs grp * nsx >= 2)
    o.bx0 = 0; o.grp = grp; o.gx = nsx; o.gy = nby; o.xmagic = xcd_magic(grp, nsx);
    o.band = band_mode ? cdiv(nby, 8) : 0;
    const unsigned nblk = o.band ? xcd_band_blocks(grp, nsx, nby) : xcd_grid_blocks(grp, nsx, nby);
    if (v.tab) {      // (the table form is instantiated for two-row strips only)
        if constexpr (R == 2) ISX_LAUNCH("collapse_roll", bytes, st, (k_collapse_roll_tab<M, SK, R, MAXT>), dim3(nblk), dim3(64 * ROLL_WAVES), 0, v.tt, coarse, o, cx_lo);
        else return ISX_OK;
    } else ISX_LAUNCH("collapse_roll", bytes, st, (k_collapse_roll<M, SK, R, MAXT>), dim3(nblk), dim3(64 * ROLL_WAVES), 0, v.ts, coarse, o, cx_lo);
    *done = true;
    return ISX_OK;
}
// which instantiation runs the last step: 0 = none (k_collapse_gather), else 10 R + MAXT
template <int SK>
int roll_variant(const std::vector<TileDesc>& td, const LevelBuf& coarse, int cx_lo, int cx_hi) {
    static const int mode = [] { const char* e = getenv("ISX_ROLL"); return e ? atoi(e) : 1; }();
    static const int rsel = [] { const char* e = getenv("ISX_ROLL_R"); return e ? atoi(e) : 2; }();      // rows per wave (tuning runs only)
    if constexpr (SK == SK_U8 || SK == SK_S16) {
        if (!mode || coarse.cols < 2 || (unsigned long long)coarse.rows * coarse.cols * 16ull >= (1ull << 32)) return 0;
        for (const TileDesc& e : td)
            if (e.coarse.cols < 2 || e.s0.cols < 2 || e.s0.rows < 2 || e.s0.iend == 0u ||      // iend != 0: a CV_8UC3 / CV_16SC3 tile below 2 GiB, 32-bit offsets
                (unsigned long long)e.coarse.rows * e.coarse.cols * 16ull >= (1ull << 32)) return 0;
        const bool big = td.size() > (size_t)DEF_MAX;      // a device table: two-row strips or k_collapse_gather
        if (cdiv(cx_hi - cx_lo, RL_CW) <= 0 || coarse.rows <= 0) return 0;
        // two rows per wave while at most two tiles reach a strip (a pair, a row of tiles with narrow overlaps); a third slot
        // for panoramas whose tiles overlap their second neighbours (BASELINE config 5); k_collapse_gather beyond that
        // ... and two rows with a third slot (round 4: 4 waves per SIMD instead of 5, still faster than one-row strips - config 5's last step
        // 0.704 -> 0.654 ms; ISX_ROLL_R23=0: the one-row form, for A/B runs)
        static const bool r23 = [] { const char* e = getenv("ISX_ROLL_R23"); return !(e && e[0] == '0'); }();
        const int most2 = (rsel != 1 || big) ? roll_max_tiles(td, coarse, cx_lo, cx_hi, 2, 3) : 99;
        if (most2 <= 2) return 22;
        if ((r23 || big) && most2 <= 3) return 23;
        if (big) return 0;
        const int most = roll_max_tiles(td, coarse, cx_lo, cx_hi, 1, 3);
        if (most <= 2) return 12;
        if (most <= 3) return 13;
    }
    return 0;
}
template <int M, int SK>
int launch_collapse_roll(isx_blender* b, hipStream_t st, const TileViews& v, const LevelBuf& coarse, const OutMat& o, int cx_lo, int cx_hi, double bytes, int variant, bool* done) {
    *done = false;
    if constexpr (SK == SK_U8 || SK == SK_S16) {
        switch (variant) {
            case 22: return launch_collapse_roll_r<M, SK, 2, 2>(st, v, coarse, o, cx_lo, cx_hi, bytes, done);
            case 23: return launch_collapse_roll_r<M, SK, 2, 3>(st, v, coarse, o, cx_lo, cx_hi, bytes, done);
            case 12: return launch_collapse_roll_r<M, SK, 1, 2>(st, v, coarse, o, cx_lo, cx_hi, bytes, done);
            case 13: return launch_collapse_roll_r<M, SK, 1, 3>(st, v, coarse, o, cx_lo, cx_hi, bytes, done);
            default: break;
        }
    }
    return ISX_OK;
}

// timing-only ablations of the small-level tail (tools/probes/tail_ablation.sh; wrong pixels): 1 = the pyrDown launches of levels >= 2 as one block
// per tile, 2 = k_collapse_top as one block, 4 = those launches not issued at all - what any fusion of them could at most give
#ifndef ISX_TAIL_ABL
#define ISX_TAIL_ABL 0
#endif
// blend() of a fully deferred cycle: Gaussian chains of all tiles, top gather, gathering collapse chain
template <int M, int SK>
int run_blend_deferred_t(isx_blender* b, const OutMat& out) {
    hipStream_t st = b->stream;
    const int L = b->num_bands, prec = M, n = (int)b->tiles.size();
    b->path_cycle = 1; b->path_last = 0;
    LevelBuf* d = b->dst;
    // The collapsed levels out_k (k >= 1) only live between two steps of this chain.  The fp32 / fp16 destination levels
    // are float4 records already; for I16 (OpenCV's short4 + weight plane) they go to a private arena in the tile-level
    // format {int b, g, r; float w} instead, so that every precision stages out_k by LDS-DMA (OUT_DST in the kernel).
    LevelBuf od[MAX_LEVELS];
    if (M == M_I16) {
        size_t total = 0;
        layout_levels(od, L, d[0].rows, d[0].cols, prec, false, nullptr, &total);
        const size_t skip = ((size_t)d[0].rows * d[0].cols * g_px_bytes(prec) + 255) & ~(size_t)255;   // level 0 goes to the caller's mat
        ISX_TRY(b->out_arena.reserve(total - skip + 256));
        layout_levels(od, L, d[0].rows, d[0].cols, prec, false, (char*)b->out_arena.p - skip, &total);
        od[0].img = nullptr;
        d = od;
    }
    // One TileDesc per tile and launch (b->td, reused); make_views() hands them to the kernel - in its arguments up to DEF_MAX tiles, as a device
    // table beyond.  Every launch of the chain has its own slot of the table: L pyrDown levels, the top launch, L collapse steps.
    std::vector<TileDesc>& td = b->td;
    auto base = [&](int k_fine) {   // tile rectangles at level k_fine
        td.resize((size_t)n);
        for (int t = 0; t < n; ++t) {
            const isx_blender::TileRec& r = b->tiles[t];
            TileDesc& e = td[(size_t)t];
            memset(&e, 0, sizeof(e));
            e.s0 = r.s0;
            e.x_tl = r.x_tl >> k_fine; e.y_tl = r.y_tl >> k_fine;
            e.w = r.g[k_fine].cols; e.h = r.g[k_fine].rows;
            e.bx_lo = 0; e.bx_hi = 1 << 30;
        }
    };
    const size_t tab_stride = tab_slot_stride(n, d[0].cols);
    if (n > DEF_MAX) ISX_TRY(b->tab.ensure(st, (size_t)(2 * L + 2) * tab_stride));
    auto views = [&](int slot, int gshift, int range_cols, TileViews* v) { return make_views(&b->tab, &b->tab_sc, st, slot, tab_stride, td, gshift, range_cols, v); };
    // Column window (see step 3): the columns of every level the window's pixels depend on, need_k, and from them the columns of the
    // tiles' Gaussian levels that have to be PRODUCED: prod_L = need_L, prod_k = need_k widened by what pyrDown reads for prod_{k+1}
    // (columns 2c - 2 .. 2c + 2).  A tile's chain runs only the blocks that hold them; the rest of its levels keeps whatever it held.
    int need_lo[MAX_LEVELS], need_hi[MAX_LEVELS], prod_lo[MAX_LEVELS], prod_hi[MAX_LEVELS];
    const bool windowed = b->win_x1 > b->win_x0;
    need_lo[0] = windowed ? b->win_x0 : 0; need_hi[0] = windowed ? std::min(b->win_x1, b->fw) : d[0].cols;
    for (int k = 1; k <= L; ++k) {
        need_lo[k] = windowed ? std::max(need_lo[k - 1] / 2 - 1, 0) : 0;
        need_hi[k] = windowed ? std::min((need_hi[k - 1] - 1) / 2 + 2, d[k].cols) : d[k].cols;
    }
    prod_lo[L] = need_lo[L]; prod_hi[L] = need_hi[L];
    for (int k = L - 1; k >= 0; --k) {
        prod_lo[k] = std::max(std::min(need_lo[k], 2 * prod_lo[k + 1] - 2), 0);
        prod_hi[k] = std::min(std::max(need_hi[k], 2 * (prod_hi[k + 1] - 1) + 3), d[k].cols);
    }
    const double gin0 = src_px_bytes(SK) + 1.0;
    // 0. Which kernel will run the last step?  Known before anything is launched, because two formats follow from it (round 4): when it is
    //    k_collapse_roll and level 1 of the collapsed pyramid comes from k_collapse_gather (L = 2, or L >= 5 with k_collapse_top above it),
    //    out_1 is written as dense 12-byte image records (OutMat::rec12) and level 1 of every tile PLANAR (load_px_planar) - the last step
    //    reads the image channels of both and never their fourth dword: 26.9 MB of 265 per 4K pair fetched for nothing with 16-byte records.
    //    ISX_OUT12=0 / ISX_G1P=0: the 16-byte records (A/B runs).
    int roll_var = 0;
    bool rec12 = false, g1_planar = false;
    if (L >= 2) {
        static const bool top_on0 = [] { const char* e = getenv("ISX_TOP"); return !(e && e[0] == '0'); }();
        static const bool out12_on = [] { const char* e = getenv("ISX_OUT12"); return !(e && e[0] == '0'); }();
        static const bool g1p_on = [] { const char* e = getenv("ISX_G1P"); return !(e && e[0] == '0'); }();
        base(0);
        for (int t = 0; t < n; ++t) { td[(size_t)t].fine = b->tiles[t].g[0]; td[(size_t)t].coarse = b->tiles[t].g[1]; }
        roll_var = roll_variant<SK>(td, d[1], need_lo[0] / 2, std::min((need_hi[0] + 1) / 2, d[1].cols));
        const int D0 = std::min(TOP_DMAX, L - 1);
        const bool lvl1_by_gather = !(top_on0 && D0 >= 2) || L - D0 >= 2;
        rec12 = roll_var != 0 && lvl1_by_gather && out12_on;
        g1_planar = roll_var != 0 && lvl1_by_gather && g1p_on && (M == M_F32 || M == M_I16);
    }
    // Level 1 in Q8 records (load_px_planar): fp32 pyramids over tiles of bytes - the caller's CV_8UC3 tiles, their private copies, narrowed copies of
    // CV_16SC3 tiles (if those turn out not to be bytes the cycle is widened: level 1 is produced again, see narrow_resolve's callers below)
    bool g1_q8 = false;
    if constexpr (M == M_F32 && SK == SK_U8) {
        static const bool q8_on = [] { const char* e = getenv("ISX_G1Q8"); return !(e && e[0] == '0'); }();
        bool side = b->chain_on_side.size() >= (size_t)n;       // (chains launched by feed() wrote 16-byte records: all_on_side below)
        for (int t = 0; t < n && side; ++t) side = b->chain_on_side[t] != 0;
        g1_q8 = g1_planar && q8_on && !side;
        for (int t = 0; t < n && g1_q8; ++t) {      // (SK_U8: the caller's CV_8UC3 tiles, private CV_8UC3 copies, narrowed copies of CV_16SC3 tiles)
            const int s = b->tiles[t].g1;
            g1_q8 = s == 0 || s == 3 || s == 4;      // a level 1 that feed() wrote in another layout keeps that layout
        }
    }
    const double g1_b = g1_q8 ? 10.0 : alg_g(prec), g1_rgb_b = g1_q8 ? 6.0 : alg_g_rgb(prec);      // algorithmic bytes of a level-1 record / its image channels
    auto planar_of = [g1_q8](LevelBuf g) { g.wgt = (float*)((char*)g.img + planar_wgt_offset(M, g.rows, g.cols, g1_q8)); return g; };
    // A pair with three fused top steps: k_collapse_top2 (collapse_top2.inc) rebuilds the tiles' level L itself, out of the one read of level L - 1
    // it makes anyway - the pyrDown launch that produces level L is not issued.  ISX_TOP2=0: k_collapse_top behind that launch (A/B runs).
    bool use_top2 = false;
    {
        static const bool top_on2 = [] { const char* e = getenv("ISX_TOP"); return !(e && e[0] == '0'); }();
        static const bool top2_on = [] { const char* e = getenv("ISX_TOP2"); return !(e && e[0] == '0'); }();
        use_top2 = top_on2 && top2_on && n <= 2 && std::min(TOP_DMAX, L - 1) == 3;
    }
    // 1. Gaussian chains: one launch per level for all tiles.  (Per-tile chains on side streams, started
    //    by feed() to overlap with the next tile's VALU-bound warp, were measured: no gain — a kernel that
    //    fills every wave slot leaves nothing for a concurrent one — so the simpler form stays.)
    bool all_on_side = b->chain_on_side.size() >= (size_t)n;
    for (int t = 0; t < n && all_on_side; ++t) all_on_side = b->chain_on_side[t] != 0;
    if (all_on_side) { ISX_TRY(join_side_streams(b)); g1_planar = false; }     // (chains launched by feed() wrote 16-byte records)
    b->path_g1 = g1_planar ? (g1_q8 ? 2 : 1) : 0;
    // Level 1 of tiles that came through the fused feed (k_feed_pd0) exists already, in the layout feed() expected this chain to want; when every
    // tile has it, whole and in that layout, the level-0 launch is skipped.  Otherwise level 1 is produced here from the private copies - whose type
    // must be known for that: the narrowed copies are confirmed first (a widened cycle re-enters as a cycle of CV_16SC3 tiles).
    bool g1_done = true;
    const int g1_state = g1_planar ? (g1_q8 ? 4 : 2) : 1;
    for (int t = 0; t < n; ++t) g1_done = g1_done && b->tiles[t].g1 == g1_state;
    if (b->narrow_pending) {
        if (!(g1_done && L >= 2 && !all_on_side)) ISX_TRY(narrow_publish(b));      // (otherwise the level 1 -> 2 pyrDown below carries the words)
        if (!g1_done) {
            bool widened = false;
            ISX_TRY(narrow_resolve(b, &widened));
            if constexpr (SK == SK_U8) { if (widened) return run_blend_deferred_t<M, SK_S16>(b, out); }
        }
    }
    for (int k = 0; k < L && !all_on_side; ++k) {
        if (k == 0 && g1_done) {
            if (k == b->mark_level && b->mark_event) ISX_HIP(hipEventRecord(b->mark_event, st));
            continue;
        }
        base(k);
        int maxc = 0, maxr = 0;
        double bytes = 0.0;
        for (int t = 0; t < n; ++t) {
            const isx_blender::TileRec& r = b->tiles[t];
            TileDesc& e = td[(size_t)t];
            e.fine = r.g[k]; e.coarse = r.g[k + 1];
            if (g1_planar && k == 0) e.coarse = planar_of(r.g[1]);
            if (g1_planar && k == 1) e.fine = planar_of(r.g[1]);
            maxc = std::max(maxc, r.g[k + 1].cols); maxr = std::max(maxr, r.g[k + 1].rows);
            double share = 1.0;
            if (windowed) {   // the tile's columns of level k + 1 inside prod_{k+1}, as block columns of its own grid
                const int x_t = r.x_tl >> (k + 1), nbx = cdiv(r.g[k + 1].cols, PD_OW);
                const int c0 = std::max(prod_lo[k + 1] - x_t, 0), c1 = std::min(prod_hi[k + 1] - x_t, r.g[k + 1].cols);
                e.bx_lo = c1 > c0 ? c0 / PD_OW : 0; e.bx_hi = c1 > c0 ? cdiv(c1, PD_OW) : 0;
                share = (double)(e.bx_hi - e.bx_lo) / nbx;
            }
            bytes += share * ((double)r.g[k].rows * r.g[k].cols * (k == 0 ? gin0 : (k == 1 ? g1_b : alg_g(prec))) + (double)r.g[k + 1].rows * r.g[k + 1].cols * (k == 0 ? g1_b : alg_g(prec)));
        }
        dim3 grid(cdiv(maxc, PD_OW), cdiv(maxr, PD_TY), n);
        if ((ISX_TAIL_ABL & 1) && k >= 2) grid = dim3(1, 1, n);
        if ((ISX_TAIL_ABL & 4) && k >= 2) continue;
        if ((ISX_TAIL_ABL & 8) && k == 1) continue;                            // round 6 (tools/probes/level2_ablation.sh): the level 1 -> 2 launch not issued
        if ((ISX_TAIL_ABL & 16) && k == 0) grid.y = (grid.y * 13 + 9) / 10;    // ... and the level-0 pyrDown doing 1.3 x its work (pyr_down0_body wraps the rows)
        if (use_top2 && k == L - 1) {       // level L is rebuilt inside k_collapse_top2
            if (k == b->mark_level && b->mark_event) ISX_HIP(hipEventRecord(b->mark_event, st));
            continue;
        }
        TileViews v;
        ISX_TRY(views(k, 0, 0, &v));
        if (k == 0) {
            ISX_TRY((launch_pyr_down0<M, SK>(v, grid, bytes, st, g1_planar ? (g1_q8 ? 2 : 1) : 0)));
            // Level 1 of these tiles now holds what THIS chain wants, on the columns this chain needs.  A tile whose level 1 came from feed() (g1 = 1 / 2)
            // no longer has it whole in that layout when only a window's columns were rewritten (3 = fused-fed, level 1 to be produced again): a later
            // column strip of the same cycle that shares the tile (run_blend_deferred_strips) must not take it for done.  (Found by the fuzzer in round 5:
            // 35 tiles, two bands, a strip with more than three tiles over one place took k_collapse_gather and 16-byte records, its neighbour read the
            // shared tile's planar level 1 behind it.)
            for (int t = 0; t < n; ++t)
                if (b->tiles[t].g1 != 0) b->tiles[t].g1 = windowed ? 3 : g1_state;
        } else {
            FeedPub fp{nullptr, 0, nullptr, 0};
            if (b->narrow_pending && !b->narrow_published) {      // the violation words ride on this launch (the first of the chain: level 1 came from feed())
                fp = FeedPub{(unsigned*)b->feed_state.p, n, b->feed_pin, ++b->feed_seq};
                b->narrow_published = true;
            }
            if (k == 1 && g1_q8) {
                if constexpr (M == M_F32) {
                    if (v.tab) ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi_tab<M, SK_LEVEL, 2>), grid, dim3(512), 0, v.tt, fp);
                    else ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi<M, SK_LEVEL, 2>), grid, dim3(512), 0, v.ts, fp);
                }
            } else if (k == 1 && g1_planar) {
                if constexpr (M == M_F32 || M == M_I16) {
                    if (v.tab) ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi_tab<M, SK_LEVEL, 1>), grid, dim3(512), 0, v.tt, fp);
                    else ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi<M, SK_LEVEL, 1>), grid, dim3(512), 0, v.ts, fp);
                }
            } else if (v.tab) ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi_tab<M, SK_LEVEL>), grid, dim3(512), 0, v.tt, fp);
            else ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi<M, SK_LEVEL>), grid, dim3(512), 0, v.ts, fp);
        }
        // the full-size level-0 kernel is behind us: from here to the last collapse step the launches are small
        if (k == b->mark_level && b->mark_event) ISX_HIP(hipEventRecord(b->mark_event, st));
    }
    // 2. the top level out_L = norm(SUM_t cast(G_{L,t} W_{L,t})) is gathered inside the first collapse step (TOP) while it
    //    stages its coarse tile: no launch, no level-L buffer.
    //    (algorithmic bytes of the launches below = the traffic their own dataflow needs: every input
    //    record read once, every output written once; the destination pyramid of the eager path and of
    //    SURVEY's model does not exist here)
    // 3. collapse chain; each step gathers the tiles' Laplacians of its fine level in registers
    // Column window (isx_blender_set_window): the step that produces level k - 1 runs only the block columns that hold the columns of
    // level k - 1 the window needs: need_0 = the window, need_k = need_{k-1} halved and widened by the one coarse column pyrUp reads on
    // either side.  Whatever else those blocks compute (and whatever they read outside need_k: levels left over from an earlier cycle)
    // reaches no pixel of the window - every input of a pixel of need_{k-1} lies in need_k by construction - so the window's pixels
    // are those of the whole mosaic, bit for bit, and a mosaic can be cut into column strips computed on different GPUs.
    // The steps above the last two or three levels in ONE launch (collapse_top.inc): out_L .. out_{kout + 1} then never touch memory.
    // ISX_TOP=0: one launch per step (A/B runs).
    int k_first = L;
    {
        static const bool top_on = [] { const char* e = getenv("ISX_TOP"); return !(e && e[0] == '0'); }();
        const int D = std::min(TOP_DMAX, L - 1), kout = L - D;
        if (top_on && D >= 2) {
            std::vector<TopDesc>& tp = b->tpd;
            tp.resize((size_t)n);
            double bytes = (double)d[kout].rows * d[kout].cols * alg_d_rgb(prec);
            for (int t = 0; t < n; ++t) {
                const isx_blender::TileRec& r = b->tiles[t];
                TopDesc& e = tp[(size_t)t];
                memset(&e, 0, sizeof(e));
                e.x_tl = r.x_tl >> kout; e.y_tl = r.y_tl >> kout; e.w = r.g[kout].cols; e.h = r.g[kout].rows;
                for (int i = 0; i <= D; ++i) {
                    e.g[i] = r.g[kout + i].img;
                    // every level once as a fine level (record + weight; the top one as the gathered top), every level but the output's once as a pyrUp source
                    bytes += (double)r.g[kout + i].rows * r.g[kout + i].cols * (alg_g(prec) + (i > 0 ? alg_g_rgb(prec) : 0.0));
                }
            }
            const int gx_all = cdiv(d[kout].cols, TOP_BW);
            const int bx_lo = need_lo[kout] / TOP_BW, bx_hi = std::min(cdiv(need_hi[kout], TOP_BW), gx_all);
            dim3 grid(bx_hi - bx_lo, cdiv(d[kout].rows, TOP_BH));
            if (ISX_TAIL_ABL & 2) grid = dim3(1, 1);
            bytes *= (double)(bx_hi - bx_lo) / gx_all;
            if (ISX_TAIL_ABL & 4) {
            } else if (n <= DEF_MAX) {
                TopTiles tt;
                memset(&tt, 0, sizeof(tt));
                tt.n = n; tt.D = D;
                for (int t = 0; t < n; ++t) {
                    const TopDesc& e = tp[(size_t)t];
                    tt.x_tl[t] = e.x_tl; tt.y_tl[t] = e.y_tl; tt.w[t] = e.w; tt.h[t] = e.h;
                    for (int i = 0; i <= D; ++i) tt.g[t][i] = e.g[i];
                }
                if (use_top2) ISX_LAUNCH("collapse_top", bytes, st, (k_collapse_top2<M>), grid, dim3(256), 0, tt, d[kout], bx_lo, d[L].rows, d[L].cols);
                else ISX_LAUNCH("collapse_top", bytes, st, (k_collapse_top<M>), grid, dim3(256), 0, tt, d[kout], bx_lo, d[L].rows, d[L].cols);
            } else {
                // the tiles a block has to look at: its cone widens, level by level, by less than 3 * 2^D output-level columns on either side
                TabScratch& sc = b->tab_sc;
                const int m = 3 << D;
                sc.xs.resize((size_t)n); sc.ws.resize((size_t)n);
                for (int t = 0; t < n; ++t) { sc.xs[(size_t)t] = tp[(size_t)t].x_tl - m; sc.ws[(size_t)t] = tp[(size_t)t].w + 2 * m; }
                int gsh = 0;
                while ((1 << gsh) < TOP_BW) ++gsh;
                static_assert((TOP_BW & (TOP_BW - 1)) == 0, "one range per block column");
                tile_ranges(sc.xs.data(), sc.ws.data(), n, gsh, d[kout].cols, &sc.rng, nullptr);
                const size_t dbytes = (size_t)n * sizeof(TopDesc), rbytes = (sc.rng.size() * sizeof(int2) + 15) & ~(size_t)15;
                ISX_CHECK_ARG(dbytes + rbytes <= tab_stride, ISX_ERR_INTERNAL, "tile table: %zu bytes exceed the slot's %zu", dbytes + rbytes, tab_stride);
                sc.img.assign(dbytes + rbytes, 0);
                memcpy(sc.img.data(), tp.data(), dbytes);
                memcpy(sc.img.data() + dbytes, sc.rng.data(), sc.rng.size() * sizeof(int2));
                const unsigned char* dev = nullptr;
                ISX_TRY(b->tab.put(st, (size_t)L * tab_stride, sc.img.data(), dbytes + rbytes, &dev));
                TopTab tt;
                memset(&tt, 0, sizeof(tt));
                tt.n = n; tt.D = D; tt.nrng = (int)sc.rng.size(); tt.rng = (const int2*)(dev + dbytes);
                const char* bs = (const char*)dev;
                tt.x_tl.base = bs; tt.y_tl.base = bs; tt.w.base = bs; tt.h.base = bs; tt.g.base = bs;
                ISX_LAUNCH("collapse_top", bytes, st, (k_collapse_top_tab<M>), grid, dim3(256), 0, tt, d[kout], bx_lo, d[L].rows, d[L].cols);
            }
            k_first = kout;
        }
    }
    // one collapse step; SKC = the tiles' type as the LAST step reads them (the steps above it read pyramid levels only)
    auto step = [&](int k, auto SKC) -> int {
        constexpr int SKL = decltype(SKC)::value;
        base(k - 1);
        const int gx_all = cdiv(d[k].cols, WAVE);
        const int bx_lo = need_lo[k - 1] / (2 * WAVE), bx_hi = std::min(cdiv(need_hi[k - 1], 2 * WAVE), gx_all);
        const double frac = (double)(bx_hi - bx_lo) / gx_all;                                  // share of the level this launch works on
        double bytes = k == L ? 0.0 : (double)d[k].rows * d[k].cols * alg_d_rgb(prec);      // out_k as pyrUp source (k = L: gathered from G_L)
        for (int t = 0; t < n; ++t) {
            const isx_blender::TileRec& r = b->tiles[t];
            TileDesc& e = td[(size_t)t];
            e.fine = r.g[k - 1]; e.coarse = r.g[k];
            if (g1_planar && k == 2) e.fine = planar_of(r.g[1]);
            if (g1_planar && k == 1) e.coarse = planar_of(r.g[1]);
            if (k == L) bytes += (double)r.g[k].rows * r.g[k].cols * 4.0;                        // + the weights of G_L
            bytes += (double)r.g[k - 1].rows * r.g[k - 1].cols * (k == 1 ? src_px_bytes(SKL) + 1.0 : (k == 2 ? g1_b : alg_g(prec)))   // G_{k-1,t} (level 0: the tile + mask)
                   + (double)r.g[k].rows * r.g[k].cols * (k == 1 ? g1_rgb_b : alg_g_rgb(prec));         // G_{k,t} as pyrUp source
        }
        dim3 grid(bx_hi - bx_lo, cdiv(d[k].rows, UP_TY));
        // (index ranges per 128 columns of the step's fine level = one block of k_collapse_gather; a strip of k_collapse_roll spans two)
        TileViews v;
        ISX_TRY(views(L + 1 + k, 7, d[k - 1].cols, &v));
        v.ts.q8 = v.tt.q8 = (g1_q8 && k <= 2) ? 1 : 0;      // k = 2 reads level 1 as its fine level, k = 1 as its pyrUp source
        OutMat o = out;
        o.bx0 = bx_lo;
        o.rec12 = (rec12 && k <= 2) ? 1 : 0;      // k = 2 writes out_1, k = 1 reads it
        // last two steps (levels 0 and 1; the level-1 step runs at the fabric's rate: 48.5 -> 46.2 us for the four upper steps; no gain
        // from level 2 up): XCD-aware block order in groups of 2 block rows (see OutMat).  Measured on the 4K pair (tools/measure_traffic.py,
        // profiles/round2_xcd_order.txt): fabric traffic per launch 383 MB in plain row-major order, 282 MB with groups of 2, 268 with 4,
        // 262 with 8 (algorithmic: 229); launch time 67.1 / 65.7 / 68.3 / 73.5 us - larger groups leave the XCDs uneven shares.
        constexpr int XCD_GRP = 2;
        if (k <= 2 && (int)grid.y >= 8 * XCD_GRP) {
            o.grp = XCD_GRP; o.gx = (int)grid.x; o.gy = (int)grid.y; o.xmagic = xcd_magic(XCD_GRP, o.gx);
            grid = dim3(xcd_grid_blocks(XCD_GRP, o.gx, o.gy), 1);
        }
        if (k == 1) {
            bytes = bytes * frac + (double)out.rows * (need_hi[0] - need_lo[0]) * (out.img_f32 == 1 ? 13.0 : (out.img_f32 == 2 ? 4.0 : 7.0));   // result + mask
            if (k != L) {
                bool done = false;
                OutMat o1 = out;
                o1.rec12 = rec12 ? 1 : 0;
                ISX_TRY((launch_collapse_roll<M, SKL>(b, st, v, d[1], o1, need_lo[0] / 2, std::min((need_hi[0] + 1) / 2, d[1].cols), bytes, roll_var, &done)));
                ISX_CHECK_ARG(done || !(rec12 || g1_planar), ISX_ERR_STATE, "blend: the last step's kernel was planned as k_collapse_roll and did not run");
                if (done) { b->path_last = 3; return ISX_OK; }
            }
            b->path_last = 2;
            if (v.tab) {
                if (k == L) ISX_LAUNCH("collapse_gather_final", bytes, st, (k_collapse_gather_tab<M, SKL, true, true>), grid, dim3(256), 0, v.tt, d[1], d[0], o);
                else ISX_LAUNCH("collapse_gather_final", bytes, st, (k_collapse_gather_tab<M, SKL, true, false>), grid, dim3(256), 0, v.tt, d[1], d[0], o);
            } else if (k == L) ISX_LAUNCH("collapse_gather_final", bytes, st, (k_collapse_gather<M, SKL, true, true>), grid, dim3(256), 0, v.ts, d[1], d[0], o);
            else ISX_LAUNCH("collapse_gather_final", bytes, st, (k_collapse_gather<M, SKL, true, false>), grid, dim3(256), 0, v.ts, d[1], d[0], o);
        } else {
            bytes = (bytes + (double)d[k - 1].rows * d[k - 1].cols * alg_d_rgb(prec)) * frac;  // + out_{k-1}
            if (v.tab) {
                if (k == L) ISX_LAUNCH("collapse_gather", bytes, st, (k_collapse_gather_tab<M, SK_U8, false, true>), grid, dim3(256), 0, v.tt, d[k], d[k - 1], o);
                else ISX_LAUNCH("collapse_gather", bytes, st, (k_collapse_gather_tab<M, SK_U8, false, false>), grid, dim3(256), 0, v.tt, d[k], d[k - 1], o);
            } else if (k == L) ISX_LAUNCH("collapse_gather", bytes, st, (k_collapse_gather<M, SK_U8, false, true>), grid, dim3(256), 0, v.ts, d[k], d[k - 1], o);
            else ISX_LAUNCH("collapse_gather", bytes, st, (k_collapse_gather<M, SK_U8, false, false>), grid, dim3(256), 0, v.ts, d[k], d[k - 1], o);
        }
        return ISX_OK;
    };
    for (int k = k_first; k >= 1; --k) {
        if (k == 1 && b->narrow_pending) {
            // the chain up to here never looked at a level-0 pixel; the last step does: are the narrowed copies good?  (The answer has been on
            // its way since this blend() began - k_feed_publish - and the launches above keep the GPU busy while the host reads it.)
            bool widened = false;
            ISX_TRY(narrow_resolve(b, &widened));
            if constexpr (SK == SK_U8) {
                if (widened) {      // (the roll variant was chosen on the tiles' geometry, which widening does not change)
                    if (g1_q8) {        // the chain above read a level 1 in Q8 records that feed() wrote from shorts that were not bytes: all of it again
                        for (int t = 0; t < n; ++t) if (b->tiles[t].g1 != 0) b->tiles[t].g1 = 3;
                        return run_blend_deferred_t<M, SK_S16>(b, out);
                    }
                    if (roll_var != 0) {
                        bool ok = true;
                        for (int t = 0; t < n; ++t) ok = ok && b->tiles[t].s0.iend != 0u;
                        ISX_CHECK_ARG(ok, ISX_ERR_UNSUPPORTED, "blend: a widened tile exceeds the 2 GiB the planned last step addresses");
                    }
                    ISX_TRY(step(1, IC<SK_S16>{}));
                    continue;
                }
            }
        }
        ISX_TRY(step(k, IC<SK>{}));
    }
    return ISX_OK;
}
template <int M>
int run_blend_deferred(isx_blender* b, const OutMat& out) {
    switch (b->tiles[0].sk) {
        case SK_U8: return run_blend_deferred_t<M, SK_U8>(b, out);
        case SK_S16: return run_blend_deferred_t<M, SK_S16>(b, out);
        default: return run_blend_deferred_t<M, SK_F32>(b, out);
    }
}

// A deferred cycle of MORE than DEF_MAX tiles (a long panorama: BASELINE config 4's 64 tiles in one mosaic): the result is produced in column
// strips, each by the deferred chain over the tiles that can reach it - the rule and the machinery of isx_blender_set_window (every strip equals
// the same columns of the whole blend bit for bit, tests/test_gpu_strips.py), applied by the library itself.  Strips are multiples of
// ISX_WINDOW_GRANULE wide and as wide as DEF_MAX tiles allow (greedy from the left); a tile that reaches several strips has the columns of its
// Gaussian levels each strip needs produced once per strip.  *done = false: some 128-column strip is reached by more than DEF_MAX tiles (tiles
// stacked in many rows) - the caller falls back to the eager cycle.
template <int M>
int run_blend_deferred_strips(isx_blender* b, const OutMat& out, bool* done) {
    *done = false;
    const int L = b->num_bands;
    { bool widened; ISX_TRY(narrow_resolve(b, &widened)); }      // every strip blends a subset of the records: their type is settled first
    const bool user_win = b->win_x1 > b->win_x0;
    const int X0 = user_win ? b->win_x0 : 0, X1 = user_win ? std::min(b->win_x1, b->fw) : b->fw;
    // Which tiles can reach the columns [x0, x1) of the result?  mosaic.tiles_for_window's rule - the tile's fed rectangle meets, at some level k,
    // the columns need_k of that level the window depends on (need_k = need_{k-1} halved and widened by pyrUp's one column on either side) - bounded
    // from outside by ONE interval per tile: need_k lies within (x0 / 2^k - 3, x1 / 2^k + 3), so a tile that reaches the window has
    // x_tl - 4 * 2^L < x1 and x0 < x_tl + width + 3 * 2^L.  Tiles inside the interval that do not reach the window change none of its pixels (every
    // input of a pixel of need_{k-1} lies in need_k): a superset is as exact as the set itself, and planning is interval counting.
    const int n_all = (int)b->tiles.size(), m5 = 5 << L, m3 = 3 << L;
    std::vector<std::pair<int, int>> iv((size_t)n_all);            // (lo, hi): the window [x0, x1) takes the tile iff x0 < hi && x1 > lo
    for (int t = 0; t < n_all; ++t) iv[(size_t)t] = std::make_pair(b->tiles[(size_t)t].x_tl - m5, b->tiles[(size_t)t].x_tl + b->tiles[(size_t)t].width + m3);
    std::vector<int> by_lo((size_t)n_all);
    for (int t = 0; t < n_all; ++t) by_lo[(size_t)t] = t;
    std::sort(by_lo.begin(), by_lo.end(), [&](int a, int c) { return iv[(size_t)a].first < iv[(size_t)c].first; });
    std::vector<std::pair<int, int>> wins;
    for (int x = X0; x < X1;) {
        // the window starts at x and grows until the (DEF_MAX + 1)-th tile (in the order of their intervals' left ends, those that end at or before x skipped) begins
        int cnt = 0, limit = X1;
        for (int q = 0; q < n_all; ++q) {
            const std::pair<int, int>& v = iv[(size_t)by_lo[(size_t)q]];
            if (v.second <= x) continue;
            if (++cnt > DEF_MAX) { limit = std::max(v.first, 0); break; }
        }
        int e = X1;
        if (limit < X1) {
            e = (limit / ISX_WINDOW_GRANULE) * ISX_WINDOW_GRANULE;      // x1 > lo is strict: a window ending AT the tile's left end does not take it
            if (e <= x) return ISX_OK;                                  // more than DEF_MAX tiles over the first granule: the caller goes eager
        }
        wins.emplace_back(x, e);
        x = e;
    }
    auto reach = [&](int x0, int x1, std::vector<int>* idx) {
        for (int t = 0; t < n_all; ++t)
            if (x0 < iv[(size_t)t].second && x1 > iv[(size_t)t].first) idx->push_back(t);
    };
    ISX_TRY(join_side_streams(b));       // chains that feed() started wrote whole levels of single tiles: the strips produce their own columns
    std::vector<isx_blender::TileRec> all;
    std::vector<char> on_side;
    all.swap(b->tiles);
    on_side.swap(b->chain_on_side);
    const int wx0 = b->win_x0, wx1 = b->win_x1;
    int rc = ISX_OK, last = 0;
    for (size_t j = 0; j < wins.size() && rc == ISX_OK; ++j) {
        std::vector<int> idx;
        reach(wins[j].first, wins[j].second, &idx);
        b->tiles.clear();
        for (int t : idx) b->tiles.push_back(all[(size_t)t]);
        OutMat o = out;
        o.cols = std::min(wins[j].second, b->fw);
        if (b->tiles.empty()) {          // no tile reaches the strip: Blender::blend zeroes what no weight covers (W:313)
            b->tiles.push_back(all[0]);
        }
        b->win_x0 = wins[j].first; b->win_x1 = wins[j].second;
        rc = run_blend_deferred<M>(b, o);
        last = std::max(last, b->path_last);
        if (b->tiles.size() == idx.size())      // what the strip's chain did to the state of its tiles' level 1 (see the level-0 launch there)
            for (size_t i = 0; i < idx.size(); ++i) all[(size_t)idx[i]].g1 = b->tiles[i].g1;
    }
    b->tiles.swap(all);
    b->chain_on_side.swap(on_side);
    b->win_x0 = wx0; b->win_x1 = wx1;
    ISX_TRY(rc);
    b->path_cycle = 3; b->path_last = last;
    *done = true;
    return ISX_OK;
}

template <int M>
int run_blend(isx_blender* b, const OutMat& out) {
    hipStream_t st = b->stream;
    const int L = b->num_bands, prec = M;
    LevelBuf* d = b->dst;
    if (b->level0_pending) {
        if (b->tiles.size() <= (size_t)DEF_MAX) return run_blend_deferred<M>(b, out);
        // More than DEF_MAX tiles (round 5): ONE chain over all of them, their descriptors in a device table (TileTab; cycle 4 in isx_blender_last_path).
        // ISX_TAB=0: round 4's column strips of at most DEF_MAX tiles (run_blend_deferred_strips), ISX_STRIPS=0 as well: the eager cycle (A/B runs).
        // One guard of the int16 arithmetic leans on the tile count: a sum of CV_8UC3 Laplacians (+-255 each) cannot wrap a short while at most 128
        // tiles meet in a pixel - k_collapse_gather's last step does not issue the wrap for them - so a cycle with a deeper stack than that over
        // some 128 columns takes the strips / the eager cycle as before.
        const char* tab_env = getenv("ISX_TAB");       // (read per call: the tests switch it inside one process)
        const bool tab_on = !(tab_env && tab_env[0] == '0');
        bool tab_ok = tab_on;
        if (tab_ok && M == M_I16) {
            TabScratch& sc = b->tab_sc;
            const int n = (int)b->tiles.size();
            sc.xs.resize((size_t)n); sc.ws.resize((size_t)n);
            for (int t = 0; t < n; ++t) { sc.xs[(size_t)t] = b->tiles[(size_t)t].x_tl; sc.ws[(size_t)t] = b->tiles[(size_t)t].width; }
            int deepest = 0;
            tile_ranges(sc.xs.data(), sc.ws.data(), n, 7, d[0].cols, &sc.rng, &deepest);
            tab_ok = deepest <= 128;
        }
        if (tab_ok) {
            ISX_TRY(run_blend_deferred<M>(b, out));
            b->path_cycle = 4;
            return ISX_OK;
        }
        static const bool strips_on = [] { const char* e = getenv("ISX_STRIPS"); return !(e && e[0] == '0'); }();    // ISX_STRIPS=0: the eager cycle, as before round 4 (A/B runs)
        bool done = false;
        if (strips_on) ISX_TRY(run_blend_deferred_strips<M>(b, out, &done));
        if (done) return ISX_OK;
        ISX_CHECK_ARG(b->win_x1 <= b->win_x0, ISX_ERR_UNSUPPORTED, "blend: more than %d tiles reach a %d-column strip of the window - a column window needs the deferred cycle",
                      DEF_MAX, ISX_WINDOW_GRANULE);
        ISX_TRY(flush_deferred(b));      // the cycle continues eagerly
    }
    b->path_cycle = 0; b->path_last = L >= 1 ? 1 : 0;
    if (L == 0) {
        dim3 grid(cdiv(d[0].cols, 64), cdiv(d[0].rows, 4));
        double px = (double)d[0].rows * d[0].cols;
        ISX_LAUNCH("norm_top_final", px * alg_d(prec) + (double)out.rows * out.cols * (out.img_f32 == 1 ? 13.0 : (out.img_f32 == 2 ? 4.0 : 7.0)), st, (k_norm_top<M, true>), grid, dim3(256), 0, d[0], out, make_cover(b, 0));
        return ISX_OK;
    }
    // the top level is normalised while it is staged as the coarse tile of the first collapse step
    for (int k = L; k >= 1; --k) {
        dim3 grid(cdiv(d[k].cols, WAVE), cdiv(d[k].rows, UP_TY));
        double coarse_px = (double)d[k].rows * d[k].cols, fine_px = (double)d[k - 1].rows * d[k - 1].cols;
        double cb = coarse_px * (k == L ? alg_d(prec) : alg_d_rgb(prec));
        Cover cov = make_cover(b, k - 1), ccov = make_cover(b, k);
        const bool normc = k == L;
        if (k == 1) {
            double bytes = cb + (double)out.rows * out.cols * (alg_d(prec) + (out.img_f32 == 1 ? 13.0 : (out.img_f32 == 2 ? 4.0 : 7.0)));
            if (normc) ISX_LAUNCH("collapse_final", bytes, st, (k_collapse<M, true, true>), grid, dim3(256), 0, d[1], d[0], out, cov, ccov);
            else ISX_LAUNCH("collapse_final", bytes, st, (k_collapse<M, true, false>), grid, dim3(256), 0, d[1], d[0], out, cov, ccov);
        } else {
            double bytes = cb + fine_px * (alg_d(prec) + alg_d_rgb(prec));
            if (normc) ISX_LAUNCH("collapse", bytes, st, (k_collapse<M, false, true>), grid, dim3(256), 0, d[k], d[k - 1], out, cov, ccov);
            else ISX_LAUNCH("collapse", bytes, st, (k_collapse<M, false, false>), grid, dim3(256), 0, d[k], d[k - 1], out, cov, ccov);
        }
    }
    return ISX_OK;
}

// blend() of nb fully deferred cycles in ONE chain of launches (isx_blender_blend_batch): the Gaussian chains of all tiles of all
// mosaics level by level, then the collapse chain with the mosaic as grid.z.  The caller has checked that the blenders agree in
// precision, band count, tile type, device and stream, hold no window, and that their tiles fit one TileSet.
template <int M, int SK>
int run_blend_batch_t(isx_blender** bs, int nb, const OutMat* outs) {
    isx_blender* b0 = bs[0];
    hipStream_t st = b0->stream;
    const int L = b0->num_bands, prec = M;
    LevelBuf* d[BATCH_MAX];
    LevelBuf od[BATCH_MAX][MAX_LEVELS];
    int first[BATCH_MAX + 1], nt = 0;
    for (int m = 0; m < nb; ++m) {
        isx_blender* b = bs[m];
        d[m] = b->dst;
        if (M == M_I16) {      // the collapsed levels as 16-byte register records (see run_blend_deferred_t)
            size_t total = 0;
            layout_levels(od[m], L, d[m][0].rows, d[m][0].cols, prec, false, nullptr, &total);
            const size_t skip = ((size_t)d[m][0].rows * d[m][0].cols * g_px_bytes(prec) + 255) & ~(size_t)255;
            ISX_TRY(b->out_arena.reserve(total - skip + 256));
            layout_levels(od[m], L, d[m][0].rows, d[m][0].cols, prec, false, (char*)b->out_arena.p - skip, &total);
            od[m][0].img = nullptr;
            d[m] = od[m];
        }
        first[m] = nt;
        nt += (int)b->tiles.size();
        ISX_TRY(join_side_streams(b));
    }
    first[nb] = nt;
    for (int m = 0; m < nb; ++m) { bs[m]->path_cycle = 2; bs[m]->path_last = 2; }
    auto base = [&](int k_fine) {
        TileSet ts;
        memset(&ts, 0, sizeof(ts));
        ts.n = nt;
        int t = 0;
        for (int m = 0; m < nb; ++m)
            for (const isx_blender::TileRec& r : bs[m]->tiles) {
                ts.s0[t] = r.s0;
                ts.x_tl[t] = r.x_tl >> k_fine; ts.y_tl[t] = r.y_tl >> k_fine;
                ts.w[t] = r.g[k_fine].cols; ts.h[t] = r.g[k_fine].rows;
                ts.bx_lo[t] = 0; ts.bx_hi[t] = 1 << 30;
                ++t;
            }
        return ts;
    };
    auto tile_rec = [&](int t) -> const isx_blender::TileRec& {
        int m = 0;
        while (t >= first[m + 1]) ++m;
        return bs[m]->tiles[(size_t)(t - first[m])];
    };
    const double gin0 = src_px_bytes(SK) + 1.0;
    // 0. the last step as the rolling kernel when every mosaic qualifies (roll_variant's conditions, two tiles per strip at most) - decided
    //    before anything is launched: level 1 then goes without its dead fourth dword, as in run_blend_deferred_t (rec12, planar G_1)
    bool roll_ok = false;
    unsigned roll_blocks = 0;
    int r_grp[BATCH_MAX], r_gx[BATCH_MAX], r_gy[BATCH_MAX];
    if (L >= 2) {
        static const int mode = [] { const char* e = getenv("ISX_ROLL"); return e ? atoi(e) : 1; }();
        const TileSet ts = base(0);
        bool ok = mode != 0 && (SK == SK_U8 || SK == SK_S16);
        for (int m = 0; m < nb && ok; ++m) {
            const LevelBuf& c = d[m][1];
            ok = c.cols >= 2 && (unsigned long long)c.rows * c.cols * 16ull < (1ull << 32);
            TileSet one;
            memset(&one, 0, sizeof(one));
            one.n = first[m + 1] - first[m];
            for (int t = 0; t < one.n && ok; ++t) {
                const int g = first[m] + t;
                const LevelBuf& g1 = tile_rec(g).g[1];
                one.x_tl[t] = ts.x_tl[g]; one.y_tl[t] = ts.y_tl[g]; one.w[t] = ts.w[g]; one.h[t] = ts.h[g];
                ok = g1.cols >= 2 && ts.s0[g].cols >= 2 && ts.s0[g].rows >= 2 && ts.s0[g].iend != 0u &&
                     (unsigned long long)g1.rows * g1.cols * 16ull < (1ull << 32);
            }
            ok = ok && roll_max_tiles(one, c, 0, c.cols, 2) <= 2;
            if (ok) {
                const int nsx = cdiv(c.cols, RL_CW), nby = cdiv(cdiv(c.rows, 2), ROLL_WAVES), grp = std::max(2, nsx == 1 ? 2 : 1);
                r_grp[m] = grp; r_gx[m] = nsx; r_gy[m] = nby;
                roll_blocks = std::max(roll_blocks, xcd_band_blocks(grp, nsx, nby));
            }
        }
        roll_ok = ok;
    }
    static const bool out12_on = [] { const char* e = getenv("ISX_OUT12"); return !(e && e[0] == '0'); }();
    static const bool g1p_on = [] { const char* e = getenv("ISX_G1P"); return !(e && e[0] == '0'); }();
    const bool rec12 = roll_ok && out12_on, g1_planar = roll_ok && g1p_on && (M == M_F32 || M == M_I16);
    bool g1_q8 = false;      // level 1 in Q8 records, as in run_blend_deferred_t
    if constexpr (M == M_F32 && SK == SK_U8) {
        static const bool q8_on = [] { const char* e = getenv("ISX_G1Q8"); return !(e && e[0] == '0'); }();
        g1_q8 = g1_planar && q8_on;
        for (int m = 0; m < nb && g1_q8; ++m) g1_q8 = !bs[m]->narrow_pending;
        for (int t = 0; t < nt && g1_q8; ++t) g1_q8 = tile_rec(t).g1 == 0 && tile_rec(t).fed_sk == SK_U8;
    }
    for (int m = 0; m < nb; ++m) bs[m]->path_g1 = g1_planar ? (g1_q8 ? 2 : 1) : 0;
    const double g1_b = g1_q8 ? 10.0 : alg_g(prec), g1_rgb_b = g1_q8 ? 6.0 : alg_g_rgb(prec);
    auto planar_of = [g1_q8](LevelBuf g) { g.wgt = (float*)((char*)g.img + planar_wgt_offset(M, g.rows, g.cols, g1_q8)); return g; };
    // 1. Gaussian chains: one launch per level for every tile of every mosaic
    for (int k = 0; k < L; ++k) {
        TileSet ts = base(k);
        int maxc = 0, maxr = 0;
        double bytes = 0.0;
        for (int t = 0; t < nt; ++t) {
            const isx_blender::TileRec& r = tile_rec(t);
            ts.fine[t] = r.g[k]; ts.coarse[t] = r.g[k + 1];
            if (g1_planar && k == 0) ts.coarse[t] = planar_of(r.g[1]);
            if (g1_planar && k == 1) ts.fine[t] = planar_of(r.g[1]);
            maxc = std::max(maxc, r.g[k + 1].cols); maxr = std::max(maxr, r.g[k + 1].rows);
            bytes += (double)r.g[k].rows * r.g[k].cols * (k == 0 ? gin0 : (k == 1 ? g1_b : alg_g(prec))) + (double)r.g[k + 1].rows * r.g[k + 1].cols * (k == 0 ? g1_b : alg_g(prec));
        }
        dim3 grid(cdiv(maxc, PD_OW), cdiv(maxr, PD_TY), nt);
        if (k == 0) ISX_TRY((launch_pyr_down0<M, SK>(ts, grid, bytes, st, g1_planar ? (g1_q8 ? 2 : 1) : 0)));
        else if (k == 1 && g1_q8) {
            if constexpr (M == M_F32) ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi<M, SK_LEVEL, 2>), grid, dim3(512), 0, ts, FeedPub{nullptr, 0, nullptr, 0});
        } else if (k == 1 && g1_planar) {
            if constexpr (M == M_F32 || M == M_I16) ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi<M, SK_LEVEL, 1>), grid, dim3(512), 0, ts, FeedPub{nullptr, 0, nullptr, 0});
        } else ISX_LAUNCH("pyr_down", bytes, st, (k_pyr_down_multi<M, SK_LEVEL>), grid, dim3(512), 0, ts, FeedPub{nullptr, 0, nullptr, 0});
        for (int m = 0; m < nb; ++m)
            if (k == bs[m]->mark_level && bs[m]->mark_event) ISX_HIP(hipEventRecord(bs[m]->mark_event, st));
    }
    // 2. collapse chain, the mosaic as grid.z
    for (int k = L; k >= 1; --k) {
        TileSet ts = base(k - 1);
        ts.q8 = (g1_q8 && k <= 2) ? 1 : 0;
        BatchOut bo;
        memset(&bo, 0, sizeof(bo));
        std::copy(first, first + nb + 1, bo.first);
        double bytes = 0.0;
        int gx = 0, gy = 0;
        for (int t = 0; t < nt; ++t) {
            const isx_blender::TileRec& r = tile_rec(t);
            ts.fine[t] = r.g[k - 1]; ts.coarse[t] = r.g[k];
            if (g1_planar && k == 2) ts.fine[t] = planar_of(r.g[1]);
            if (g1_planar && k == 1) ts.coarse[t] = planar_of(r.g[1]);
            if (k == L) bytes += (double)r.g[k].rows * r.g[k].cols * 4.0;
            bytes += (double)r.g[k - 1].rows * r.g[k - 1].cols * (k == 1 ? gin0 : (k == 2 ? g1_b : alg_g(prec))) + (double)r.g[k].rows * r.g[k].cols * (k == 1 ? g1_rgb_b : alg_g_rgb(prec));
        }
        for (int m = 0; m < nb; ++m) {
            bo.coarse_out[m] = d[m][k]; bo.fine_out[m] = d[m][k - 1]; bo.out[m] = outs[m];
            bo.out[m].bx0 = 0; bo.out[m].grp = 0; bo.out[m].band = 0;
            bo.out[m].rec12 = (rec12 && k <= 2) ? 1 : 0;       // k = 2 writes out_1, k = 1 reads it
            if (k != L) bytes += (double)d[m][k].rows * d[m][k].cols * alg_d_rgb(prec);
            if (k == 1) bytes += (double)outs[m].rows * outs[m].cols * (outs[m].img_f32 == 1 ? 13.0 : (outs[m].img_f32 == 2 ? 4.0 : 7.0));
            else bytes += (double)d[m][k - 1].rows * d[m][k - 1].cols * alg_d_rgb(prec);
            gx = std::max(gx, cdiv(d[m][k].cols, WAVE)); gy = std::max(gy, cdiv(d[m][k].rows, UP_TY));
        }
        if (k == 1 && roll_ok) {      // the last step as the rolling kernel (decided above)
            for (int m = 0; m < nb; ++m) {
                OutMat& o = bo.out[m];
                o.grp = r_grp[m]; o.gx = r_gx[m]; o.gy = r_gy[m]; o.xmagic = xcd_magic(r_grp[m], r_gx[m]); o.band = cdiv(r_gy[m], 8);
            }
            if constexpr (SK == SK_U8 || SK == SK_S16) {
                ISX_LAUNCH("collapse_roll", bytes, st, (k_collapse_roll_batch<M, SK, 2, 2>), dim3(roll_blocks, 1, nb), dim3(64 * ROLL_WAVES), 0, ts, bo, 0);
                for (int m = 0; m < nb; ++m) bs[m]->path_last = 3;
                continue;
            }
        }
        dim3 grid(gx, gy, nb);
        if (k <= 2 && gy >= 16) {     // the XCD-aware block order of the single path (groups of 2 block rows), every mosaic with its own extent
            unsigned nblk = 0;
            for (int m = 0; m < nb; ++m) {
                OutMat& o = bo.out[m];
                o.grp = 2; o.gx = cdiv(d[m][k].cols, WAVE); o.gy = cdiv(d[m][k].rows, UP_TY); o.xmagic = xcd_magic(2, o.gx);
                nblk = std::max(nblk, xcd_grid_blocks(2, o.gx, o.gy));
            }
            grid = dim3(nblk, 1, nb);
        }
        if (k == 1) {
            if (k == L) ISX_LAUNCH("collapse_gather_final", bytes, st, (k_collapse_gather_batch<M, SK, true, true>), grid, dim3(256), 0, ts, bo);
            else ISX_LAUNCH("collapse_gather_final", bytes, st, (k_collapse_gather_batch<M, SK, true, false>), grid, dim3(256), 0, ts, bo);
        } else {
            if (k == L) ISX_LAUNCH("collapse_gather", bytes, st, (k_collapse_gather_batch<M, SK_U8, false, true>), grid, dim3(256), 0, ts, bo);
            else ISX_LAUNCH("collapse_gather", bytes, st, (k_collapse_gather_batch<M, SK_U8, false, false>), grid, dim3(256), 0, ts, bo);
        }
    }
    return ISX_OK;
}
template <int M>
int run_blend_batch(isx_blender** bs, int nb, const OutMat* outs) {
    switch (bs[0]->tiles[0].sk) {
        case SK_U8: return run_blend_batch_t<M, SK_U8>(bs, nb, outs);
        case SK_S16: return run_blend_batch_t<M, SK_S16>(bs, nb, outs);
        default: return run_blend_batch_t<M, SK_F32>(bs, nb, outs);
    }
}

int do_prepare(isx_blender* b, int x, int y, int width, int height) {
    ISX_CHECK_ARG(width > 0 && height > 0, ISX_ERR_INVALID, "prepare: empty destination ROI %d x %d", width, height);
    ISX_HIP(hipSetDevice(b->device));
    b->fw = width; b->fh = height;
    if (b->type == ISX_BLEND_NO) {      // Blender::prepare(Rect): dst_.create(size, CV_16SC3), dst_mask_.create(size, CV_8U), both setTo(0)
        ISX_CHECK_ARG((unsigned long long)width * height < (1ull << 31), ISX_ERR_UNSUPPORTED, "prepare: destination ROI %d x %d exceeds 2^31 pixels", width, height);
        const size_t n = (size_t)width * height, img_bytes = (n * 6 + 255) & ~(size_t)255;
        ISX_TRY(b->dst_arena.reserve(img_bytes + n));
        ISX_HIP(hipMemsetAsync(b->dst_arena.p, 0, img_bytes + n, b->stream));
        b->rx = x; b->ry = y; b->rw = width; b->rh = height; b->num_bands = 0;
        b->fed.clear(); b->cleared = false; b->tiles.clear(); b->ftiles.clear(); b->level0_pending = false;
        b->narrow_pending = 0; b->narrow_published = false; b->fused_cycle = false;
        b->prepared = true;
        return ISX_OK;
    }
    // num_bands_ = min(actual_num_bands_, (int)ceil(log(max_len) / log(2.0)))
    double max_len = (double)(width > height ? width : height);
    int cl = (int)std::ceil(std::log(max_len) / std::log(2.0));
    b->num_bands = b->actual_num_bands < cl ? b->actual_num_bands : cl;
    int L = b->num_bands, m = 1 << L;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    // device indexing: 32-bit record indices built from 24-bit multiplies
    ISX_CHECK_ARG(width < (1 << 24) && height < (1 << 24) && (unsigned long long)width * height < (1ull << 31), ISX_ERR_UNSUPPORTED,
                  "prepare: destination ROI %d x %d exceeds the supported 2^31 pixels / 2^24 per side", width, height);
    b->rx = x; b->ry = y; b->rw = width; b->rh = height;
    size_t total = 0;
    layout_levels(b->dst, L, height, width, b->prec, true, nullptr, &total);
    ISX_TRY(b->dst_arena.reserve(total));
    layout_levels(b->dst, L, height, width, b->prec, true, (char*)b->dst_arena.p, &total);
    // dst_.setTo(0) / the weight maps' setTo(0) are not executed: uncovered pixels are defined as zero (Cover).
    b->fed.clear();
    b->cleared = false;
    b->path_narrow = 0;
    if (b->narrow_pending && b->feed_state.p)      // an abandoned cycle: its violation words are not carried into the next one
        ISX_HIP(hipMemsetAsync(b->feed_state.p, 0, (size_t)DEF_REC_MAX * 4, b->stream));
    b->narrow_pending = 0; b->narrow_published = false; b->fused_cycle = false;
    b->tiles.clear();
    b->ftiles.clear();
    b->level0_pending = false;
    b->prepared = true;
    return ISX_OK;
}

// dst += short(img * w), dst_weight += w for one tile (eager FeatherBlender::feed, and the replay of recorded tiles)
int feather_accumulate(isx_blender* b, const isx_blender::FeatherRec& r) {
    hipStream_t st = b->stream;
    if (!b->cleared && b->fed.size() >= (size_t)MAX_COVER) {   // more tiles than a Cover holds: clear once, then plain RMW
        ISX_TRY(fill_uncovered(b, 0));
        b->cleared = true;
    }
    const Cover cov = make_cover(b, 0);
    dim3 grid(cdiv(r.cols, 64), cdiv(r.rows, 4));
    const double bytes = (double)r.rows * r.cols * ((r.sk == SK_U8 ? 3.0 : 6.0) + 4.0 + 12.0) + covered_px(b, r.dx, r.dy, r.cols, r.rows) * 12.0;
    if (r.sk == SK_U8) ISX_LAUNCH("feather_acc", bytes, st, (k_feather_acc<SK_U8>), grid, dim3(256), 0, r.img, r.istep, (const float*)r.wgt, r.wpitch, r.rows, r.cols, b->dst[0], r.dx, r.dy, cov);
    else ISX_LAUNCH("feather_acc", bytes, st, (k_feather_acc<SK_S16>), grid, dim3(256), 0, r.img, r.istep, (const float*)r.wgt, r.wpitch, r.rows, r.cols, b->dst[0], r.dx, r.dy, cov);
    if (!b->cleared) b->fed.push_back(make_int4(r.dx, r.dy, r.cols, r.rows));
    return ISX_OK;
}

// leave the deferred FeatherBlender cycle: accumulate the recorded tiles now (their buffers stay in use until blend())
int flush_feather(isx_blender* b) {
    std::vector<isx_blender::FeatherRec> recs;
    recs.swap(b->ftiles);
    for (const auto& r : recs) ISX_TRY(feather_accumulate(b, r));
    return ISX_OK;
}

// FeatherBlender::feed (createWeightMap + the weighted accumulate loop)
// deferred mode 2: a recorded tile must not reference the caller's device buffer once feed() has returned (OpenCV's
// feed() consumes its inputs - the reference releases the fed mats before blend(), W:305-308): take a private copy, one
// device-to-device pass on the blender's stream (4 B/px for a CV_8UC3 tile + its mask).  A copy kernel, not
// hipMemcpy2DAsync: the runtime's pitched D2D copy moved the 59 MB of a 4K pair at 0.46 TB/s (127 us).
// 16-byte units per thread: loads from the caller's mat at whatever alignment its rows have (a dense cv::Mat row of 3425 CV_16SC3 pixels
// starts on a 2-byte boundary; unaligned global access is legal on this part and costs nothing at a regular stride), stores to the
// private buffer aligned.  A unit that overhangs the row reads into the next row (inside the mat) and writes into the copy's own pitch;
// the last row's overhanging unit is copied byte by byte.
typedef unsigned u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned u32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));
__global__ __launch_bounds__(256) void k_copy_rows(const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep, unsigned long long row_bytes, int rows) {
    const unsigned long long off = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16ull;
    const int y0 = blockIdx.y * 8;
    if (off >= row_bytes) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int y = y0 + i;
        if (y >= rows) break;
        const unsigned char* sp = src + (size_t)y * sstep + off;
        unsigned char* dp = dst + (size_t)y * dstep + off;
        // a unit that overhangs its row reads into the next rows of the mat: only while its 16 bytes end inside the mat's last row (rows narrower
        // than a unit under a small step - a mask under 16 columns wide - could otherwise run past the caller's allocation from row rows - 2 on)
        if (off + 16ull <= row_bytes || (unsigned long long)y * sstep + off + 16ull <= (unsigned long long)(rows - 1) * sstep + row_bytes) *(u32x4_a16*)dp = *(const u32x4_a1*)sp;
        else for (unsigned b = 0; off + b < row_bytes; ++b) dp[b] = sp[b];
    }
}

int private_copy(isx_mat& d, DevBuf& buf, hipStream_t st) {
    const size_t row_bytes = (size_t)d.cols * mat_elem_size(d.type);
    const unsigned char* src = (const unsigned char*)d.data;
    const double bytes = 2.0 * (double)row_bytes * d.rows;
    // the copy's rows start on 64-byte boundaries whatever the caller's do: the level-0 kernels read a CV_16SC3 pair as one 12-byte load,
    // and against a continuous copy of 20 550-byte rows (2-byte aligned starts) the last collapse step ran 85 us instead of 58
    const size_t pitch = (row_bytes + 63) & ~(size_t)63;
    ISX_TRY(buf.reserve(pitch * (size_t)d.rows + 64));
    ISX_LAUNCH("feed_copy", bytes, st, k_copy_rows, dim3(cdiv((int)cdiv((int)row_bytes, 16), 256), cdiv(d.rows, 8)), dim3(256), 0, src, d.step, (unsigned char*)buf.p, pitch,
               (unsigned long long)row_bytes, d.rows);
    d.data = buf.p; d.step = pitch;
    return ISX_OK;
}

// isx_blender_feed_dilated (W:286-301 fused into the feed): mask = dilate(seam_mask, MORPH_RECT kw x kh) & warped_mask, written straight
// into a buffer the blender owns - `dst` - so that no mask travels through the caller and, for a recorded tile, the private copy of
// the mask (OpenCV's feed contract) IS the dilation's output: one pass over the mask less.  On return st_mask.d describes that buffer.
struct DilateSpec { const isx_mat* other; int kw, kh; };
int stage_dilated_mask(isx_blender* b, MatStage& st_mask, const isx_mat* seam, const DilateSpec& ds, DevBuf& dst, hipStream_t st) {
    ISX_TRY(check_mat(ds.other, "feed_dilated: warped mask"));
    ISX_CHECK_ARG(ds.other->type == ISX_8UC1 && ds.other->rows == seam->rows && ds.other->cols == seam->cols, ISX_ERR_SIZE,
                  "feed_dilated: the warped mask must be a CV_8U mask of the seam mask's size");
    ISX_TRY(st_mask.use_in(seam, st, "feed_dilated: seam mask"));
    ISX_TRY(b->st_other.use_in(ds.other, st, "feed_dilated: warped mask"));
    const size_t pitch = ((size_t)seam->cols + 63) & ~(size_t)63;
    ISX_TRY(dst.reserve(pitch * (size_t)seam->rows + 64));
    ISX_TRY(dilate_and_device((const unsigned char*)st_mask.d.data, st_mask.d.step, (const unsigned char*)b->st_other.d.data, b->st_other.d.step, seam->rows, seam->cols,
                              ds.kw, ds.kh, (unsigned char*)dst.p, pitch, st));
    st_mask.d.data = dst.p; st_mask.d.step = pitch; st_mask.d.device = b->device;
    return ISX_OK;
}

int do_feed_feather(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y, bool u8_entry, const DilateSpec* dil = nullptr) {
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "feed: prepare() has not been called (or blend() already released the accumulators)");
    ISX_TRY(check_mat(img, "feed: img"));
    ISX_TRY(check_mat(mask, "feed: mask"));
    if (u8_entry) ISX_CHECK_ARG(img->type == ISX_8UC3, ISX_ERR_TYPE, "feed_u8: img must be CV_8UC3, got %s", type_name(img->type));
    else ISX_CHECK_ARG(img->type == ISX_16SC3, ISX_ERR_TYPE, "FeatherBlender::feed: img must be CV_16SC3, got %s", type_name(img->type));
    ISX_CHECK_ARG(mask->type == ISX_8UC1, ISX_ERR_TYPE, "feed: mask must be CV_8U, got %s", type_name(mask->type));
    ISX_CHECK_ARG(mask->rows == img->rows && mask->cols == img->cols, ISX_ERR_SIZE, "feed: mask %dx%d does not match img %dx%d",
                  mask->cols, mask->rows, img->cols, img->rows);
    const int dx = tl_x - b->rx, dy = tl_y - b->ry;
    ISX_CHECK_ARG(dx >= 0 && dy >= 0 && dx + img->cols <= b->rw && dy + img->rows <= b->rh, ISX_ERR_INVALID,
                  "feed: tile at (%d,%d) %dx%d lies outside the prepared ROI", tl_x, tl_y, img->cols, img->rows);
    ISX_HIP(hipSetDevice(b->device));
    hipStream_t st = b->stream;
    // deferred cycle (the fed mats outlive blend()): only the weight map is built here, into a buffer of the tile's own
    const int sk = u8_entry ? SK_U8 : SK_S16;
    const bool can_defer = b->deferred && !b->cleared && (b->ftiles.empty() ? b->fed.empty() : true) && b->ftiles.size() < (size_t)DEF_MAX &&
                           (b->ftiles.empty() || b->ftiles[0].sk == sk);
    if (!can_defer) ISX_TRY(flush_feather(b));
    const size_t slot = b->ftiles.size();
    if (can_defer && b->tile_arenas.size() <= slot) {
        b->tile_arenas.emplace_back(new DevBuf());
        b->tile_img.emplace_back(new MatStage());
        b->tile_mask.emplace_back(new MatStage());
    }
    MatStage& st_img = can_defer ? *b->tile_img[slot] : b->st_img;
    DevBuf& wbuf = can_defer ? *b->tile_arenas[slot] : b->feather_w;
    ISX_TRY(st_img.use_in(img, st, "feed: img"));
    if (dil) ISX_TRY(stage_dilated_mask(b, b->st_mask, mask, *dil, b->dil_mask, st));
    else ISX_TRY(b->st_mask.use_in(mask, st, "feed: mask"));
    if (can_defer && b->deferred_copy && img->device >= 0) {   // the mask is consumed here (weight map); the image is read by blend()
        if (b->tile_copy_img.size() <= slot) { b->tile_copy_img.emplace_back(new DevBuf()); b->tile_copy_mask.emplace_back(new DevBuf()); }
        ISX_TRY(private_copy(st_img.d, *b->tile_copy_img[slot], st));
    }
    const int rows = img->rows, cols = img->cols;
    const int nseg = cdiv(rows, DT_SEG);
    const int pitch = (cols + 3) & ~3;   // int4 stores of the row pass
    const size_t map_bytes = ((size_t)rows * pitch * 4 + 255) & ~(size_t)255;
    ISX_TRY(wbuf.reserve(map_bytes + (size_t)nseg * cols * 8));
    int* rowd = (int*)wbuf.p;
    int* seg_f = (int*)((char*)wbuf.p + map_bytes);
    int* seg_b = seg_f + (size_t)nseg * cols;
    ISX_LAUNCH("dt_rows", (double)rows * cols * 5.0, st, k_dt_rows, dim3(rows), dim3(256), 0, (const unsigned char*)b->st_mask.d.data, b->st_mask.d.step, rows, cols, rowd, pitch);
    ISX_LAUNCH("dt_seg_min", (double)rows * cols * 4.0, st, k_dt_seg_min, dim3(cdiv(cols, 64), nseg), dim3(64), 0, (const int*)rowd, pitch, rows, cols, seg_f, seg_b);
    ISX_LAUNCH("dt_cols_weight", (double)rows * cols * 8.0, st, k_dt_cols_weight, dim3(cdiv(cols, 64), nseg), dim3(64), 0, rowd, pitch, rows, cols, (const int*)seg_f,
               (const int*)seg_b, nseg, b->sharpness);
    isx_blender::FeatherRec r;
    r.img = (const unsigned char*)st_img.d.data; r.istep = st_img.d.step; r.sk = sk; r.wgt = (float*)rowd; r.wpitch = pitch;
    r.dx = dx; r.dy = dy; r.rows = rows; r.cols = cols;
    if (can_defer) { b->ftiles.push_back(r); return ISX_OK; }
    return feather_accumulate(b, r);
}

// Blender::feed of the base class (Blender::NO): a masked copy into dst_, dst_mask_ |= mask
int do_feed_no(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y, bool u8_entry, const DilateSpec* dil = nullptr) {
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "feed: prepare() has not been called (or blend() already released dst_)");
    ISX_TRY(check_mat(img, "feed: img"));
    ISX_TRY(check_mat(mask, "feed: mask"));
    if (u8_entry) ISX_CHECK_ARG(img->type == ISX_8UC3, ISX_ERR_TYPE, "feed_u8: img must be CV_8UC3, got %s", type_name(img->type));
    else ISX_CHECK_ARG(img->type == ISX_16SC3, ISX_ERR_TYPE, "Blender::feed: img must be CV_16SC3, got %s", type_name(img->type));
    ISX_CHECK_ARG(mask->type == ISX_8UC1, ISX_ERR_TYPE, "feed: mask must be CV_8U, got %s", type_name(mask->type));
    ISX_CHECK_ARG(mask->rows == img->rows && mask->cols == img->cols, ISX_ERR_SIZE, "feed: mask %dx%d does not match img %dx%d",
                  mask->cols, mask->rows, img->cols, img->rows);
    const int dx = tl_x - b->rx, dy = tl_y - b->ry;
    ISX_CHECK_ARG(dx >= 0 && dy >= 0 && dx + img->cols <= b->rw && dy + img->rows <= b->rh, ISX_ERR_INVALID,
                  "feed: tile at (%d,%d) %dx%d lies outside the prepared ROI", tl_x, tl_y, img->cols, img->rows);
    ISX_HIP(hipSetDevice(b->device));
    hipStream_t st = b->stream;
    ISX_TRY(b->st_img.use_in(img, st, "feed: img"));
    if (dil) ISX_TRY(stage_dilated_mask(b, b->st_mask, mask, *dil, b->dil_mask, st));
    else ISX_TRY(b->st_mask.use_in(mask, st, "feed: mask"));
    const size_t n = (size_t)b->rw * b->rh, img_bytes = (n * 6 + 255) & ~(size_t)255;
    short* dst = (short*)b->dst_arena.p;
    unsigned char* dmask = (unsigned char*)b->dst_arena.p + img_bytes;
    const dim3 grid(cdiv(img->cols, 64), cdiv(img->rows, 4));
    const double bytes = (double)img->rows * img->cols * ((u8_entry ? 3.0 : 6.0) + 1.0 + 7.0);
    if (u8_entry) ISX_LAUNCH("no_feed", bytes, st, (k_no_feed<SK_U8>), grid, dim3(256), 0, (const unsigned char*)b->st_img.d.data, b->st_img.d.step,
                             (const unsigned char*)b->st_mask.d.data, b->st_mask.d.step, img->rows, img->cols, dst, dmask, b->rw, dx, dy);
    else ISX_LAUNCH("no_feed", bytes, st, (k_no_feed<SK_S16>), grid, dim3(256), 0, (const unsigned char*)b->st_img.d.data, b->st_img.d.step,
                    (const unsigned char*)b->st_mask.d.data, b->st_mask.d.step, img->rows, img->cols, dst, dmask, b->rw, dx, dy);
    return ISX_OK;      // (the shared staging buffers are reused by the next feed on the same stream: ordered)
}

int do_feed(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y, bool u8_entry, const DilateSpec* dil = nullptr) {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "feed: null blender");
    if (b->type == ISX_BLEND_FEATHER) return do_feed_feather(b, img, mask, tl_x, tl_y, u8_entry, dil);
    if (b->type == ISX_BLEND_NO) return do_feed_no(b, img, mask, tl_x, tl_y, u8_entry, dil);
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "feed: null blender");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "feed: prepare() has not been called (or blend() already released the pyramids)");
    ISX_TRY(check_mat(img, "feed: img"));
    ISX_TRY(check_mat(mask, "feed: mask"));
    if (u8_entry) ISX_CHECK_ARG(img->type == ISX_8UC3, ISX_ERR_TYPE, "feed_u8: img must be CV_8UC3, got %s", type_name(img->type));
    else {
        // CV_8UC3 selects createLaplacePyr's 8-bit branch in OpenCV 3.4.2 (blenders.cpp): pyrDown / pyrUp on CV_8U levels, subtract(...,
        // CV_16S), the top level converted to CV_16S.  With the same work type (int), the same casts ((v + 128) >> 8, (v + 32) >> 6) and every
        // intermediate inside [0, 255] (kernel weights sum to 256 / 64 over bytes), saturate_cast<uchar> never acts and that branch produces
        // exactly the CV_16S branch's numbers for the converted image: it is the fused-conversion path of isx_blender_feed_u8.
        ISX_CHECK_ARG(img->type == ISX_8UC3 || img->type == ISX_16SC3 || (img->type == ISX_32FC3 && b->prec != ISX_PREC_I16), ISX_ERR_TYPE,
                      "feed: img must be CV_16SC3 or CV_8UC3%s, got %s", b->prec != ISX_PREC_I16 ? " or CV_32FC3" : "", type_name(img->type));
    }
    ISX_CHECK_ARG(mask->type == ISX_8UC1, ISX_ERR_TYPE, "feed: mask must be CV_8U, got %s", type_name(mask->type));
    ISX_CHECK_ARG(mask->rows == img->rows && mask->cols == img->cols, ISX_ERR_SIZE, "feed: mask %dx%d does not match img %dx%d",
                  mask->cols, mask->rows, img->cols, img->rows);
    ISX_HIP(hipSetDevice(b->device));
    // deferred level 0: possible while every feed of this cycle has been deferred, at most DEF_MAX tiles,
    // one source type.  A recorded tile's staging buffers and pyramid arena are its own (they must
    // outlive this call); an eager feed uses the shared ones.
    const int sk = src_kind_of(img->type);
    const int L0 = b->num_bands;
    // Fused feed (round 5, k_feed_pd0): in mode 2 a CV_8UC3 / CV_16SC3 device tile is read ONCE - level 1 of its pyramid and the private copy
    // come out of the same pass (ISX_FEED_FUSE=0: private_copy + the level-0 pyrDown inside blend(), as in round 4) - and the copy of a
    // CV_16SC3 tile is written as CV_8UC3 (ISX_FEED_NARROW=0: as CV_16SC3), see pyrdown_l0.inc.  A cycle narrows all of its tiles or none.
    static const bool fuse_on = [] { const char* e = getenv("ISX_FEED_FUSE"); return !(e && e[0] == '0'); }();
    static const bool narrow_on = [] { const char* e = getenv("ISX_FEED_NARROW"); return !(e && e[0] == '0'); }();
    const bool fusable = b->deferred && b->deferred_copy && fuse_on && !dil && !b->overlap && img->device >= 0 && mask->device >= 0 && (sk == SK_U8 || sk == SK_S16) &&
                         (unsigned long long)img->step * img->rows < (1ull << 31) && (unsigned long long)mask->step * img->rows < (1ull << 31) &&
                         img->step < (1u << 24) && mask->step < (1u << 24) && img->cols >= 2 && img->rows >= 2;
    const bool cycle_narrow = !b->tiles.empty() && b->tiles[0].narrow != 0;
    const bool can_defer = b->deferred && L0 >= 1 && L0 <= ACC_MAXL && !b->cleared &&
                           (b->tiles.empty() ? b->fed.empty() : b->level0_pending) && b->tiles.size() < (size_t)DEF_REC_MAX &&
                           (b->tiles.empty() || b->tiles[0].fed_sk == sk) && (!cycle_narrow || (fusable && sk == SK_S16));
    if (!can_defer) ISX_TRY(flush_deferred(b));
    const bool fused = can_defer && fusable;
    bool narrow = false;
    if (fused && sk == SK_S16) {
        if (!b->tiles.empty()) narrow = cycle_narrow;
        else if (narrow_on && !b->narrow_off) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;      // blend() reads the violation word on the host: not while the stream is being captured
            narrow = hipStreamIsCapturing(b->stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
        }
    }
    const size_t slot = b->tiles.size();
    if (can_defer && b->tile_arenas.size() <= slot) {
        b->tile_arenas.emplace_back(new DevBuf());
        b->tile_img.emplace_back(new MatStage());
        b->tile_mask.emplace_back(new MatStage());
    }
    MatStage& st_img = can_defer ? *b->tile_img[slot] : b->st_img;
    MatStage& st_mask = can_defer ? *b->tile_mask[slot] : b->st_mask;
    DevBuf& arena = can_defer ? *b->tile_arenas[slot] : b->tile_arena;
    ISX_TRY(st_img.use_in(img, b->stream, "feed: img"));
    if (can_defer && b->tile_copy_img.size() <= slot) {
        b->tile_copy_img.emplace_back(new DevBuf()); b->tile_copy_mask.emplace_back(new DevBuf());
        b->tile_copy_wide.emplace_back(new DevBuf()); b->tile_chunk.emplace_back(new DevBuf());
    }
    if (dil) ISX_TRY(stage_dilated_mask(b, st_mask, mask, *dil, can_defer ? *b->tile_copy_mask[slot] : b->dil_mask, b->stream));   // the blender's own copy already
    else ISX_TRY(st_mask.use_in(mask, b->stream, "feed: mask"));
    if (can_defer && b->deferred_copy && !fused) {
        if (img->device >= 0) ISX_TRY(private_copy(st_img.d, *b->tile_copy_img[slot], b->stream));
        if (mask->device >= 0 && !dil) ISX_TRY(private_copy(st_mask.d, *b->tile_copy_mask[slot], b->stream));
    }
    const isx_mat& di = st_img.d;
    const isx_mat& dm = st_mask.d;

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

    // the tile as level 0 of its pyramid, read from (data, step) of an image of `type` and a mask
    auto src0_of = [&](const void* idata, size_t istep, int type, const void* mdata, size_t mstep) {
        Src0 q;
        q.img = (const unsigned char*)idata; q.img_step = istep;
        q.mask = (const unsigned char*)mdata; q.mask_step = mstep;
        q.rows = img->rows; q.cols = img->cols;
        q.top = tl_y - tlny; q.left = tl_x - tlnx;
        q.height = height; q.width = width;
        q.imis = (unsigned)((uintptr_t)q.img & 3); q.mmis = (unsigned)((uintptr_t)q.mask & 3);
        q.img_al = q.img - q.imis; q.mask_al = q.mask - q.mmis;
        q.iend = 0; q.mend = 0;
        if ((type == ISX_8UC3 || type == ISX_16SC3) && (unsigned long long)istep * img->rows < (1ull << 31) &&
            (unsigned long long)mstep * img->rows < (1ull << 31) && istep < (1u << 24) && mstep < (1u << 24)) {
            q.iend = (unsigned)((size_t)(img->rows - 1) * istep + (size_t)img->cols * (type == ISX_8UC3 ? 3 : 6)) + q.imis;
            q.mend = (unsigned)((size_t)(img->rows - 1) * mstep + (size_t)img->cols) + q.mmis;
        }
        return q;
    };
    Src0 s0 = src0_of(di.data, di.step, img->type, dm.data, dm.step);

    LevelBuf g[MAX_LEVELS];
    size_t total = 0;
    // tile Gaussian levels 1..L live in the tile arena; level 0 is the caller's tile
    layout_levels(g, L, height, width, b->prec, false, nullptr, &total);
    size_t skip = 0;
    {   // do not allocate level 0
        size_t n0 = (size_t)height * width;
        skip = (n0 * g_px_bytes(b->prec) + 255) & ~(size_t)255;
    }
    ISX_TRY(arena.reserve(total - skip + 256));
    layout_levels(g, L, height, width, b->prec, false, (char*)arena.p - skip, &total);
    g[0].img = nullptr; g[0].wgt = nullptr;

    // the fused pass: G_1 and the private copies out of one read of the caller's tile
    isx_blender::TileRec rec;
    rec.fed_sk = sk; rec.sk = sk;
    if (fused) {
        const size_t ipx = narrow ? 3 : (sk == SK_U8 ? 3 : 6);
        const size_t ipitch = ((size_t)img->cols * ipx + 63) & ~(size_t)63, mpitch = ((size_t)img->cols + 63) & ~(size_t)63;
        ISX_TRY(b->tile_copy_img[slot]->reserve(ipitch * (size_t)img->rows + 64));
        ISX_TRY(b->tile_copy_mask[slot]->reserve(mpitch * (size_t)img->rows + 64));
        FeedCopy fc;
        memset(&fc, 0, sizeof(fc));
        fc.cimg = (unsigned char*)b->tile_copy_img[slot]->p; fc.cstep = (unsigned)ipitch;
        fc.cmask = (unsigned char*)b->tile_copy_mask[slot]->p; fc.cmstep = (unsigned)mpitch;
        const int nbx = cdiv(g[1].cols, PD_OW);
        if (narrow) {
            const size_t wpitch = ((size_t)img->cols * 6 + 63) & ~(size_t)63;
            ISX_TRY(b->tile_copy_wide[slot]->reserve(wpitch * (size_t)img->rows + 64));
            ISX_TRY(b->tile_chunk[slot]->reserve((size_t)nbx * (size_t)img->rows + 64));
            if (!b->feed_state.p) {
                ISX_TRY(b->feed_state.reserve((size_t)DEF_REC_MAX * 4));
                ISX_HIP(hipMemsetAsync(b->feed_state.p, 0, (size_t)DEF_REC_MAX * 4, b->stream));
            }
            if (!b->feed_pin) {
                ISX_HIP(hipHostMalloc((void**)&b->feed_pin, 64, hipHostMallocCoherent | hipHostMallocMapped));
                b->feed_pin[0] = 0; b->feed_pin[1] = 0;
            }
            fc.wimg = (unsigned char*)b->tile_copy_wide[slot]->p; fc.wstep = (unsigned)wpitch;
            fc.chunk = (unsigned char*)b->tile_chunk[slot]->p; fc.nbx = nbx;
            fc.state = (unsigned*)b->feed_state.p + slot;
            rec.narrow = 1; rec.wide = fc.wimg; rec.wide_step = wpitch; rec.chunk = fc.chunk; rec.nbx = nbx;
        }
        // level 1 PLANAR when the chain will most likely want it so (run_blend_deferred_t: the last step as k_collapse_roll with level 1 produced
        // by k_collapse_gather); blend() produces the level again from the private copy in the rare cycle that wants the other layout
        static const bool top_on0 = [] { const char* e = getenv("ISX_TOP"); return !(e && e[0] == '0'); }();
        static const bool g1p_on = [] { const char* e = getenv("ISX_G1P"); return !(e && e[0] == '0'); }();
        static const bool roll_on = [] { const char* e = getenv("ISX_ROLL"); return !(e && atoi(e) == 0); }();
        const int D0 = std::min(TOP_DMAX, L - 1);
        const bool planar = L >= 2 && (!(top_on0 && D0 >= 2) || L - D0 >= 2) && g1p_on && roll_on && (b->prec == M_F32 || b->prec == M_I16);
        // ... in Q8 records (load_px_planar) where the tile is bytes: CV_8UC3, or CV_16SC3 being narrowed (a violated cycle is widened and produces
        // its level 1 again)
        static const bool q8_on = [] { const char* e = getenv("ISX_G1Q8"); return !(e && e[0] == '0'); }();
        const bool q8 = planar && q8_on && b->prec == M_F32 && (sk == SK_U8 || narrow);
        LevelBuf g1 = g[1];
        if (planar) g1.wgt = (float*)((char*)g1.img + planar_wgt_offset(b->prec, g1.rows, g1.cols, q8));
        const double bytes = (double)height * width * (src_px_bytes(sk) + 1.0) + (double)g[1].rows * g[1].cols * (q8 ? 10.0 : alg_g(b->prec)) + (double)img->rows * img->cols * ((double)ipx + 1.0);
        ISX_TRY(launch_feed_pd0(b->prec, sk, planar ? (q8 ? 2 : 1) : 0, narrow, s0, g1, fc, dim3(nbx, cdiv(g[1].rows, PD_TY)), bytes, b->stream));
        rec.g1 = planar ? (q8 ? 4 : 2) : 1;
        rec.sk = narrow ? SK_U8 : sk;
        s0 = src0_of(fc.cimg, ipitch, narrow ? ISX_8UC3 : img->type, fc.cmask, mpitch);
        if (narrow) { ++b->narrow_pending; b->narrow_published = false; }
        b->fused_cycle = true;
    }

    int x_tl = tlnx - b->rx, y_tl = tlny - b->ry;
    if (can_defer) {   // record only: blend() does all the work
        isx_blender::TileRec r = rec;
        r.s0 = s0; r.x_tl = x_tl; r.y_tl = y_tl; r.width = width; r.height = height;
        for (int k = 0; k <= L; ++k) r.g[k] = g[k];
        r.g[0].rows = height; r.g[0].cols = width;
        b->tiles.push_back(r);
        b->level0_pending = true;
        if (b->chain_on_side.size() <= slot) b->chain_on_side.resize(slot + 1, 0);
        b->chain_on_side[slot] = 0;
        if (b->overlap) {
            if (b->side.size() <= slot) {
                hipStream_t sst; hipEvent_t e1, e2;
                ISX_HIP(hipStreamCreateWithFlags(&sst, hipStreamNonBlocking));
                ISX_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
                ISX_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
                b->side.push_back(sst); b->ev_ready.push_back(e1); b->ev_done.push_back(e2);
            }
            ISX_HIP(hipEventRecord(b->ev_ready[slot], b->stream));            // the tile is complete on the main stream here
            ISX_HIP(hipStreamWaitEvent(b->side[slot], b->ev_ready[slot], 0));
            ISX_TRY(launch_down_chain(b, b->tiles[slot], b->side[slot]));
            ISX_HIP(hipEventRecord(b->ev_done[slot], b->side[slot]));
            b->chain_on_side[slot] = 1;
        }
        return ISX_OK;
    }
    if (!b->cleared && b->fed.size() >= (size_t)MAX_COVER) {   // more tiles than a Cover holds: clear once, then plain RMW
        for (int k = 0; k <= L; ++k) ISX_TRY(fill_uncovered(b, k));
        b->cleared = true;
    }
    int rc;
    switch (b->prec) {
        case M_I16: rc = run_feed_kind<M_I16>(b, sk, s0, g, L, x_tl, y_tl, false); break;
        case M_F32: rc = run_feed_kind<M_F32>(b, sk, s0, g, L, x_tl, y_tl, false); break;
        default: rc = run_feed_kind<M_F16>(b, sk, s0, g, L, x_tl, y_tl, false); break;
    }
    ISX_TRY(rc);
    if (!b->cleared) b->fed.push_back(make_int4(x_tl, y_tl, width, height));
    return ISX_OK;
}

}  // namespace

extern "C" {

int isx_blender_create(int type, int num_bands, int precision, int device, isx_blender** out) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(out != nullptr, ISX_ERR_INVALID, "isx_blender_create: null out pointer");
    *out = nullptr;
    ISX_CHECK_ARG(type == ISX_BLEND_MULTI_BAND || type == ISX_BLEND_FEATHER || type == ISX_BLEND_NO, ISX_ERR_INVALID,
                  "isx_blender_create: Blender::NO (0), Blender::FEATHER (1) or Blender::MULTI_BAND (2), got %d", type);
    if (type != ISX_BLEND_MULTI_BAND) { num_bands = 0; precision = ISX_PREC_I16; }   // CV_16SC3 accumulator (+ CV_32F weights: FEATHER), one level
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
} ISX_EXIT("isx_blender_create")

int isx_blender_destroy(isx_blender* b) ISX_ENTRY {
    if (!b) return ISX_OK;
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->stream);
    for (size_t i = 0; i < b->side.size(); ++i) {
        (void)hipStreamSynchronize(b->side[i]); (void)hipStreamDestroy(b->side[i]);
        (void)hipEventDestroy(b->ev_ready[i]); (void)hipEventDestroy(b->ev_done[i]);
    }
    if (b->feed_pin) (void)hipHostFree(b->feed_pin);
    delete b;
    return ISX_OK;
} ISX_EXIT("isx_blender_destroy")

int isx_blender_set_stream(isx_blender* b, void* hip_stream) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_stream: null blender");
    b->stream = (hipStream_t)hip_stream;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_stream")

int isx_blender_set_num_bands(isx_blender* b, int num_bands) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_num_bands: null blender");
    ISX_CHECK_ARG(num_bands >= 0 && num_bands < MAX_LEVELS - 1, ISX_ERR_INVALID, "setNumBands(%d) out of range", num_bands);
    b->actual_num_bands = num_bands;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_num_bands")

int isx_blender_set_deferred_level0(isx_blender* b, int on) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_deferred_level0: null blender");
    ISX_CHECK_ARG(!b->prepared || b->fed.empty(), ISX_ERR_STATE, "isx_blender_set_deferred_level0: tiles have already been fed in this cycle");
    b->deferred = on != 0;
    b->deferred_copy = on == 2;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_deferred_level0")

int isx_blender_set_window(isx_blender* b, int x0, int x1) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "set_window: null blender");
    ISX_CHECK_ARG((x0 == 0 && x1 == 0) || (x0 >= 0 && x1 > x0 && x0 % ISX_WINDOW_GRANULE == 0), ISX_ERR_INVALID,
                  "set_window: columns [%d, %d): the first must be a non-negative multiple of %d and below the second (0, 0 = no window)", x0, x1, ISX_WINDOW_GRANULE);
    b->win_x0 = x0; b->win_x1 = x1;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_window")

int isx_blender_set_mark_event(isx_blender* b, void* hip_event, int after_level) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_mark_event: null blender");
    b->mark_event = (hipEvent_t)hip_event;
    b->mark_level = after_level;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_mark_event")

#ifdef ISX_PHASE_TIMING
// instrumented builds only (not declared in the header): sums of the 11 phases + the wave count; reset != 0 clears them
int isx_debug_phase(unsigned long long* out12, int reset) ISX_ENTRY {
    static unsigned long long h[1024][12];
    ISX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)));
    for (int k = 0; k < 12; ++k) { out12[k] = 0; for (int i = 0; i < 1024; ++i) out12[k] += h[i][k]; }
    if (reset) { memset(h, 0, sizeof(h)); ISX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof(h))); }
    return ISX_OK;
} ISX_EXIT("isx_debug_phase")
#endif

int isx_blender_set_sharpness(isx_blender* b, float sharpness) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_sharpness: null blender");
    ISX_CHECK_ARG(b->type == ISX_BLEND_FEATHER, ISX_ERR_STATE, "setSharpness: not a FeatherBlender");
    ISX_CHECK_ARG(sharpness == sharpness, ISX_ERR_INVALID, "setSharpness: NaN");
    b->sharpness = sharpness;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_sharpness")

int isx_blender_set_overlap(isx_blender* b, int on) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_overlap: null blender");
    b->overlap = on != 0;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_overlap")

int isx_blender_num_bands(isx_blender* b, int* num_bands) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr && num_bands != nullptr, ISX_ERR_INVALID, "isx_blender_num_bands: null argument");
    *num_bands = b->prepared ? b->num_bands : b->actual_num_bands;
    return ISX_OK;
} ISX_EXIT("isx_blender_num_bands")

int isx_blender_prepare(isx_blender* b, int n, const int* c, const int* s) ISX_ENTRY {
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
} ISX_EXIT("isx_blender_prepare")

int isx_blender_prepare_roi(isx_blender* b, int x, int y, int width, int height) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "prepare: null blender");
    return do_prepare(b, x, y, width, height);
} ISX_EXIT("isx_blender_prepare_roi")

int isx_blender_feed(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y) ISX_ENTRY {
    clear_error();
    return do_feed(b, img, mask, tl_x, tl_y, false);
} ISX_EXIT("isx_blender_feed")

int isx_blender_feed_u8(isx_blender* b, const isx_mat* img, const isx_mat* mask, int tl_x, int tl_y) ISX_ENTRY {
    clear_error();
    return do_feed(b, img, mask, tl_x, tl_y, true);
} ISX_EXIT("isx_blender_feed_u8")

int isx_blender_feed_dilated(isx_blender* b, const isx_mat* img, const isx_mat* seam_mask, const isx_mat* warped_mask, int kw, int kh, int tl_x, int tl_y) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(b != nullptr && img != nullptr && seam_mask != nullptr && warped_mask != nullptr, ISX_ERR_INVALID, "feed_dilated: null argument");
    ISX_TRY(check_mat(seam_mask, "feed_dilated: seam mask"));
    ISX_CHECK_ARG(seam_mask->type == ISX_8UC1, ISX_ERR_TYPE, "feed_dilated: seam mask must be CV_8U, got %s", type_name(seam_mask->type));
    ISX_CHECK_ARG(kw >= 1 && kh >= 1, ISX_ERR_INVALID, "feed_dilated: bad structuring element %d x %d", kw, kh);
    const DilateSpec ds{warped_mask, kw, kh};
    return do_feed(b, img, seam_mask, tl_x, tl_y, img->type == ISX_8UC3, &ds);
} ISX_EXIT("isx_blender_feed_dilated")

int isx_blender_last_path(isx_blender* b, int* cycle, int* last_step) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_last_path: null blender");
    if (cycle) *cycle = b->type == ISX_BLEND_MULTI_BAND ? b->path_cycle : 0;
    if (last_step) *last_step = b->type == ISX_BLEND_MULTI_BAND ? b->path_last : 0;
    return ISX_OK;
} ISX_EXIT("isx_blender_last_path")

int isx_blender_level1_format(isx_blender* b, int* format) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr && format != nullptr, ISX_ERR_INVALID, "isx_blender_level1_format: null argument");
    *format = (b->type == ISX_BLEND_MULTI_BAND && b->path_cycle != 0) ? b->path_g1 : 0;
    return ISX_OK;
} ISX_EXIT("isx_blender_level1_format")

int isx_blender_table_uploads(isx_blender* b, long long* pieces) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr && pieces != nullptr, ISX_ERR_INVALID, "isx_blender_table_uploads: null argument");
    *pieces = b->tab.uploads;
    return ISX_OK;
} ISX_EXIT("isx_blender_table_uploads")

int isx_blender_set_narrow_copies(isx_blender* b, int on) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_set_narrow_copies: null blender");
    ISX_CHECK_ARG(b->tiles.empty() || b->tiles[0].narrow == 0 || on != 0, ISX_ERR_STATE, "isx_blender_set_narrow_copies: a cycle with narrowed tiles is open (call it before the first feed)");
    b->narrow_off = on == 0;
    return ISX_OK;
} ISX_EXIT("isx_blender_set_narrow_copies")

int isx_blender_feed_path(isx_blender* b, int* fused_tiles, int* narrowed) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "isx_blender_feed_path: null blender");
    if (fused_tiles) *fused_tiles = b->type == ISX_BLEND_MULTI_BAND ? b->path_fused : 0;
    if (narrowed) *narrowed = b->type == ISX_BLEND_MULTI_BAND ? b->path_narrow : 0;
    return ISX_OK;
} ISX_EXIT("isx_blender_feed_path")

int isx_blender_result_size(isx_blender* b, int* width, int* height) ISX_ENTRY {
    ISX_CHECK_ARG(b != nullptr && width != nullptr && height != nullptr, ISX_ERR_INVALID, "result_size: null argument");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "result_size: prepare() has not been called");
    *width = b->fw; *height = b->fh;
    return ISX_OK;
} ISX_EXIT("isx_blender_result_size")

int isx_blender_debug_level(isx_blender* b, int level, void* lap, float* weight, int* rows, int* cols) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(b != nullptr && rows != nullptr && cols != nullptr, ISX_ERR_INVALID, "debug_level: null argument");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "debug_level: prepare() has not been called");
    ISX_CHECK_ARG(level >= 0 && level <= b->num_bands, ISX_ERR_INVALID, "debug_level: level %d of %d", level, b->num_bands);
    ISX_CHECK_ARG(b->type != ISX_BLEND_NO, ISX_ERR_UNSUPPORTED, "debug_level: Blender::NO keeps no pyramid");
    ISX_HIP(hipSetDevice(b->device));
    const LevelBuf& d = b->dst[level];
    *rows = d.rows; *cols = d.cols;
    size_t n = (size_t)d.rows * d.cols;
    ISX_TRY(flush_deferred(b));                           // deferred tiles: replay them through the eager feed now
    ISX_TRY(flush_feather(b));
    if (!b->cleared) ISX_TRY(fill_uncovered(b, level));   // materialise the "uncovered == 0" definition
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
} ISX_EXIT("isx_blender_debug_level")

// the checks, staging and OutMat of Blender::blend for one blender (shared by isx_blender_blend and isx_blender_blend_batch)
static int blend_begin(isx_blender* b, isx_mat* dst, isx_mat* dst_mask, OutMat* po) {
    ISX_CHECK_ARG(b != nullptr, ISX_ERR_INVALID, "blend: null blender");
    ISX_CHECK_ARG(b->prepared, ISX_ERR_STATE, "blend: prepare() has not been called (or blend() already released the pyramids)");
    ISX_TRY(check_mat(dst, "blend: dst"));
    ISX_CHECK_ARG(dst->type == ISX_16SC3 || dst->type == ISX_8UC3 || (dst->type == ISX_32FC3 && b->prec != ISX_PREC_I16), ISX_ERR_TYPE,
                  "blend: dst must be CV_16SC3, CV_8UC3%s, got %s", b->prec != ISX_PREC_I16 ? " or CV_32FC3" : "", type_name(dst->type));
    ISX_CHECK_ARG(!(b->type == ISX_BLEND_NO && dst->type == ISX_32FC3), ISX_ERR_TYPE, "blend: Blender::NO hands out its CV_16SC3 canvas (or CV_8UC3), not CV_32FC3");
    const bool windowed = b->win_x1 > b->win_x0;
    const int out_cols = windowed ? b->win_x1 - b->win_x0 : b->fw;   // a window's mats hold its columns only
    if (windowed) {
        ISX_CHECK_ARG(b->type == ISX_BLEND_FEATHER ? !b->ftiles.empty() : b->level0_pending, ISX_ERR_UNSUPPORTED,
                      "blend: a column window needs the deferred cycle (isx_blender_set_deferred_level0) with every tile still recorded");
        ISX_CHECK_ARG(b->win_x0 < b->fw, ISX_ERR_SIZE, "blend: the window starts at column %d, the result is %d wide", b->win_x0, b->fw);

    }
    ISX_CHECK_ARG(dst->rows == b->fh && dst->cols == out_cols, ISX_ERR_SIZE, "blend: dst is %dx%d, result%s is %dx%d", dst->cols, dst->rows,
                  windowed ? " window" : "", out_cols, b->fh);
    if (dst_mask) {
        ISX_TRY(check_mat(dst_mask, "blend: dst_mask"));
        ISX_CHECK_ARG(dst_mask->type == ISX_8UC1, ISX_ERR_TYPE, "blend: dst_mask must be CV_8U, got %s", type_name(dst_mask->type));
        ISX_CHECK_ARG(dst_mask->rows == b->fh && dst_mask->cols == out_cols, ISX_ERR_SIZE, "blend: dst_mask is %dx%d, result%s is %dx%d",
                      dst_mask->cols, dst_mask->rows, windowed ? " window" : "", out_cols, b->fh);
    }
    ISX_HIP(hipSetDevice(b->device));
    ISX_TRY(b->st_out.use_out(dst, b->stream, "blend: dst"));
    ISX_CHECK_ARG(b->st_out.d.step < (1u << 24) && (unsigned long long)b->st_out.d.step * dst->rows < (1ull << 32), ISX_ERR_UNSUPPORTED,
                  "blend: dst of %zu bytes per row x %d rows exceeds the supported 4 GiB / 16 MiB per row", b->st_out.d.step, dst->rows);
    if (dst_mask) {
        ISX_TRY(b->st_outmask.use_out(dst_mask, b->stream, "blend: dst_mask"));
        ISX_CHECK_ARG(b->st_outmask.d.step < (1u << 24), ISX_ERR_UNSUPPORTED, "blend: dst_mask row pitch %zu exceeds 16 MiB", b->st_outmask.d.step);
    }
    OutMat& o = *po;
    o.img = (unsigned char*)b->st_out.d.data; o.img_step = b->st_out.d.step; o.img_f32 = dst->type == ISX_32FC3 ? 1 : (dst->type == ISX_8UC3 ? 2 : 0);
    o.mask = dst_mask ? (unsigned char*)b->st_outmask.d.data : nullptr;
    o.mask_step = dst_mask ? b->st_outmask.d.step : 0;
    o.rows = b->fh; o.cols = b->fw; o.bx0 = 0; o.grp = 0; o.gx = 0; o.gy = 0; o.xmagic = 0; o.band = 0; o.rec12 = 0;
    if (windowed) {
        // the kernels keep addressing the mosaic's columns: the mats' origins move left by the window's first column (a multiple of
        // ISX_WINDOW_GRANULE, so every alignment is kept and a block of the last step starts exactly there), the right crop is the
        // window's end; columns of the mats past the mosaic's right edge (a last strip padded to its peers' width) are left as they are
        const size_t px = dst->type == ISX_32FC3 ? 12 : (dst->type == ISX_8UC3 ? 3 : 6);
        o.img -= (size_t)b->win_x0 * px;
        if (o.mask) o.mask -= (size_t)b->win_x0;
        o.cols = std::min(b->win_x1, b->fw);
    }
    o.vec = ((uintptr_t)o.img % 4 == 0) && (o.img_step % 4 == 0) && (!o.mask || (((uintptr_t)o.mask % 2 == 0) && (o.mask_step % 2 == 0)));
    return ISX_OK;
}
// the copies back to host mats and the release of the pyramids (dst_pyr_laplace_.clear(); dst_band_weights_.clear())
static int blend_end(isx_blender* b, isx_mat* dst_mask) {
    if (b->win_x1 > b->win_x0) {
        // a last strip padded to its peers' width keeps the columns past the mosaic's edge as they are - also in a host mat, of which
        // only the computed columns are copied back from the staging buffer
        const int valid = std::min(b->win_x1, b->fw) - b->win_x0;
        ISX_TRY(b->st_out.finish_out_cols(b->stream, 0, valid));
        if (dst_mask) ISX_TRY(b->st_outmask.finish_out_cols(b->stream, 0, valid));
    } else {
        ISX_TRY(b->st_out.finish_out(b->stream));
        if (dst_mask) ISX_TRY(b->st_outmask.finish_out(b->stream));
    }
    b->path_fused = 0;
    for (const auto& r : b->tiles) b->path_fused += r.g1 != 0;
    b->tiles.clear();
    b->ftiles.clear();
    b->level0_pending = false;
    b->narrow_pending = 0; b->narrow_published = false; b->fused_cycle = false;
    b->prepared = false;   // dst_pyr_laplace_.clear(); dst_band_weights_.clear()
    return ISX_OK;
}

int isx_blender_blend(isx_blender* b, isx_mat* dst, isx_mat* dst_mask) ISX_ENTRY {
    clear_error();
    OutMat o;
    ISX_TRY(blend_begin(b, dst, dst_mask, &o));
    const bool windowed = b->win_x1 > b->win_x0;
    int rc = ISX_OK;
    if (b->type == ISX_BLEND_NO) {        // Blender::blend of the base class
        const size_t n = (size_t)b->rw * b->rh, img_bytes = (n * 6 + 255) & ~(size_t)255;
        ISX_LAUNCH("no_blend", (double)n * (7.0 + (o.img_f32 == 2 ? 4.0 : 7.0)), b->stream, k_no_blend, dim3(cdiv(b->rw, 64), cdiv(b->rh, 4)), dim3(256), 0,
                   (const short*)b->dst_arena.p, (const unsigned char*)b->dst_arena.p + img_bytes, b->rw, o);
    } else if (b->type == ISX_BLEND_FEATHER) {   // FeatherBlender::blend: normalizeUsingWeightMap, compare(w > WEIGHT_EPS), Blender::blend
        dim3 grid(cdiv(b->dst[0].cols, 64), cdiv(b->dst[0].rows, 4));
        if (!b->ftiles.empty()) {   // deferred cycle: gather over the recorded tiles
            if (windowed) {         // a pixel of the result depends on the tiles that cover it and on nothing else: the window's block columns
                o.bx0 = b->win_x0 / 64;
                grid.x = cdiv(o.cols, 64) - o.bx0;
            }
            FeatherSet fs;
            memset(&fs, 0, sizeof(fs));
            fs.n = (int)b->ftiles.size();
            double bytes = (double)(windowed ? o.cols - b->win_x0 : b->fw) * b->fh * 7.0;
            for (int t = 0; t < fs.n; ++t) {
                const auto& r = b->ftiles[t];
                fs.img[t] = r.img; fs.istep[t] = r.istep; fs.wgt[t] = r.wgt; fs.wpitch[t] = r.wpitch;
                fs.x[t] = r.dx; fs.y[t] = r.dy; fs.w[t] = r.cols; fs.h[t] = r.rows;
                bytes += (double)r.rows * r.cols * ((r.sk == SK_U8 ? 3.0 : 6.0) + 4.0);
            }
            if (b->ftiles[0].sk == SK_U8) ISX_LAUNCH("feather_gather", bytes, b->stream, (k_feather_gather<SK_U8>), grid, dim3(256), 0, fs, o);
            else ISX_LAUNCH("feather_gather", bytes, b->stream, (k_feather_gather<SK_S16>), grid, dim3(256), 0, fs, o);
        } else
        ISX_LAUNCH("feather_blend", (double)b->fw * b->fh * 17.0, b->stream, k_feather_blend, grid, dim3(256), 0, b->dst[0], o, make_cover(b, 0));
    } else {
        switch (b->prec) {
            case M_I16: rc = run_blend<M_I16>(b, o); break;
            case M_F32: rc = run_blend<M_F32>(b, o); break;
            default: rc = run_blend<M_F16>(b, o); break;
        }
    }
    ISX_TRY(rc);
    return blend_end(b, dst_mask);
} ISX_EXIT("isx_blender_blend")

int isx_blender_blend_batch(isx_blender** bs, int n, isx_mat* dsts, isx_mat* dst_masks) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(bs != nullptr && dsts != nullptr && n >= 1, ISX_ERR_INVALID, "blend_batch: bad argument");
    for (int i = 0; i < n; ++i) ISX_CHECK_ARG(bs[i] != nullptr, ISX_ERR_INVALID, "blend_batch: null blender %d", i);
    // Groups of blenders that can share a chain of launches: the deferred multi-band cycle with every tile still recorded, no window, the
    // same precision / bands / tile type / device / stream, at most BATCH_MAX mosaics and DEF_MAX tiles.  Anything else is blended alone.
    int i = 0;
    while (i < n) {
        isx_blender* b0 = bs[i];
        auto batchable = [&](const isx_blender* b) {
            return b->type == ISX_BLEND_MULTI_BAND && b->prepared && b->level0_pending && !b->tiles.empty() && b->win_x1 <= b->win_x0 &&
                   b->prec == b0->prec && b->num_bands == b0->num_bands && b->num_bands >= 1 && b->tiles[0].sk == b0->tiles[0].sk &&
                   b->device == b0->device && b->stream == b0->stream && !b->fused_cycle;      // (a fused-feed cycle has its level 1 already: blended alone)
        };
        int j = i, tiles = 0;
        if (batchable(b0))
            while (j < n && j - i < BATCH_MAX && batchable(bs[j]) && tiles + (int)bs[j]->tiles.size() <= DEF_MAX) {
                bool dup = false;
                for (int q = i; q < j; ++q) dup = dup || bs[q] == bs[j];
                if (dup) break;
                tiles += (int)bs[j]->tiles.size(); ++j;
            }
        if (j - i < 2) {
            ISX_TRY(isx_blender_blend(bs[i], &dsts[i], dst_masks ? &dst_masks[i] : nullptr));
            ++i;
            continue;
        }
        OutMat outs[BATCH_MAX];
        for (int q = i; q < j; ++q) ISX_TRY(blend_begin(bs[q], &dsts[q], dst_masks ? &dst_masks[q] : nullptr, &outs[q - i]));
        int rc;
        switch (b0->prec) {
            case M_I16: rc = run_blend_batch<M_I16>(bs + i, j - i, outs); break;
            case M_F32: rc = run_blend_batch<M_F32>(bs + i, j - i, outs); break;
            default: rc = run_blend_batch<M_F16>(bs + i, j - i, outs); break;
        }
        ISX_TRY(rc);
        for (int q = i; q < j; ++q) ISX_TRY(blend_end(bs[q], dst_masks ? &dst_masks[q] : nullptr));
        i = j;
    }
    return ISX_OK;
} ISX_EXIT("isx_blender_blend_batch")

}  // extern "C"
