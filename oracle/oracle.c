/*
 * oracle.c — CPU restatement of the reference's warp + multi-band blend hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h for the rules and the pinning status).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC oracle.c -lm  (oracle/build.sh)
 * FMA contraction MUST stay off: the fp32 association below is the spec the HIP kernels match.
 */
#include "oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* scalar conversion helpers                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* cvRound(float) in OpenCV 3.4.2 on x86-64 = _mm_cvtss_si32: round-half-even under the default
 * MXCSR; NaN and |v| >= 2^31 give the "integer indefinite" 0x80000000. */
int orc_cvround(float v) {
    if (!(fabsf(v) < 2147483648.0f)) return INT_MIN;
    return (int)lrintf(v); /* default rounding mode = round-half-even */
}

/* static_cast<int>(float) (W:83-86) on x86-64 = cvttss2si: truncation; indefinite as above. */
int orc_f2i_trunc(float v) {
    if (!(fabsf(v) < 2147483648.0f)) return INT_MIN;
    return (int)v;
}

/* static_cast<short>(float) as x86-64 compilers emit it for blenders.cpp: cvttss2si to a
 * 32-bit register, low 16 bits kept (UB in ISO C++ when out of range; this is what runs). */
static inline int16_t f2s_trunc(float v) { return (int16_t)(uint16_t)(uint32_t)orc_f2i_trunc(v); }

static inline int16_t sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
/* saturate_cast<short>(float) = saturate_cast<short>(cvRound(v)) */
static inline int16_t sat_s16_f(float v) { return sat_s16(orc_cvround(v)); }

/* cv::borderInterpolate (OpenCV core/src/copy.cpp) */
int orc_border_interpolate(int p, int len, int border) {
    if ((unsigned)p < (unsigned)len) return p;
    if (border == ORC_BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (border == ORC_BORDER_REFLECT || border == ORC_BORDER_REFLECT_101) {
        int delta = border == ORC_BORDER_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (border == ORC_BORDER_WRAP) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1; /* BORDER_CONSTANT */
}
#define BI orc_border_interpolate

/* f32 -> f16 (round-to-nearest-even, subnormals kept, overflow -> inf) -> f32 */
float orc_f16_round(float v) {
    uint32_t x; memcpy(&x, &v, 4);
    uint32_t sign = x & 0x80000000u, a = x & 0x7fffffffu;
    uint32_t out;
    if (a >= 0x7f800000u) { out = a; /* inf / nan unchanged (nan payload kept) */ }
    else if (a >= 0x477ff000u) { out = 0x7f800000u; /* >= 65520 rounds to inf */ }
    else if (a < 0x33000001u) { out = 0; /* <= 2^-25 rounds to 0 (tie at 2^-25 -> even = 0) */ }
    else if (a < 0x38800000u) {
        /* half subnormal: quantum 2^-24 */
        float f; memcpy(&f, &a, 4);
        float q = f * 16777216.0f;            /* exact: f * 2^24 */
        float r = nearbyintf(q);              /* RNE */
        float back = r * (1.0f / 16777216.0f);
        memcpy(&out, &back, 4);
    } else {
        /* normal half: keep 10 mantissa bits, RNE on the 13 dropped bits */
        uint32_t lsb = (a >> 13) & 1u;
        out = (a + 0xfffu + lsb) & ~0x1fffu;
    }
    out |= sign;
    float r; memcpy(&r, &out, 4);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* A2 setCameraParams W:90-120                                                                */
/* ------------------------------------------------------------------------------------------ */
/* K.inv() on a 3x3 CV_32F Mat in OpenCV 3.4.2 (core/src/lapack.cpp cv::invert, n == 3):
 * determinant and adjugate in double, each entry rounded to float once.
 * R * K.inv() and K * Rinv are cv::gemm on CV_32F: GEMMSingleMul<float,double>, i.e. the dot
 * products accumulate in double and round to float once. */
static void mat3_mul_f32(const float a[9], const float b[9], float c[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (double)a[i * 3 + k] * (double)b[k * 3 + j];
            c[i * 3 + j] = (float)s;
        }
}
static void mat3_inv_f32(const float s[9], float d[9]) {
#define S(i, j) ((double)s[(i) * 3 + (j)])
    double det = S(0,0) * (S(1,1) * S(2,2) - S(1,2) * S(2,1)) - S(0,1) * (S(1,0) * S(2,2) - S(1,2) * S(2,0)) +
                 S(0,2) * (S(1,0) * S(2,1) - S(1,1) * S(2,0));
    if (det == 0.0) { memset(d, 0, 9 * sizeof(float)); return; }
    double id = 1.0 / det;
    d[0] = (float)((S(1,1) * S(2,2) - S(1,2) * S(2,1)) * id);
    d[1] = (float)((S(0,2) * S(2,1) - S(0,1) * S(2,2)) * id);
    d[2] = (float)((S(0,1) * S(1,2) - S(0,2) * S(1,1)) * id);
    d[3] = (float)((S(1,2) * S(2,0) - S(1,0) * S(2,2)) * id);
    d[4] = (float)((S(0,0) * S(2,2) - S(0,2) * S(2,0)) * id);
    d[5] = (float)((S(0,2) * S(1,0) - S(0,0) * S(1,2)) * id);
    d[6] = (float)((S(1,0) * S(2,1) - S(1,1) * S(2,0)) * id);
    d[7] = (float)((S(0,1) * S(2,0) - S(0,0) * S(2,1)) * id);
    d[8] = (float)((S(0,0) * S(1,1) - S(0,1) * S(1,0)) * id);
#undef S
}
void orc_camera(const float K[9], const float R[9], float k[9], float rinv[9],
                float r_kinv[9], float k_rinv[9]) {
    float kinv[9];
    memcpy(k, K, 9 * sizeof(float));                                         /* W:98-101 */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rinv[i * 3 + j] = R[j * 3 + i]; /* W:103 */
    mat3_inv_f32(K, kinv);
    mat3_mul_f32(R, kinv, r_kinv);                                           /* W:108 */
    mat3_mul_f32(K, rinv, k_rinv);                                           /* W:113 */
}

/* ------------------------------------------------------------------------------------------ */
/* A3/A4 projector                                                                            */
/* ------------------------------------------------------------------------------------------ */
#define ORC_PI_F ((float)3.1415926535897932384626433832795)

void orc_map_forward(int kind, float scale, const float r_kinv[9], float x, float y, float* u, float* v) {
    /* W:38-40 — evaluated exactly as written, left to right, no FMA */
    float x_ = r_kinv[0] * x + r_kinv[1] * y + r_kinv[2];
    float y_ = r_kinv[3] * x + r_kinv[4] * y + r_kinv[5];
    float z_ = r_kinv[6] * x + r_kinv[7] * y + r_kinv[8];
    if (kind == ORC_CYL) {
        *u = scale * atan2f(x_, z_);                        /* W:42 */
        *v = scale * y_ / sqrtf(x_ * x_ + z_ * z_);         /* W:43 */
    } else {
        /* OpenCV warpers_inl.hpp SphericalProjector::mapForward */
        *u = scale * atan2f(x_, z_);
        float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        *v = scale * (ORC_PI_F - acosf(w == w ? w : 0));
    }
}

void orc_map_backward(int kind, float scale, const float k_rinv[9], float u, float v, float* x, float* y) {
    float x_, y_, z_;
    u /= scale;                                             /* W:48 */
    v /= scale;                                             /* W:49 */
    if (kind == ORC_CYL) {
        x_ = sinf(u); y_ = v; z_ = cosf(u);                 /* W:51-53 */
    } else {
        float sinv = sinf(ORC_PI_F - v);
        x_ = sinv * sinf(u);
        y_ = cosf(ORC_PI_F - v);
        z_ = sinv * cosf(u);
    }
    float z;
    *x = k_rinv[0] * x_ + k_rinv[1] * y_ + k_rinv[2] * z_;  /* W:56 */
    *y = k_rinv[3] * x_ + k_rinv[4] * y_ + k_rinv[5] * z_;  /* W:57 */
    z  = k_rinv[6] * x_ + k_rinv[7] * y_ + k_rinv[8] * z_;  /* W:58 */
    if (z > 0) { *x /= z; *y /= z; }                        /* W:60 */
    else *x = *y = -1;                                      /* W:61 */
}

/* ------------------------------------------------------------------------------------------ */
/* A5 detectResultRoi                                                                         */
/* ------------------------------------------------------------------------------------------ */
#define MINF(a, b) ((b) < (a) ? (b) : (a))   /* (std::min)(a,b) */
#define MAXF(a, b) ((a) < (b) ? (b) : (a))   /* (std::max)(a,b) */

void orc_detect_roi(int kind, float scale, const float k[9], const float rinv[9],
                    const float r_kinv[9], int src_w, int src_h, int roi[4], float minmax[4]) {
    float tl_uf = 3.402823466e+38f, tl_vf = 3.402823466e+38f;      /* W:66-69 */
    float br_uf = -3.402823466e+38f, br_vf = -3.402823466e+38f;
    float u, v;
    if (kind == ORC_CYL) {
        /* W:72-81: the in-tree warper scans EVERY source pixel (RotationWarperBase::detectResultRoi) */
        for (int y = 0; y < src_h; ++y)
            for (int x = 0; x < src_w; ++x) {
                orc_map_forward(kind, scale, r_kinv, (float)x, (float)y, &u, &v);
                tl_uf = MINF(tl_uf, u); tl_vf = MINF(tl_vf, v);
                br_uf = MAXF(br_uf, u); br_vf = MAXF(br_vf, v);
            }
    } else {
        /* OpenCV SphericalWarper::detectResultRoi = detectResultRoiByBorder + pole tests */
        for (int i = 0; i < src_w; ++i) {
            orc_map_forward(kind, scale, r_kinv, (float)i, 0.f, &u, &v);
            tl_uf = MINF(tl_uf, u); tl_vf = MINF(tl_vf, v); br_uf = MAXF(br_uf, u); br_vf = MAXF(br_vf, v);
            orc_map_forward(kind, scale, r_kinv, (float)i, (float)(src_h - 1), &u, &v);
            tl_uf = MINF(tl_uf, u); tl_vf = MINF(tl_vf, v); br_uf = MAXF(br_uf, u); br_vf = MAXF(br_vf, v);
        }
        for (int i = 0; i < src_h; ++i) {
            orc_map_forward(kind, scale, r_kinv, 0.f, (float)i, &u, &v);
            tl_uf = MINF(tl_uf, u); tl_vf = MINF(tl_vf, v); br_uf = MAXF(br_uf, u); br_vf = MAXF(br_vf, v);
            orc_map_forward(kind, scale, r_kinv, (float)(src_w - 1), (float)i, &u, &v);
            tl_uf = MINF(tl_uf, u); tl_vf = MINF(tl_vf, v); br_uf = MAXF(br_uf, u); br_vf = MAXF(br_vf, v);
        }
        /* ByBorder casts to int, the pole code re-reads those ints as floats */
        tl_uf = (float)orc_f2i_trunc(tl_uf); tl_vf = (float)orc_f2i_trunc(tl_vf);
        br_uf = (float)orc_f2i_trunc(br_uf); br_vf = (float)orc_f2i_trunc(br_vf);
        float x = rinv[1], y = rinv[4], z = rinv[7];
        if (y > 0.f) {
            float x_ = (k[0] * x + k[1] * y) / z + k[2];
            float y_ = k[4] * y / z + k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                float pv = (float)(3.1415926535897932384626433832795 * scale);
                tl_uf = MINF(tl_uf, 0.f); tl_vf = MINF(tl_vf, pv);
                br_uf = MAXF(br_uf, 0.f); br_vf = MAXF(br_vf, pv);
            }
        }
        x = rinv[1]; y = -rinv[4]; z = rinv[7];
        if (y > 0.f) {
            float x_ = (k[0] * x + k[1] * y) / z + k[2];
            float y_ = k[4] * y / z + k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                tl_uf = MINF(tl_uf, 0.f); tl_vf = MINF(tl_vf, 0.f);
                br_uf = MAXF(br_uf, 0.f); br_vf = MAXF(br_vf, 0.f);
            }
        }
    }
    if (minmax) { minmax[0] = tl_uf; minmax[1] = tl_vf; minmax[2] = br_uf; minmax[3] = br_vf; }
    roi[0] = orc_f2i_trunc(tl_uf);   /* W:83-86: static_cast<int>, truncation toward zero */
    roi[1] = orc_f2i_trunc(tl_vf);
    roi[2] = orc_f2i_trunc(br_uf);
    roi[3] = orc_f2i_trunc(br_vf);
}

/* A6 buildMaps W:122-144 (the map fill, W:133-141) */
void orc_build_maps(int kind, float scale, const float k_rinv[9], const int roi[4], float* xmap, float* ymap) {
    int mw = roi[2] - roi[0] + 1;
    for (int v = roi[1]; v <= roi[3]; ++v)
        for (int u = roi[0]; u <= roi[2]; ++u) {
            float x, y;
            orc_map_backward(kind, scale, k_rinv, (float)u, (float)v, &x, &y);
            xmap[(size_t)(v - roi[1]) * mw + (u - roi[0])] = x;
            ymap[(size_t)(v - roi[1]) * mw + (u - roi[0])] = y;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* A8 cv::remap (OpenCV 3.4.2 imgproc/src/imgwarp.cpp), CV_32FC1 maps                         */
/* ------------------------------------------------------------------------------------------ */
#define INTER_BITS 5
#define INTER_TAB_SIZE 32
#define REMAP_COEF_BITS 15
#define REMAP_COEF_SCALE 32768

/* BilinearTab_i as initInterTab2D builds it: float products (exact multiples of 1/1024) times
 * 32768, saturate_cast<short>; only entry (fy=0,fx=0) sums to 32767 and the fix-up adds the
 * missing 1 to tap [1][1] -> {32767,0,0,1}. */
static void bilinear_wtab_i(int fx, int fy, int w[4]) {
    float ax[2] = { 1.f - fx * (1.f / INTER_TAB_SIZE), fx * (1.f / INTER_TAB_SIZE) };
    float ay[2] = { 1.f - fy * (1.f / INTER_TAB_SIZE), fy * (1.f / INTER_TAB_SIZE) };
    int isum = 0;
    for (int k1 = 0; k1 < 2; ++k1)
        for (int k2 = 0; k2 < 2; ++k2) {
            float v = ay[k1] * ax[k2];
            int iv = orc_cvround(v * REMAP_COEF_SCALE);
            if (iv > 32767) iv = 32767;
            w[k1 * 2 + k2] = iv; isum += iv;
        }
    if (isum != REMAP_COEF_SCALE) w[3] += REMAP_COEF_SCALE - isum; /* only (0,0): diff = -1 */
}
static void bilinear_wtab_f(int fx, int fy, float w[4]) {
    float ax[2] = { 1.f - fx * (1.f / INTER_TAB_SIZE), fx * (1.f / INTER_TAB_SIZE) };
    float ay[2] = { 1.f - fy * (1.f / INTER_TAB_SIZE), fy * (1.f / INTER_TAB_SIZE) };
    for (int k1 = 0; k1 < 2; ++k1) for (int k2 = 0; k2 < 2; ++k2) w[k1 * 2 + k2] = ay[k1] * ax[k2];
}
static inline int sat_short_i(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

#define REMAP_BODY(T, WT, WTAB, CAST)                                                            \
    for (int dy = 0; dy < dh; ++dy) {                                                            \
        T* D = (T*)((char*)dst + (size_t)dy * dstep);                                            \
        for (int dx = 0; dx < dw; ++dx, D += cn) {                                               \
            float mx = xmap[(size_t)dy * dw + dx], my = ymap[(size_t)dy * dw + dx];              \
            if (interp == ORC_NEAREST) {                                                         \
                /* saturate_cast<short>(float) = saturate(cvRound) */                            \
                int sx = sat_short_i(orc_cvround(mx)), sy = sat_short_i(orc_cvround(my));        \
                if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {                \
                    const T* S = (const T*)((const char*)src + (size_t)sy * sstep) + sx * cn;    \
                    for (int c = 0; c < cn; ++c) D[c] = S[c];                                    \
                } else if (border == ORC_BORDER_CONSTANT) {                                      \
                    for (int c = 0; c < cn; ++c) D[c] = 0;                                       \
                } else {                                                                         \
                    sx = BI(sx, sw, border); sy = BI(sy, sh, border);                            \
                    const T* S = (const T*)((const char*)src + (size_t)sy * sstep) + sx * cn;    \
                    for (int c = 0; c < cn; ++c) D[c] = S[c];                                    \
                }                                                                                \
                continue;                                                                        \
            }                                                                                    \
            int isx = orc_cvround(mx * INTER_TAB_SIZE), isy = orc_cvround(my * INTER_TAB_SIZE);  \
            int fx = isx & (INTER_TAB_SIZE - 1), fy = isy & (INTER_TAB_SIZE - 1);                \
            int sx = sat_short_i(isx >> INTER_BITS), sy = sat_short_i(isy >> INTER_BITS);        \
            WT w[4]; WTAB(fx, fy, w);                                                            \
            if (border == ORC_BORDER_CONSTANT && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) { \
                for (int c = 0; c < cn; ++c) D[c] = 0;                                           \
                continue;                                                                        \
            }                                                                                    \
            int sx0, sx1, sy0, sy1;                                                              \
            if (border == ORC_BORDER_REPLICATE) {                                                \
                sx0 = sx < 0 ? 0 : (sx > sw - 1 ? sw - 1 : sx);                                  \
                sx1 = sx + 1 < 0 ? 0 : (sx + 1 > sw - 1 ? sw - 1 : sx + 1);                      \
                sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);                                  \
                sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);                      \
            } else {                                                                             \
                sx0 = BI(sx, sw, border); sx1 = BI(sx + 1, sw, border);                          \
                sy0 = BI(sy, sh, border); sy1 = BI(sy + 1, sh, border);                          \
            }                                                                                    \
            for (int c = 0; c < cn; ++c) {                                                       \
                WT v0 = (sx0 >= 0 && sy0 >= 0) ? ((const T*)((const char*)src + (size_t)sy0 * sstep))[sx0 * cn + c] : 0; \
                WT v1 = (sx1 >= 0 && sy0 >= 0) ? ((const T*)((const char*)src + (size_t)sy0 * sstep))[sx1 * cn + c] : 0; \
                WT v2 = (sx0 >= 0 && sy1 >= 0) ? ((const T*)((const char*)src + (size_t)sy1 * sstep))[sx0 * cn + c] : 0; \
                WT v3 = (sx1 >= 0 && sy1 >= 0) ? ((const T*)((const char*)src + (size_t)sy1 * sstep))[sx1 * cn + c] : 0; \
                D[c] = CAST(v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3]);                      \
            }                                                                                    \
        }                                                                                        \
    }

/* FixedPtCast<int, uchar, 15>: saturate_cast<uchar>((v + (1 << 14)) >> 15) */
#define CAST_FIX15(v) sat_u8(((v) + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS)
#define CAST_ID(v) (v)

void orc_remap_u8(const uint8_t* src, int sh, int sw, int cn, size_t sstep,
                  uint8_t* dst, int dh, int dw, size_t dstep,
                  const float* xmap, const float* ymap, int interp, int border) {
    REMAP_BODY(uint8_t, int, bilinear_wtab_i, CAST_FIX15)
}
void orc_remap_f32(const float* src, int sh, int sw, int cn, size_t sstep,
                   float* dst, int dh, int dw, size_t dstep,
                   const float* xmap, const float* ymap, int interp, int border) {
    REMAP_BODY(float, float, bilinear_wtab_f, CAST_ID)
}

/* A7 warp W:145-161 */
void orc_warp_u8(int kind, float scale, const float K[9], const float R[9],
                 const uint8_t* src, int sh, int sw, int cn, int interp, int border,
                 int roi[4], uint8_t* dst) {
    float k[9], rinv[9], r_kinv[9], k_rinv[9];
    orc_camera(K, R, k, rinv, r_kinv, k_rinv);                                   /* W:124 */
    orc_detect_roi(kind, scale, k, rinv, r_kinv, sw, sh, roi, NULL);             /* W:126 */
    if (!dst) return;
    int mh = roi[3] - roi[1] + 1, mw = roi[2] - roi[0] + 1;                      /* W:128-129 */
    float* xmap = (float*)malloc((size_t)mh * mw * sizeof(float));
    float* ymap = (float*)malloc((size_t)mh * mw * sizeof(float));
    orc_build_maps(kind, scale, k_rinv, roi, xmap, ymap);                        /* W:133-141 */
    /* dst.create(dst_roi.height + 1, dst_roi.width + 1) (W:150) == the map size */
    orc_remap_u8(src, sh, sw, cn, (size_t)sw * cn, dst, mh, mw, (size_t)mw * cn, xmap, ymap, interp, border); /* W:157 */
    free(xmap); free(ymap);
}

/* ------------------------------------------------------------------------------------------ */
/* A10 pyrDown / pyrUp (OpenCV 3.4.2 imgproc/src/pyramids.cpp, scalar association)            */
/* ------------------------------------------------------------------------------------------ */
/* pyrDown: dst ((h+1)/2, (w+1)/2); separable [1 4 6 4 1], BORDER_REFLECT_101;
 *   row pass  r = s[2x]*6 + (s[2x-1] + s[2x+1])*4 + s[2x-2] + s[2x+2]
 *   col pass  v = r2*6 + (r1 + r3)*4 + r0 + r4
 *   cast      int16: (v + 128) >> 8 (FixPtCast<short,8>), f32: v * (1/256.f) (FltCast<float,8>) */
#define PYR_DOWN(NAME, T, WT, CAST)                                                              \
    void NAME(const T* src, int sh, int sw, int cn, T* dst) {                                    \
        int dh = (sh + 1) / 2, dw = (sw + 1) / 2;                                                \
        WT* hb = (WT*)malloc((size_t)sh * dw * cn * sizeof(WT));                                 \
        for (int y = 0; y < sh; ++y) {                                                           \
            const T* s = src + (size_t)y * sw * cn;                                              \
            WT* r = hb + (size_t)y * dw * cn;                                                    \
            for (int x = 0; x < dw; ++x) {                                                       \
                int i0 = BI(2 * x - 2, sw, ORC_BORDER_REFLECT_101) * cn, i1 = BI(2 * x - 1, sw, ORC_BORDER_REFLECT_101) * cn; \
                int i2 = BI(2 * x, sw, ORC_BORDER_REFLECT_101) * cn;                             \
                int i3 = BI(2 * x + 1, sw, ORC_BORDER_REFLECT_101) * cn, i4 = BI(2 * x + 2, sw, ORC_BORDER_REFLECT_101) * cn; \
                for (int c = 0; c < cn; ++c)                                                     \
                    r[x * cn + c] = (WT)s[i2 + c] * 6 + ((WT)s[i1 + c] + (WT)s[i3 + c]) * 4 + (WT)s[i0 + c] + (WT)s[i4 + c]; \
            }                                                                                    \
        }                                                                                        \
        for (int y = 0; y < dh; ++y) {                                                           \
            const WT* r0 = hb + (size_t)BI(2 * y - 2, sh, ORC_BORDER_REFLECT_101) * dw * cn;     \
            const WT* r1 = hb + (size_t)BI(2 * y - 1, sh, ORC_BORDER_REFLECT_101) * dw * cn;     \
            const WT* r2 = hb + (size_t)BI(2 * y, sh, ORC_BORDER_REFLECT_101) * dw * cn;         \
            const WT* r3 = hb + (size_t)BI(2 * y + 1, sh, ORC_BORDER_REFLECT_101) * dw * cn;     \
            const WT* r4 = hb + (size_t)BI(2 * y + 2, sh, ORC_BORDER_REFLECT_101) * dw * cn;     \
            T* d = dst + (size_t)y * dw * cn;                                                    \
            for (int x = 0; x < dw * cn; ++x)                                                    \
                d[x] = CAST(r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x]);                    \
        }                                                                                        \
        free(hb);                                                                                \
    }
#define CAST_DOWN_S16(v) sat_s16(((v) + 128) >> 8)
#define CAST_DOWN_F32(v) ((v) * (1.f / 256.f))
PYR_DOWN(orc_pyr_down_s16, int16_t, int, CAST_DOWN_S16)
PYR_DOWN(orc_pyr_down_f32, float, float, CAST_DOWN_F32)

/* pyrUp to exactly (2h, 2w):
 *   row pass, source row n wide:  even t0(x) = s[x-1] + s[x]*6 + s[x+1], odd t1(x) = (s[x] + s[x+1])*4
 *             left edge x=0:      t0 = s[0]*6 + s[1]*2          t1 = (s[0] + s[1])*4
 *             right edge x=n-1:   t0 = s[n-2] + s[n-1]*7        t1 = s[n-1]*8
 *             n == 1:             t0 = t1 = s[0]*8
 *   rows:     source row -1 := row 1, row h := row h-1 (borderInterpolate(2sy, 2h, REFLECT_101)/2)
 *   col pass  even d0 = r0 + r1*6 + r2,  odd d1 = (r1 + r2)*4
 *   cast      int16: (v + 32) >> 6, f32: v * (1/64.f) */
#define PYR_UP(NAME, T, WT, CAST)                                                                \
    void NAME(const T* src, int sh, int sw, int cn, T* dst) {                                    \
        int dw = sw * 2;                                                                         \
        WT* hb = (WT*)malloc((size_t)sh * dw * cn * sizeof(WT));                                 \
        for (int y = 0; y < sh; ++y) {                                                           \
            const T* s = src + (size_t)y * sw * cn;                                              \
            WT* r = hb + (size_t)y * dw * cn;                                                    \
            for (int c = 0; c < cn; ++c) {                                                       \
                if (sw == 1) { r[c] = r[cn + c] = (WT)s[c] * 8; continue; }                      \
                r[c] = (WT)s[c] * 6 + (WT)s[cn + c] * 2;                                         \
                r[cn + c] = ((WT)s[c] + (WT)s[cn + c]) * 4;                                      \
                int sx = (sw - 1) * cn + c;                                                      \
                r[(dw - 2) * cn + c] = (WT)s[sx - cn] + (WT)s[sx] * 7;                           \
                r[(dw - 1) * cn + c] = (WT)s[sx] * 8;                                            \
                for (int x = 1; x < sw - 1; ++x) {                                               \
                    int i = x * cn + c;                                                          \
                    r[2 * x * cn + c] = (WT)s[i - cn] + (WT)s[i] * 6 + (WT)s[i + cn];            \
                    r[(2 * x + 1) * cn + c] = ((WT)s[i] + (WT)s[i + cn]) * 4;                    \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        for (int y = 0; y < sh; ++y) {                                                           \
            const WT* r0 = hb + (size_t)(BI(2 * (y - 1), 2 * sh, ORC_BORDER_REFLECT_101) / 2) * dw * cn; \
            const WT* r1 = hb + (size_t)y * dw * cn;                                             \
            const WT* r2 = hb + (size_t)(BI(2 * (y + 1), 2 * sh, ORC_BORDER_REFLECT_101) / 2) * dw * cn; \
            T* d0 = dst + (size_t)(2 * y) * dw * cn;                                             \
            T* d1 = d0 + (size_t)dw * cn;                                                        \
            for (int x = 0; x < dw * cn; ++x) {                                                  \
                d1[x] = CAST((r1[x] + r2[x]) * 4);                                               \
                d0[x] = CAST(r0[x] + r1[x] * 6 + r2[x]);                                         \
            }                                                                                    \
        }                                                                                        \
        free(hb);                                                                                \
    }
#define CAST_UP_S16(v) sat_s16(((v) + 32) >> 6)
#define CAST_UP_F32(v) ((v) * (1.f / 64.f))
PYR_UP(orc_pyr_up_s16, int16_t, int, CAST_UP_S16)
PYR_UP(orc_pyr_up_f32, float, float, CAST_UP_F32)

/* ------------------------------------------------------------------------------------------ */
/* A9/A11/A12 MultiBandBlender (OpenCV 3.4.2 stitching/src/blenders.cpp)                      */
/* ------------------------------------------------------------------------------------------ */
#define WEIGHT_EPS 1e-5f
#define ORC_MAX_LEVELS 32

struct orc_mb {
    int actual_num_bands, num_bands, prec;
    int rx, ry, rw, rh;        /* dst_roi_ (padded)           */
    int fw, fh;                /* dst_roi_final_ width/height */
    int lrows[ORC_MAX_LEVELS], lcols[ORC_MAX_LEVELS];
    int16_t* lap_s[ORC_MAX_LEVELS]; /* I16 */
    float*   lap_f[ORC_MAX_LEVELS]; /* F32 / F16ACC32 */
    float*   wgt[ORC_MAX_LEVELS];
    int prepared;
};

orc_mb* orc_mb_create(int num_bands, int precision) {
    orc_mb* b = (orc_mb*)calloc(1, sizeof(orc_mb));
    /* setNumBands: actual_num_bands_ = val (the ctor default is 5) */
    b->actual_num_bands = num_bands; b->prec = precision;
    return b;
}
static void mb_release(orc_mb* b) {
    for (int i = 0; i < ORC_MAX_LEVELS; ++i) {
        free(b->lap_s[i]); free(b->lap_f[i]); free(b->wgt[i]);
        b->lap_s[i] = NULL; b->lap_f[i] = NULL; b->wgt[i] = NULL;
    }
    b->prepared = 0;
}
void orc_mb_destroy(orc_mb* b) { if (b) { mb_release(b); free(b); } }
int  orc_mb_num_bands(const orc_mb* b) { return b->num_bands; }
void orc_mb_result_size(const orc_mb* b, int* w, int* h) { *w = b->fw; *h = b->fh; }

void orc_mb_prepare(orc_mb* b, int n, const int* c, const int* s) {
    mb_release(b);
    /* resultRoi(corners, sizes): tl = min corners, br = max(corner + size) (stitching/util.cpp) */
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {
        if (c[2 * i] < tlx) tlx = c[2 * i];
        if (c[2 * i + 1] < tly) tly = c[2 * i + 1];
        if (c[2 * i] + s[2 * i] > brx) brx = c[2 * i] + s[2 * i];
        if (c[2 * i + 1] + s[2 * i + 1] > bry) bry = c[2 * i + 1] + s[2 * i + 1];
    }
    int w = brx - tlx, h = bry - tly;
    b->fw = w; b->fh = h;                                   /* dst_roi_final_ */
    /* num_bands_ = min(actual_num_bands_, (int)ceil(log(max_len)/log(2.0))) */
    double max_len = (double)(w > h ? w : h);
    int cl = (int)ceil(log(max_len) / log(2.0));
    b->num_bands = b->actual_num_bands < cl ? b->actual_num_bands : cl;
    int L = b->num_bands, m = 1 << L;
    w += (m - w % m) % m;
    h += (m - h % m) % m;
    b->rx = tlx; b->ry = tly; b->rw = w; b->rh = h;
    int rows = h, cols = w;
    for (int i = 0; i <= L; ++i) {
        b->lrows[i] = rows; b->lcols[i] = cols;
        if (b->prec == ORC_I16) b->lap_s[i] = (int16_t*)calloc((size_t)rows * cols * 3, sizeof(int16_t));
        else b->lap_f[i] = (float*)calloc((size_t)rows * cols * 3, sizeof(float));
        b->wgt[i] = (float*)calloc((size_t)rows * cols, sizeof(float));
        rows = (rows + 1) / 2; cols = (cols + 1) / 2;
    }
    b->prepared = 1;
}

void orc_mb_feed(orc_mb* b, const void* img, int img_is_f32, const uint8_t* mask,
                 int rows, int cols, int tl_x, int tl_y) {
    int L = b->num_bands, m = 1 << L;
    int gap = 3 * m;
    int br_dx = b->rx + b->rw, br_dy = b->ry + b->rh;       /* dst_roi_.br() */
    int tlnx = b->rx > tl_x - gap ? b->rx : tl_x - gap;
    int tlny = b->ry > tl_y - gap ? b->ry : tl_y - gap;
    int brnx = br_dx < tl_x + cols + gap ? br_dx : tl_x + cols + gap;
    int brny = br_dy < tl_y + rows + gap ? br_dy : tl_y + rows + gap;
    tlnx = b->rx + (((tlnx - b->rx) >> L) << L);
    tlny = b->ry + (((tlny - b->ry) >> L) << L);
    int width = brnx - tlnx, height = brny - tlny;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    brnx = tlnx + width; brny = tlny + height;
    int dy = brny - br_dy > 0 ? brny - br_dy : 0;
    int dx = brnx - br_dx > 0 ? brnx - br_dx : 0;
    tlnx -= dx; brnx -= dx; tlny -= dy; brny -= dy;
    int top = tl_y - tlny, left = tl_x - tlnx;
    /* bottom = br_new.y - tl.y - img.rows; right likewise: implied by width/height */

    /* level sizes of the tile pyramid */
    int prow[ORC_MAX_LEVELS], pcol[ORC_MAX_LEVELS];
    prow[0] = height; pcol[0] = width;
    for (int i = 1; i <= L; ++i) { prow[i] = (prow[i - 1] + 1) / 2; pcol[i] = (pcol[i - 1] + 1) / 2; }

    int isf = b->prec != ORC_I16;
    int16_t* gs[ORC_MAX_LEVELS] = {0}; float* gf[ORC_MAX_LEVELS] = {0}; float* wp[ORC_MAX_LEVELS] = {0};
    /* copyMakeBorder(img, BORDER_REFLECT) and weight = mask*(1/255.f), copyMakeBorder(CONSTANT 0) */
    size_t n0 = (size_t)height * width;
    if (isf) gf[0] = (float*)malloc(n0 * 3 * sizeof(float)); else gs[0] = (int16_t*)malloc(n0 * 3 * sizeof(int16_t));
    wp[0] = (float*)malloc(n0 * sizeof(float));
    const float inv255 = (float)(1. / 255.);
    for (int y = 0; y < height; ++y) {
        int syr = y - top, sy = BI(syr, rows, ORC_BORDER_REFLECT);
        for (int x = 0; x < width; ++x) {
            int sxr = x - left, sx = BI(sxr, cols, ORC_BORDER_REFLECT);
            size_t si = ((size_t)sy * cols + sx) * 3, di = ((size_t)y * width + x) * 3;
            for (int c = 0; c < 3; ++c) {
                if (isf) gf[0][di + c] = img_is_f32 ? ((const float*)img)[si + c] : (float)((const int16_t*)img)[si + c];
                else gs[0][di + c] = img_is_f32 ? sat_s16_f(((const float*)img)[si + c]) : ((const int16_t*)img)[si + c];
            }
            int inside = (unsigned)syr < (unsigned)rows && (unsigned)sxr < (unsigned)cols;
            wp[0][(size_t)y * width + x] = inside ? (float)mask[(size_t)syr * cols + sxr] * inv255 : 0.f;
        }
    }
    /* createLaplacePyr (non-8U branch): Gaussian chain, then pyr[i] -= pyrUp(pyr[i+1]) */
    for (int i = 0; i < L; ++i) {
        size_t n1 = (size_t)prow[i + 1] * pcol[i + 1];
        if (isf) {
            gf[i + 1] = (float*)malloc(n1 * 3 * sizeof(float));
            orc_pyr_down_f32(gf[i], prow[i], pcol[i], 3, gf[i + 1]);
            if (b->prec == ORC_F16ACC32) for (size_t k = 0; k < n1 * 3; ++k) gf[i + 1][k] = orc_f16_round(gf[i + 1][k]);
        } else {
            gs[i + 1] = (int16_t*)malloc(n1 * 3 * sizeof(int16_t));
            orc_pyr_down_s16(gs[i], prow[i], pcol[i], 3, gs[i + 1]);
        }
        wp[i + 1] = (float*)malloc(n1 * sizeof(float));
        orc_pyr_down_f32(wp[i], prow[i], pcol[i], 1, wp[i + 1]);
        if (b->prec == ORC_F16ACC32) for (size_t k = 0; k < n1; ++k) wp[i + 1][k] = orc_f16_round(wp[i + 1][k]);
    }
    for (int i = 0; i < L; ++i) {
        size_t n = (size_t)prow[i] * pcol[i] * 3;
        if (isf) {
            float* up = (float*)malloc(n * sizeof(float));
            orc_pyr_up_f32(gf[i + 1], prow[i + 1], pcol[i + 1], 3, up);
            for (size_t k = 0; k < n; ++k) gf[i][k] = gf[i][k] - up[k];
            free(up);
        } else {
            int16_t* up = (int16_t*)malloc(n * sizeof(int16_t));
            orc_pyr_up_s16(gs[i + 1], prow[i + 1], pcol[i + 1], 3, up);
            for (size_t k = 0; k < n; ++k) gs[i][k] = sat_s16((int)gs[i][k] - (int)up[k]); /* cv::subtract saturates */
            free(up);
        }
    }
    /* accumulate */
    int y_tl = tlny - b->ry, y_br = brny - b->ry, x_tl = tlnx - b->rx, x_br = brnx - b->rx;
    for (int i = 0; i <= L; ++i) {
        int rcw = x_br - x_tl, rch = y_br - y_tl;
        for (int y = 0; y < rch; ++y)
            for (int x = 0; x < rcw; ++x) {
                size_t si = (size_t)y * pcol[i] + x;
                size_t di = (size_t)(y + y_tl) * b->lcols[i] + (x + x_tl);
                float w = wp[i][si];
                for (int c = 0; c < 3; ++c) {
                    if (isf) b->lap_f[i][di * 3 + c] = b->lap_f[i][di * 3 + c] + gf[i][si * 3 + c] * w;
                    else b->lap_s[i][di * 3 + c] = (int16_t)(b->lap_s[i][di * 3 + c] + f2s_trunc((float)gs[i][si * 3 + c] * w));
                }
                b->wgt[i][di] = b->wgt[i][di] + w;
            }
        x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
    }
    for (int i = 0; i <= L; ++i) { free(gs[i]); free(gf[i]); free(wp[i]); }
}

void orc_mb_level(const orc_mb* b, int level, void* lap, float* weight, int* rows, int* cols) {
    *rows = b->lrows[level]; *cols = b->lcols[level];
    size_t n = (size_t)*rows * *cols;
    if (lap) {
        if (b->prec == ORC_I16) memcpy(lap, b->lap_s[level], n * 3 * sizeof(int16_t));
        else memcpy(lap, b->lap_f[level], n * 3 * sizeof(float));
    }
    if (weight) memcpy(weight, b->wgt[level], n * sizeof(float));
}

void orc_mb_blend(orc_mb* b, void* dst, int dst_is_f32, uint8_t* dst_mask) {
    int L = b->num_bands, isf = b->prec != ORC_I16;
    /* normalizeUsingWeightMap on every level */
    for (int i = 0; i <= L; ++i) {
        size_t n = (size_t)b->lrows[i] * b->lcols[i];
        for (size_t k = 0; k < n; ++k) {
            float d = b->wgt[i][k] + WEIGHT_EPS;
            for (int c = 0; c < 3; ++c) {
                if (isf) b->lap_f[i][k * 3 + c] = b->lap_f[i][k * 3 + c] / d;
                else b->lap_s[i][k * 3 + c] = f2s_trunc((float)b->lap_s[i][k * 3 + c] / d);
            }
        }
    }
    /* restoreImageFromLaplacePyr: pyr[i-1] = pyrUp(pyr[i]) + pyr[i-1] */
    for (int i = L; i > 0; --i) {
        size_t n = (size_t)b->lrows[i - 1] * b->lcols[i - 1] * 3;
        if (isf) {
            float* up = (float*)malloc(n * sizeof(float));
            orc_pyr_up_f32(b->lap_f[i], b->lrows[i], b->lcols[i], 3, up);
            for (size_t k = 0; k < n; ++k) b->lap_f[i - 1][k] = up[k] + b->lap_f[i - 1][k];
            free(up);
        } else {
            int16_t* up = (int16_t*)malloc(n * sizeof(int16_t));
            orc_pyr_up_s16(b->lap_s[i], b->lrows[i], b->lcols[i], 3, up);
            for (size_t k = 0; k < n; ++k) b->lap_s[i - 1][k] = sat_s16((int)up[k] + (int)b->lap_s[i - 1][k]);
            free(up);
        }
    }
    /* crop to dst_roi_final_, dst_mask = w0 > WEIGHT_EPS, Blender::blend: dst.setTo(0, mask == 0) */
    for (int y = 0; y < b->fh; ++y)
        for (int x = 0; x < b->fw; ++x) {
            size_t si = (size_t)y * b->lcols[0] + x, di = (size_t)y * b->fw + x;
            int on = b->wgt[0][si] > WEIGHT_EPS;
            if (dst_mask) dst_mask[di] = on ? 255 : 0;
            for (int c = 0; c < 3; ++c) {
                if (isf) {
                    float v = on ? b->lap_f[0][si * 3 + c] : 0.f;
                    if (dst_is_f32) ((float*)dst)[di * 3 + c] = v;
                    else ((int16_t*)dst)[di * 3 + c] = sat_s16_f(v);
                } else {
                    int16_t v = on ? b->lap_s[0][si * 3 + c] : 0;
                    if (dst_is_f32) ((float*)dst)[di * 3 + c] = (float)v;
                    else ((int16_t*)dst)[di * 3 + c] = v;
                }
            }
        }
    mb_release(b); /* blend() releases dst_pyr_laplace_ / dst_band_weights_ */
}

/* ------------------------------------------------------------------------------------------ */
/* A13 in-tree linear-ramp pair blend  B:141-717                                              */
/* ------------------------------------------------------------------------------------------ */
/* Restated with the reference's arithmetic.  Where the reference indexes out of bounds
 * (image rows past images1.rows in the cost loop B:216-221 when the tiles have different
 * heights; seam walking off the cost map) the restatement skips the row / clamps the column;
 * all in-bounds behaviour is literal. */
void orc_blend_pair_linear_size(int rows1, int cols1, int rows2, int cols2,
                                int tl1x, int tl1y, int tl2x, int tl2y, int* pr, int* pc) {
    (void)cols1;
    *pc = tl2x - tl1x + cols2;                                                    /* B:152 */
    int a = tl1y + rows1 > tl2y + rows2 ? tl1y + rows1 : tl2y + rows2;
    int m = tl1y < tl2y ? tl1y : tl2y;
    *pr = a - m;                                                                  /* B:153 */
}

static inline float sqrf(float v) { return v * v; }

/* costV of B:207-261 alone (interSectHe_ x (interSectBr_ + 2) floats, what B:265 writes to costV.bmp): test infrastructure for the
 * comparison with the reference's committed bitmap */
static float* g_costv_out = NULL;
int orc_blend_pair_linear(const float* img1, int rows1, int cols1, const float* img2, int rows2, int cols2,
                          int tl1x, int tl1y, int tl2x, int tl2y, float* pano, int* seam_out);
int orc_pair_linear_costv(const float* img1, int rows1, int cols1, const float* img2, int rows2, int cols2,
                          int tl1x, int tl1y, int tl2x, int tl2y, float* pano_scratch, float* costv_out) {
    g_costv_out = costv_out;
    int rc = orc_blend_pair_linear(img1, rows1, cols1, img2, rows2, cols2, tl1x, tl1y, tl2x, tl2y, pano_scratch, NULL);
    g_costv_out = NULL;
    return rc;
}

int orc_blend_pair_linear(const float* img1, int rows1, int cols1,
                          const float* img2, int rows2, int cols2,
                          int tl1x, int tl1y, int tl2x, int tl2y, float* pano, int* seam_out) {
    int panoBr_, panoHe_;
    orc_blend_pair_linear_size(rows1, cols1, rows2, cols2, tl1x, tl1y, tl2x, tl2y, &panoHe_, &panoBr_);
    int dx2 = tl2x - tl1x;                                                        /* B:158 */
    int dy = tl2y - tl1y, dy1 = 0, dy2 = 0;                                       /* B:159-172 */
    if (dy > 0) dy2 = dy;
    if (dy < 0) dy1 = -dy;
    int itlx = tl1x > tl2x ? tl1x : tl2x, itly = tl1y > tl2y ? tl1y : tl2y;      /* B:175 */
    int ibrx = tl1x + cols1 < tl2x + cols2 ? tl1x + cols1 : tl2x + cols2;         /* B:177-178 */
    int ibry = tl1y + rows1 < tl2y + rows2 ? tl1y + rows1 : tl2y + rows2;
    if (itlx >= ibrx || itly >= ibry) return 1;                                   /* B:182-183 */
    int height = ibry - itly, width = ibrx - itlx;                                /* B:185-186 */
    int iBr = cols1 - dx2;                                                        /* B:191 interSectBr_ */
    int iHe = panoHe_;                                                            /* B:192 */
    if (dx2 < 0 || iBr != width || iBr < 3 || iBr > cols2) return 2;              /* geometry the demo assumes */
    memset(pano, 0, (size_t)panoHe_ * panoBr_ * 3 * sizeof(float));

    /* costV  B:207-261 */
    int cw = iBr + 2;
    float* costV = (float*)calloc((size_t)iHe * cw, sizeof(float));
    int y0, y1, off;
    if (dy > 0) { y0 = dy2; y1 = iHe - dy2; off = dy2; }
    else if (dy < 0) { y0 = dy1; y1 = iHe - dy1; off = dy1; }
    else { y0 = 0; y1 = rows1 < rows2 ? rows1 : rows2; off = 0; }
    for (int y = y0; y < y1; ++y) {
        if (y >= rows1 || y - off < 0 || y - off >= rows2) continue;  /* reference reads out of bounds here */
        const float* p1 = img1 + (size_t)y * cols1 * 3;
        const float* p2 = img2 + (size_t)(y - off) * cols2 * 3;
        float* p3 = costV + (size_t)y * cw;
        for (int x = 1; x < iBr - 1; ++x)
            p3[x] = ((sqrf(p1[(x + dx2) * 3] - p2[x * 3]) + sqrf(p1[(x + dx2) * 3 + 1] - p2[x * 3 + 1]) + sqrf(p1[(x + dx2) * 3 + 2] - p2[x * 3 + 2])) +
                     (sqrf(p1[(x + dx2 + 1) * 3] - p2[(x - 1) * 3]) + sqrf(p1[(x + dx2 + 1) * 3 + 1] - p2[(x - 1) * 3 + 1]) + sqrf(p1[(x + dx2 + 1) * 3 + 2] - p2[(x - 1) * 3 + 2]))) / 2;
    }
    if (g_costv_out) memcpy(g_costv_out, costV, (size_t)iHe * cw * sizeof(float));
    /* greedy seam  B:268-307 */
    int* seam = (int*)malloc((size_t)iHe * sizeof(int));
    int px = iBr / 2, py = 0;
    seam[0] = px;
    while (py < iHe - 1) {
        const float* p = costV + (size_t)(py + 1) * cw;
        int xl = px - 1 < 0 ? 0 : px - 1, xr = px + 1 > cw - 1 ? cw - 1 : px + 1; /* clamp (OOB in reference) */
        float a = p[xl], bb = p[px], c = p[xr];
        if (a == bb && a == c) { }
        else if (a <= bb && a <= c) px = xl;
        else if (bb <= a && bb <= c) { }
        else if (c <= a && c <= bb) px = xr;
        py += 1;
        seam[py] = px;
    }
    if (seam_out) memcpy(seam_out, seam, (size_t)iHe * sizeof(int));

    /* gray  B:313-314: cvtColor(CV_RGB2GRAY) on CV_32FC3 = c0*0.299f + c1*0.587f + c2*0.114f */
    float* ga = (float*)malloc((size_t)rows1 * cols1 * sizeof(float));
    float* gb = (float*)malloc((size_t)rows2 * cols2 * sizeof(float));
    for (size_t i = 0; i < (size_t)rows1 * cols1; ++i) ga[i] = img1[i * 3] * 0.299f + img1[i * 3 + 1] * 0.587f + img1[i * 3 + 2] * 0.114f;
    for (size_t i = 0; i < (size_t)rows2 * cols2; ++i) gb[i] = img2[i * 3] * 0.299f + img2[i * 3 + 1] * 0.587f + img2[i * 3 + 2] * 0.114f;

    /* overlap classification  B:329-470 */
    int mw = width + 2;
    float* m1 = (float*)calloc((size_t)height * mw, sizeof(float));
    float* m2 = (float*)calloc((size_t)height * mw, sizeof(float));
    float thr = dy == 0 ? 10.f : 20.f;
    for (int y = 0; y < height; ++y) {
        const float* p1 = ga + (size_t)(dy > 0 ? y + dy2 : y) * cols1;
        const float* p2 = gb + (size_t)(dy < 0 ? y + dy1 : y) * cols2;
        float* p3 = m1 + (size_t)y * mw; float* p4 = m2 + (size_t)y * mw;
        p3[0] = 128; p4[0] = 128; p3[width + 1] = 128; p4[width + 1] = 128;
        for (int x = 1; x < width + 1; ++x) {
            float a = p1[x + dx2 - 1], bq = p2[x - 1];
            if (a >= thr && bq >= thr) { p3[x] = 255; p4[x] = 255; }
            if (a >= thr && bq < thr) { p3[x] = 1; p4[x] = 0; }
            if (a < thr && bq >= thr) { p3[x] = 0; p4[x] = 1; }
            if (a < thr && bq < thr) { p3[x] = 1; p4[x] = 1; }
        }
    }
    /* per-row scan + ramp weights  B:483-558 */
    for (int y = 0; y < height; ++y) {
        float* p1 = m1 + (size_t)y * mw; float* p2 = m2 + (size_t)y * mw;
        int left = 0, right = 0;
        for (int x = 1; x < width + 1; ++x) {
            if (p2[x] == 255 && p2[x - 1] == 0 && p2[x + 1] == 1) left = x;
            if ((p2[x] == 255 && p2[x - 1] == 0 && p2[x + 1] == 255) ||
                (p2[x - 1] == 128 && p2[x] == 255 && p2[x + 1] == 255 &&
                 (x + 2 < mw ? p2[x + 2] : 0.f) == 255 && (x + 3 < mw ? p2[x + 3] : 0.f) == 255)) left = x;
        }
        for (int x = 1; x < width + 1; ++x) {
            if (p2[x - 1] == 0 && p2[x] == 255 && p2[x + 1] == 1) right = x;
            if (p2[x - 1] == 255 && p2[x] == 255 && (p2[x + 1] == 1 || p2[x + 1] == 128)) right = x;
        }
        int sx = seam[y + dy2 + dy1];
        for (int x = 1; x < width + 1; ++x) {
            if (p2[x] == 255) {
                if (left && left == right) { p1[x] = 1; p2[x] = 0; }
                else if (x <= sx + 1) {
                    p1[x] = (float)(1 - 0.5 * (x - left) / (sx + 1 - left));      /* B:542 double expr */
                    p2[x] = 1 - p1[x];
                } else if (x > sx + 1 && x <= right) {
                    p1[x] = (float)(0.5 * (right - x) / (right - sx - 1));        /* B:548 */
                    p2[x] = 1 - p1[x];
                }
            }
        }
    }
    /* leftovers  B:560-572 */
    for (int y = 0; y < height; ++y) {
        float* p1 = m1 + (size_t)y * mw; float* p2 = m2 + (size_t)y * mw;
        for (int x = 0; x < width + 1; ++x) if (p1[x] == 255) { p1[x] = 1; p2[x] = 0; }
    }
    /* compose  B:579-711 */
    for (int y = 0; y < rows1; ++y) {                       /* image 1 only: x < dx2 */
        const float* p1 = img1 + (size_t)y * cols1 * 3;
        float* p2 = pano + (size_t)(y + dy1) * panoBr_ * 3;
        for (int x = 0; x < dx2 * 3; ++x) p2[x] = p1[x];
    }
    for (int y = (dy > 0 ? dy2 : 0); y < rows2; ++y) {      /* image 2 only: x >= images1.cols */
        const float* p1 = img2 + (size_t)(dy > 0 ? y - dy2 : y) * cols2 * 3;
        float* p2 = pano + (size_t)y * panoBr_ * 3;
        for (int x = cols1; x < panoBr_; ++x)
            for (int c = 0; c < 3; ++c) p2[3 * x + c] = p1[3 * (x - dx2) + c];
    }
    for (int y = 0; y < height; ++y) {                      /* overlap */
        const float* p1 = img1 + (size_t)(dy > 0 ? y + dy2 : y) * cols1 * 3;
        const float* p2 = img2 + (size_t)(dy < 0 ? y + dy1 : y) * cols2 * 3;
        const float* q1 = m1 + (size_t)y * mw; const float* q2 = m2 + (size_t)y * mw;
        float* p3 = pano + (size_t)(y + dy2 + dy1) * panoBr_ * 3;
        /* NB for dy<0 the reference writes pano.ptr(y + dy1), for dy>0 pano.ptr(y + dy2) */
        for (int x = dx2; x < dx2 + width; ++x)
            for (int c = 0; c < 3; ++c)
                p3[3 * x + c] = p1[3 * x + c] * q1[x - dx2 + 1] + p2[3 * (x - dx2) + c] * q2[x - dx2 + 1];
    }
    free(costV); free(seam); free(ga); free(gb); free(m1); free(m2);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* N3  mask preparation  W:286-301: dilate(masks_seam, MORPH_RECT 20x20) then & masks_warped     */
/* ------------------------------------------------------------------------------------------ */
/* cv::dilate with a MORPH_RECT kernel, default anchor (-1,-1) = (kw/2, kh/2), BORDER_CONSTANT with
 * morphologyDefaultBorderValue (outside pixels never win the max): out(x,y) = max src over
 * [x - kw/2, x - kw/2 + kw) x [y - kh/2, y - kh/2 + kh) clipped to the image.  (OpenCV 3.4.2
 * imgproc/src/morph.cpp; source absent: parity unpinned, known-answer tests only.) */
void orc_dilate_rect_u8(const uint8_t* src, int h, int w, int kw, int kh, uint8_t* dst) {
    int ax = kw / 2, ay = kh / 2;
    uint8_t* tmp = (uint8_t*)malloc((size_t)h * w);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int lo = x - ax < 0 ? 0 : x - ax, hi = x - ax + kw > w ? w : x - ax + kw, m = 0;
            for (int k = lo; k < hi; ++k) if (src[(size_t)y * w + k] > m) m = src[(size_t)y * w + k];
            tmp[(size_t)y * w + x] = (uint8_t)m;
        }
    for (int y = 0; y < h; ++y) {
        int lo = y - ay < 0 ? 0 : y - ay, hi = y - ay + kh > h ? h : y - ay + kh;
        for (int x = 0; x < w; ++x) {
            int m = 0;
            for (int k = lo; k < hi; ++k) if (tmp[(size_t)k * w + x] > m) m = tmp[(size_t)k * w + x];
            dst[(size_t)y * w + x] = (uint8_t)m;
        }
    }
    free(tmp);
}

/* N3  GainCompensator::apply  W:241-244: multiply(image, gains_(index, 0), image).  OpenCV 3.4.2 core/src/arithm.cpp
 * arithm_op with a scalar second operand: for mul/div depth2 = CV_64F, so wtype = CV_64F — the bytes are converted
 * to double, multiplied by the gain (mul64f, scale == 1) and stored with saturate_cast<uchar>(double) =
 * clamp(cvRound(v)), cvRound(double) = cvtsd2si: round-half-even, NaN / out of int range -> INT_MIN (-> 0).
 * (Source absent: parity unpinned, known-answer tests only.) */
void orc_gain_apply_u8(uint8_t* img, size_t n, double gain) {
    for (size_t i = 0; i < n; ++i) {
        double v = (double)img[i] * gain;
        double t = nearbyint(v);                       /* default rounding mode: to nearest even */
        int iv = (t >= -2147483648.0 && t <= 2147483647.0) ? (int)t : INT_MIN;
        img[i] = (uint8_t)((unsigned)iv <= 255u ? iv : (iv > 0 ? 255 : 0));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* N2  FeatherBlender  W:278-281,302,313 (OpenCV 3.4.2 stitching/src/blenders.cpp, absent)       */
/* ------------------------------------------------------------------------------------------ */
/* distanceTransform(mask, CV_32F, DIST_L1, 3) = distanceTransform_3x3 with metrics {1, 2} in 16.16 fixed
 * point (imgproc/src/distransform.cpp): a 1-pixel border of INIT_DIST0, forward + backward chamfer pass. */
void orc_distance_transform_l1(const uint8_t* src, int h, int w, float* dst) {
    const unsigned HV = 1u << 16, DIAG = 2u << 16, INIT = (unsigned)(INT_MAX >> 2);
    const float scale = 1.f / 65536.f;
    int step = w + 2;
    unsigned* t = (unsigned*)malloc((size_t)(h + 2) * step * sizeof(unsigned));
    for (size_t i = 0; i < (size_t)(h + 2) * step; ++i) t[i] = INIT;
    for (int i = 0; i < h; ++i) {
        unsigned* tmp = t + (size_t)(i + 1) * step + 1;
        for (int j = 0; j < w; ++j) {
            if (!src[(size_t)i * w + j]) tmp[j] = 0;
            else {
                unsigned t0 = tmp[j - step - 1] + DIAG, v = tmp[j - step] + HV;
                if (t0 > v) t0 = v;
                v = tmp[j - step + 1] + DIAG; if (t0 > v) t0 = v;
                v = tmp[j - 1] + HV; if (t0 > v) t0 = v;
                tmp[j] = t0;
            }
        }
    }
    for (int i = h - 1; i >= 0; --i) {
        unsigned* tmp = t + (size_t)(i + 1) * step + 1;
        for (int j = w - 1; j >= 0; --j) {
            unsigned t0 = tmp[j];
            if (t0 > HV) {
                unsigned v = tmp[j + step + 1] + DIAG; if (t0 > v) t0 = v;
                v = tmp[j + step] + HV; if (t0 > v) t0 = v;
                v = tmp[j + step - 1] + DIAG; if (t0 > v) t0 = v;
                v = tmp[j + 1] + HV; if (t0 > v) t0 = v;
                tmp[j] = t0;
            }
            dst[(size_t)i * w + j] = (float)t0 * scale;
        }
    }
    free(t);
}

/* createWeightMap: distanceTransform, multiply(weight, sharpness), threshold(1.f, THRESH_TRUNC) */
void orc_feather_weight_map(const uint8_t* mask, int h, int w, float sharpness, float* weight) {
    orc_distance_transform_l1(mask, h, w, weight);
    for (size_t i = 0; i < (size_t)h * w; ++i) {
        float v = weight[i] * sharpness;
        weight[i] = v > 1.f ? 1.f : v;
    }
}

struct orc_fb { float sharpness; int rx, ry, rw, rh; int16_t* dst; float* wgt; };
orc_fb* orc_fb_create(float sharpness) { orc_fb* b = (orc_fb*)calloc(1, sizeof(orc_fb)); b->sharpness = sharpness; return b; }
void orc_fb_destroy(orc_fb* b) { if (b) { free(b->dst); free(b->wgt); free(b); } }
void orc_fb_prepare(orc_fb* b, int n, const int* c, const int* s) {
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {
        if (c[2 * i] < tlx) tlx = c[2 * i];
        if (c[2 * i + 1] < tly) tly = c[2 * i + 1];
        if (c[2 * i] + s[2 * i] > brx) brx = c[2 * i] + s[2 * i];
        if (c[2 * i + 1] + s[2 * i + 1] > bry) bry = c[2 * i + 1] + s[2 * i + 1];
    }
    free(b->dst); free(b->wgt);
    b->rx = tlx; b->ry = tly; b->rw = brx - tlx; b->rh = bry - tly;
    b->dst = (int16_t*)calloc((size_t)b->rw * b->rh * 3, sizeof(int16_t));
    b->wgt = (float*)calloc((size_t)b->rw * b->rh, sizeof(float));
}
void orc_fb_result_size(const orc_fb* b, int* w, int* h) { *w = b->rw; *h = b->rh; }
void orc_fb_feed(orc_fb* b, const int16_t* img, const uint8_t* mask, int rows, int cols, int tl_x, int tl_y) {
    float* wm = (float*)malloc((size_t)rows * cols * sizeof(float));
    orc_feather_weight_map(mask, rows, cols, b->sharpness, wm);
    int dx = tl_x - b->rx, dy = tl_y - b->ry;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            size_t di = (size_t)(dy + y) * b->rw + dx + x, si = (size_t)y * cols + x;
            float w = wm[si];
            for (int c = 0; c < 3; ++c)
                b->dst[di * 3 + c] = (int16_t)(b->dst[di * 3 + c] + f2s_trunc((float)img[si * 3 + c] * w));
            b->wgt[di] = b->wgt[di] + w;
        }
    free(wm);
}
void orc_fb_blend(orc_fb* b, int16_t* dst, uint8_t* dst_mask) {
    size_t n = (size_t)b->rw * b->rh;
    for (size_t k = 0; k < n; ++k) {
        float d = b->wgt[k] + WEIGHT_EPS;
        int on = b->wgt[k] > WEIGHT_EPS;
        if (dst_mask) dst_mask[k] = on ? 255 : 0;
        for (int c = 0; c < 3; ++c) {
            int16_t v = f2s_trunc((float)b->dst[k * 3 + c] / d);   /* normalizeUsingWeightMap */
            dst[k * 3 + c] = on ? v : 0;                            /* Blender::blend: setTo(0, mask == 0) */
        }
    }
}

/* ---- Blender::NO: cv::detail::Blender itself, what Blender::createDefault(Blender::NO, false) returns (W:276).
 * OpenCV 3.4.2 modules/stitching/src/blenders.cpp (absent here; restated from the published source):
 *   prepare(Rect): dst_.create(size, CV_16SC3); dst_.setTo(0); dst_mask_.create(size, CV_8U); dst_mask_.setTo(0)
 *   feed(img CV_16SC3, mask CV_8U, tl): for every pixel: if (mask) dst_(dy + y, dx + x) = img(y, x); dst_mask_(dy + y, dx + x) |= mask
 *   blend(dst, dst_mask): dst_.setTo(0, dst_mask_ == 0); dst = dst_; dst_mask = dst_mask_                                   */
struct orc_nb { int rx, ry, rw, rh; int16_t* dst; uint8_t* mask; };
orc_nb* orc_nb_create(void) { return (orc_nb*)calloc(1, sizeof(orc_nb)); }
void orc_nb_destroy(orc_nb* b) { if (b) { free(b->dst); free(b->mask); free(b); } }
void orc_nb_prepare(orc_nb* b, int n, const int* c, const int* s) {
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {      /* resultRoi(corners, sizes) */
        if (c[2 * i] < tlx) tlx = c[2 * i];
        if (c[2 * i + 1] < tly) tly = c[2 * i + 1];
        if (c[2 * i] + s[2 * i] > brx) brx = c[2 * i] + s[2 * i];
        if (c[2 * i + 1] + s[2 * i + 1] > bry) bry = c[2 * i + 1] + s[2 * i + 1];
    }
    free(b->dst); free(b->mask);
    b->rx = tlx; b->ry = tly; b->rw = brx - tlx; b->rh = bry - tly;
    b->dst = (int16_t*)calloc((size_t)b->rw * b->rh * 3, sizeof(int16_t));
    b->mask = (uint8_t*)calloc((size_t)b->rw * b->rh, 1);
}
void orc_nb_result_size(const orc_nb* b, int* w, int* h) { *w = b->rw; *h = b->rh; }
void orc_nb_feed(orc_nb* b, const int16_t* img, const uint8_t* mask, int rows, int cols, int tl_x, int tl_y) {
    int dx = tl_x - b->rx, dy = tl_y - b->ry;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            size_t di = (size_t)(dy + y) * b->rw + dx + x, si = (size_t)y * cols + x;
            if (mask[si]) for (int c = 0; c < 3; ++c) b->dst[di * 3 + c] = img[si * 3 + c];
            b->mask[di] |= mask[si];
        }
}
void orc_nb_blend(orc_nb* b, int16_t* dst, uint8_t* dst_mask) {
    size_t n = (size_t)b->rw * b->rh;
    for (size_t k = 0; k < n; ++k) {
        for (int c = 0; c < 3; ++c) dst[k * 3 + c] = b->mask[k] ? b->dst[k * 3 + c] : 0;
        if (dst_mask) dst_mask[k] = b->mask[k];
    }
}

/* ---- A14: Mat::convertTo (alpha 1, beta 0) where it narrows: saturate_cast<short / uchar>(float) = clamp(cvRound(v)), W:294, W:315 */
void orc_convert_f32_s16(const float* src, size_t n, int16_t* dst) { for (size_t i = 0; i < n; ++i) dst[i] = sat_s16(orc_cvround(src[i])); }
void orc_convert_f32_u8(const float* src, size_t n, uint8_t* dst) {
    for (size_t i = 0; i < n; ++i) { int v = orc_cvround(src[i]); dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
}

/* ------------------------------------------------------------------------------------------ */
/* N1  DP seam finder, the data-parallel part: computeCosts S:733-803, estimateSeam S:806-957     */
/* ------------------------------------------------------------------------------------------ */
/* diffL2Square3<T> S:712-718: static_cast<float>(sqr(a0 - b0) + sqr(a1 - b1) + sqr(a2 - b2)); for uchar the
 * differences and squares are int, for float every operation is a float operation, left to right */
static float seam_diff(const void* i1, int cols1, int y1, int x1, const void* i2, int cols2, int y2, int x2, int is_u8) {
    if (is_u8) {
        const uint8_t* a = (const uint8_t*)i1 + ((size_t)y1 * cols1 + x1) * 3;
        const uint8_t* b = (const uint8_t*)i2 + ((size_t)y2 * cols2 + x2) * 3;
        int d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
        return (float)(d0 * d0 + d1 * d1 + d2 * d2);
    }
    const float* a = (const float*)i1 + ((size_t)y1 * cols1 + x1) * 3;
    const float* b = (const float*)i2 + ((size_t)y2 * cols2 + x2) * 3;
    float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    float s = d0 * d0;
    s = s + d1 * d1;
    s = s + d2 * d2;
    return s;
}
static int seam_label(const int32_t* labels, int uh, int uw, int y, int x) {
    return ((unsigned)y < (unsigned)uh && (unsigned)x < (unsigned)uw) ? labels[(size_t)y * uw + x] : 0;
}

void orc_seam_costs(const void* img1, int rows1, int cols1, const void* img2, int rows2, int cols2, int is_u8,
                    int tl1x, int tl1y, int tl2x, int tl2y, int utlx, int utly,
                    const int32_t* labels, int uh, int uw, int label, int rx, int ry, int rw, int rh,
                    float* costV, float* costH) {
    (void)rows1; (void)rows2;
    const int dx1 = utlx - tl1x, dy1 = utly - tl1y, dx2 = utlx - tl2x, dy2 = utly - tl2y;      /* S:750-751 */
    const float bad = 3.f * 255.f * 255.f;   /* normL2(Point3f(255,255,255), Point3f(0,0,0)) = (a-b).dot(a-b), S:754 */
    for (int y = ry; y < ry + rh; ++y)                                                            /* S:757-779 */
        for (int x = rx; x < rx + rw + 1; ++x) {
            float c = bad;
            if (seam_label(labels, uh, uw, y, x) == label && x > 0 && seam_label(labels, uh, uw, y, x - 1) == label)
                c = (seam_diff(img1, cols1, y + dy1, x + dx1 - 1, img2, cols2, y + dy2, x + dx2, is_u8) +
                     seam_diff(img1, cols1, y + dy1, x + dx1, img2, cols2, y + dy2, x + dx2 - 1, is_u8)) / 2;
            costV[(size_t)(y - ry) * (rw + 1) + (x - rx)] = c;
        }
    for (int y = ry; y < ry + rh + 1; ++y)                                                        /* S:784-802 */
        for (int x = rx; x < rx + rw; ++x) {
            float c = bad;
            if (seam_label(labels, uh, uw, y, x) == label && y > 0 && seam_label(labels, uh, uw, y - 1, x) == label)
                c = (seam_diff(img1, cols1, y + dy1 - 1, x + dx1, img2, cols2, y + dy2, x + dx2, is_u8) +
                     seam_diff(img1, cols1, y + dy1, x + dx1, img2, cols2, y + dy2 - 1, x + dx2, is_u8)) / 2;
            costH[(size_t)(y - ry) * rw + (x - rx)] = c;
        }
}

int orc_seam_estimate(const void* img1, int rows1, int cols1, const void* img2, int rows2, int cols2, int is_u8,
                      int tl1x, int tl1y, int tl2x, int tl2y, int utlx, int utly,
                      const int32_t* labels, int uh, int uw, int label, int rx, int ry, int rw, int rh,
                      int p1x, int p1y, int p2x, int p2y, int* seam_xy, int cap, int* is_horizontal) {
    if (is_horizontal) *is_horizontal = 0;
    if (p1x < rx || p1x >= rx + rw || p1y < ry || p1y >= ry + rh || p2x < rx || p2x >= rx + rw || p2y < ry || p2y >= ry + rh) return 0;   /* tips outside the component rectangle */
    float* costV = (float*)malloc(sizeof(float) * (size_t)rh * (rw + 1));
    float* costH = (float*)malloc(sizeof(float) * (size_t)(rh + 1) * rw);
    orc_seam_costs(img1, rows1, cols1, img2, rows2, cols2, is_u8, tl1x, tl1y, tl2x, tl2y, utlx, utly, labels, uh, uw, label, rx, ry, rw, rh,
                   costV, costH);
#define CV_(y, x) costV[(size_t)(y) * (rw + 1) + (x)]
#define CH_(y, x) costH[(size_t)(y) * rw + (x)]
    int sx = p1x - rx, sy = p1y - ry, dx = p2x - rx, dy = p2y - ry, swapped = 0;                 /* S:816-817 */
    int horiz = abs(dx - sx) > abs(dy - sy);                                                      /* S:827 */
    if (horiz ? sx > dx : sy > dy) { int t = sx; sx = dx; dx = t; t = sy; sy = dy; dy = t; swapped = 1; }   /* S:829-842 */
    uint8_t* control = (uint8_t*)calloc((size_t)rw * rh, 1);
    uint8_t* reach = (uint8_t*)calloc((size_t)rw * rh, 1);
    float* cost = (float*)calloc((size_t)rw * rh, sizeof(float));
#define AT(m, y, x) m[(size_t)(y) * rw + (x)]
    AT(reach, sy, sx) = 1; AT(cost, sy, sx) = 0.f;                                                /* S:850-851 */
    if (horiz) {                                                                                  /* S:858-886 */
        for (int x = sx + 1; x <= dx; ++x)
            for (int y = 0; y < rh; ++y) {
                int n = 0; float sc[3]; int sd[3];
                if (seam_label(labels, uh, uw, y + ry, x + rx) == label) {
                    if (AT(reach, y, x - 1)) { sc[n] = AT(cost, y, x - 1) + CH_(y, x - 1); sd[n++] = 1; }
                    if (y > 0 && AT(reach, y - 1, x - 1)) { sc[n] = AT(cost, y - 1, x - 1) + CH_(y - 1, x - 1) + CV_(y - 1, x); sd[n++] = 2; }
                    if (y < rh - 1 && AT(reach, y + 1, x - 1)) { sc[n] = AT(cost, y + 1, x - 1) + CH_(y + 1, x - 1) + CV_(y, x); sd[n++] = 3; }
                }
                if (n) {   /* min_element over pair<float, int>: first minimum, ties broken by the smaller step code */
                    int b = 0;
                    for (int k = 1; k < n; ++k) if (sc[k] < sc[b]) b = k;
                    AT(cost, y, x) = sc[b]; AT(control, y, x) = (uint8_t)sd[b]; AT(reach, y, x) = 255;
                }
            }
    } else {                                                                                      /* S:888-916 */
        for (int y = sy + 1; y <= dy; ++y)
            for (int x = 0; x < rw; ++x) {
                int n = 0; float sc[3]; int sd[3];
                if (seam_label(labels, uh, uw, y + ry, x + rx) == label) {
                    if (AT(reach, y - 1, x)) { sc[n] = AT(cost, y - 1, x) + CV_(y - 1, x); sd[n++] = 1; }
                    if (x > 0 && AT(reach, y - 1, x - 1)) { sc[n] = AT(cost, y - 1, x - 1) + CV_(y - 1, x - 1) + CH_(y, x - 1); sd[n++] = 2; }
                    if (x < rw - 1 && AT(reach, y - 1, x + 1)) { sc[n] = AT(cost, y - 1, x + 1) + CV_(y - 1, x + 1) + CH_(y, x); sd[n++] = 3; }
                }
                if (n) {
                    int b = 0;
                    for (int k = 1; k < n; ++k) if (sc[k] < sc[b]) b = k;
                    AT(cost, y, x) = sc[b]; AT(control, y, x) = (uint8_t)sd[b]; AT(reach, y, x) = 255;
                }
            }
    }
    int len = 0;
    if (AT(reach, dy, dx)) {                                                                      /* S:918-953 */
        int px = dx, py = dy;
        int* tmp = (int*)malloc(sizeof(int) * 2 * (size_t)(rw + rh + 2));
        tmp[0] = px + rx; tmp[1] = py + ry; len = 1;
        if (horiz) {
            while (px != sx) {
                int c = AT(control, py, px);
                if (c == 2) py--; else if (c == 3) py++;
                px--;
                tmp[2 * len] = px + rx; tmp[2 * len + 1] = py + ry; ++len;
            }
        } else {
            while (py != sy) {
                int c = AT(control, py, px);
                if (c == 2) px--; else if (c == 3) px++;
                py--;
                tmp[2 * len] = px + rx; tmp[2 * len + 1] = py + ry; ++len;
            }
        }
        for (int i = 0; i < len && i < cap; ++i) {
            int j = swapped ? i : len - 1 - i;                                                    /* S:947-948: reverse unless swapped */
            seam_xy[2 * i] = tmp[2 * j]; seam_xy[2 * i + 1] = tmp[2 * j + 1];
        }
        free(tmp);
    }
    if (is_horizontal) *is_horizontal = horiz;
    free(costV); free(costH); free(control); free(reach); free(cost);
#undef CV_
#undef CH_
#undef AT
    return len;
}
