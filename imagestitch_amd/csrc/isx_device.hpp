// isx_device.hpp — device-side helpers shared by the gfx950 kernels.
// Compiled with -ffp-contract=off: every fp32 expression below is evaluated exactly as written
// (mul, then add), because the CPU reference code these kernels must match bit-for-bit is built
// without FMA contraction.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

namespace isxd {

constexpr int WAVE = 64;

// cv::borderInterpolate, BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba) and BORDER_REFLECT
// (fedcba|abcdefgh|hgfedcb).  Loop form: any p is legal, n == 1 returns 0.
// One reflection is done branch-free (the common case: borders narrower than the image); the loop
// for repeated reflections (tiny images) is kept out of line so that it does not bloat hot kernels.
__device__ __noinline__ int reflect_loop(int p, int n, int delta) {
    if (n == 1) return 0;
    do {
        if (p < 0) p = -p - 1 + delta;
        else p = 2 * n - 1 - p - delta;
    } while ((unsigned)p >= (unsigned)n);
    return p;
}
__device__ __forceinline__ int reflect101(int p, int n) {
    int q = p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p);
    if ((unsigned)q >= (unsigned)n) q = reflect_loop(p, n, 1);
    return q;
}
__device__ __forceinline__ int reflect(int p, int n) {
    int q = p < 0 ? -p - 1 : (p >= n ? 2 * n - 1 - p : p);
    if ((unsigned)q >= (unsigned)n) q = reflect_loop(p, n, 0);
    return q;
}
// generic border: returns -1 for BORDER_CONSTANT outside
__device__ __forceinline__ int border_index(int p, int n, int border) {
    if ((unsigned)p < (unsigned)n) return p;
    switch (border) {
        case 1: return p < 0 ? 0 : n - 1;      // REPLICATE
        case 2: return reflect(p, n);          // REFLECT
        case 4: return reflect101(p, n);       // REFLECT_101
        case 3: {                              // WRAP
            int q = p % n;
            return q < 0 ? q + n : q;
        }
        default: return -1;                    // CONSTANT
    }
}

// x86 conversions the reference code runs (v_cvt_* on the GPU saturates, cvt(t)ss2si does not):
//   cvRound          = cvtss2si : round-half-even; NaN or |v| >= 2^31 -> 0x80000000
//   static_cast<int> = cvttss2si: truncate;        NaN or |v| >= 2^31 -> 0x80000000
__device__ __forceinline__ int cvround_x86(float v) {
    return (fabsf(v) < 2147483648.0f) ? __float2int_rn(v) : INT_MIN;
}
__device__ __forceinline__ int f2i_x86(float v) {
    return (fabsf(v) < 2147483648.0f) ? (int)v : INT_MIN;
}
// static_cast<short>(float): cvttss2si then the low 16 bits
__device__ __forceinline__ int f2s_x86(float v) { return (int)(short)(unsigned short)(unsigned)f2i_x86(v); }
__device__ __forceinline__ int sat_s16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int sat_u8(int v) { return min(max(v, 0), 255); }
__device__ __forceinline__ int wrap_s16(int v) { return (int)(short)(unsigned short)(unsigned)v; }

// f16 storage (RNE both ways; conversions are exact widening on load)
__device__ __forceinline__ unsigned short f2h_bits(float v) { return __half_as_ushort(__float2half_rn(v)); }
__device__ __forceinline__ float h2f_bits(unsigned short b) { return __half2float(__ushort_as_half(b)); }

}  // namespace isxd
