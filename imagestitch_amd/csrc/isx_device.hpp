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

// ---- packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two lanes per instruction, each half rounded on its own, so a packed
// expression gives the bits of its scalar form) ------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { f32x2 r = {v, v}; return r; }

// XCD-aware block order for 2-D stencil / gather grids.  MI355X hands the workgroups of a launch to its 8 XCDs round robin (block L of
// a 1-D launch runs on XCD L % 8), and each XCD has its own L2: in a plain row-major grid neighbouring blocks - which share halo rows,
// or the cache lines their row segments straddle - always sit on different XCDs, and every XCD fetches the shared lines from the
// fabric itself.  Here the blocks of one XCD walk groups of `grp` block rows column by column (vertical and horizontal neighbours are
// dispatched back to back on the same XCD and meet in its L2), and the groups are dealt to the XCDs round robin, which keeps their
// shares of the image even.  Launch xcd_grid_blocks(grp, gx, gy) blocks; false = a padding block.
// per = grp * gx blocks per group, magic = 2^32 / per + 1 (xcd_magic on the host): j / per without a division sequence per wave
__device__ __forceinline__ bool xcd_block(unsigned L, int grp, int gx, int gy, unsigned magic, int& bx, int& by) {
    const unsigned xcd = L & 7u, j = L >> 3, per = (unsigned)(grp * gx);
    const unsigned g = __umulhi(j, magic), r = j - g * per;          // exact while j * per < 2^32
    const unsigned c = grp == 2 ? r >> 1 : r / (unsigned)grp;
    bx = (int)c;
    by = (int)((g * 8u + xcd) * (unsigned)grp + (r - c * (unsigned)grp));
    return by < gy;
}

// The same walk with every XCD owning ONE contiguous band of `band` block rows (band = ceil(gy / 8)) instead of every 8th group: the
// groups an XCD works on one after the other are vertically adjacent, so the halo rows two groups share are still in that XCD's L2
// when the second group reads them (a group's working set is a fraction of the 4 MB).  Right when the work per block row is even
// from top to bottom (a pair's overlap is a vertical stripe).  Launch xcd_band_blocks(grp, gx, gy) blocks.
__device__ __forceinline__ bool xcd_band_block(unsigned L, int grp, int gx, int gy, int band, unsigned magic, int& bx, int& by) {
    const unsigned xcd = L & 7u, j = L >> 3, per = (unsigned)(grp * gx);
    const unsigned g = __umulhi(j, magic), r = j - g * per;
    const unsigned c = grp == 2 ? r >> 1 : r / (unsigned)grp;
    const unsigned rb = g * (unsigned)grp + (r - c * (unsigned)grp);      // block row inside the band
    bx = (int)c;
    by = (int)(xcd * (unsigned)band + rb);
    return (rb < (unsigned)band) & (by < gy);
}

// IEEE division a / z by the hardware's own recurrence, written out so that several numerators share one reciprocal and two of them
// ride in one packed FMA:  r1 = r0 + r0 (1 - z r0);  q0 = a r1;  q1 = q0 + r1 (a - z q0);  q = q1 + r1 (a - z q1)   with r0 = v_rcp_f32(z).
// This is what v_div_scale / v_rcp / v_fma x 5 / v_div_fmas / v_div_fixup compute whenever v_div_scale does not rescale: z and
// 1 / z normal, the quotient neither denormal nor near overflow, exponent(a) - exponent(z) < 96, |a| >= 2^-103 or a == 0 (then the
// quotient is a zero, possibly of the other sign).  Callers guarantee that range (isx_selftest_division checks the equality).
__device__ __forceinline__ f32x2 refine_rcp(f32x2 z, f32x2 r0) {
    const f32x2 e = pk_fma(-z, r0, splat2(1.f));
    return pk_fma(e, r0, r0);
}
__device__ __forceinline__ f32x2 div_by_refined(f32x2 a, f32x2 z, f32x2 r1) {
    f32x2 q = a * r1;
    f32x2 e = pk_fma(-z, q, a);
    q = pk_fma(e, r1, q);
    e = pk_fma(-z, q, a);
    return pk_fma(e, r1, q);
}

// f16 storage (RNE both ways; conversions are exact widening on load)
__device__ __forceinline__ unsigned short f2h_bits(float v) { return __half_as_ushort(__float2half_rn(v)); }
__device__ __forceinline__ float h2f_bits(unsigned short b) { return __half2float(__ushort_as_half(b)); }

}  // namespace isxd
