// roihost.cpp — detectResultRoi's border scan (W:64-88 where the extrema provably lie on the border; SphericalWarper's
// detectResultRoiByBorder) ranked ON THE HOST.  Plain C++ (g++), no HIP: called by warp.hip's border_scan_sync.
//
// Round 5's literal drop-in leg spent 155 of its 346 us of caller-thread time in four detectResultRoi round trips: one workgroup ranks the
// 2 (W + H) border pixels on the GPU (k_roi_border_pin) and the host polls for the answer - 25 us on an idle GPU, 40 - 52 us on a busy one,
// because the launch queues behind whatever the device is doing (profiles/round5_roi_latency.txt).  The ranking needs no transcendental and
// no image: 12 000 points x (nine multiply-adds, a division, a square root, a division) is microseconds of AVX2 / AVX-512 on the caller's thread, with
// no launch and nothing to wait for.  Same scheme as the kernel: two strictly monotone stand-ins
//     u = scale * atan2f(x_, z_)          ~  d = "diamond angle" of (x_, z_) in (-2, 2]
//     v = scale * y_ / sqrtf(x_^2 + z_^2) ~  q = y_ / sqrt(x_^2 + z_^2)              (spherical: w = y_ / |r|, NaN -> 0)
// rank the border pixels, every pixel within a tolerance of one of the four extrema is a candidate, and the caller evaluates mapForward with
// the host's own libm on exactly those (warp.hip: detect_roi) - min / max over a set that contains the true extremal pixels and only pixels
// of the scan is min / max over the scan.  The stand-ins here are Newton-refined reciprocal estimates (AVX2) or plain divisions (scalar fallback); the
// kernel's are v_rcp_f32 / v_rsq_f32: all a few ulp off at most, under the same tolerances, which cover that many times over.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

constexpr float BIG = 3.402823466e+38f;

struct Ext { float dmin, qmin, dmax, qmax; };

inline void proxy_scalar(const float* rk, bool sph, float x, float y, float& d, float& q) {
    const float x_ = rk[0] * x + rk[1] * y + rk[2];
    const float y_ = rk[3] * x + rk[4] * y + rk[5];
    const float z_ = rk[6] * x + rk[7] * y + rk[8];
    const float ax = std::fabs(x_), az = std::fabs(z_);
    const float t = ax / (ax + az);
    d = std::copysign(z_ >= 0.f ? t : 2.f - t, x_);
    if (sph) {
        const float w = y_ / std::sqrt(x_ * x_ + y_ * y_ + z_ * z_);
        q = (w == w) ? w : 0.f;
    } else q = y_ / std::sqrt(x_ * x_ + z_ * z_);
}

// one edge: n points (x0 + i dx, y0 + i dy), stand-ins to d[0..n), q[0..n), extrema folded into e (NaN never wins, as (std::min)(tl, u))
void edge_scalar(const float* rk, bool sph, float x0, float y0, float dx, float dy, int n, float* d, float* q, Ext& e) {
    for (int i = 0; i < n; ++i) {
        proxy_scalar(rk, sph, x0 + dx * (float)i, y0 + dy * (float)i, d[i], q[i]);
        e.dmin = (d[i] < e.dmin) ? d[i] : e.dmin; e.qmin = (q[i] < e.qmin) ? q[i] : e.qmin;
        e.dmax = (e.dmax < d[i]) ? d[i] : e.dmax; e.qmax = (e.qmax < q[i]) ? q[i] : e.qmax;
    }
}

#if defined(__x86_64__)
// 1 / a and 1 / sqrt(a) from the 12-bit estimates and one Newton step: relative error below 2^-22 (the kernel's v_rcp_f32 / v_rsq_f32 are 1 ulp;
// the tolerances below are 30 ulp and more), a third of the time of vdivps + vsqrtps + vdivps per eight points
__attribute__((target("avx2"))) inline __m256 rcp_nr(__m256 a) {
    const __m256 r = _mm256_rcp_ps(a);
    return _mm256_mul_ps(r, _mm256_sub_ps(_mm256_set1_ps(2.f), _mm256_mul_ps(a, r)));
}
__attribute__((target("avx2"))) inline __m256 rsqrt_nr(__m256 a) {
    const __m256 r = _mm256_rsqrt_ps(a);
    return _mm256_mul_ps(_mm256_mul_ps(_mm256_set1_ps(0.5f), r), _mm256_sub_ps(_mm256_set1_ps(3.f), _mm256_mul_ps(_mm256_mul_ps(a, r), r)));
}
__attribute__((target("avx2"))) void edge_avx2(const float* rk, bool sph, float x0, float y0, float dx, float dy, int n, float* d, float* q, Ext& e) {
    const __m256 iota = _mm256_setr_ps(0.f, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f);
    const __m256 sign = _mm256_set1_ps(-0.f), two = _mm256_set1_ps(2.f), zero = _mm256_setzero_ps();
    __m256 dmin = _mm256_set1_ps(BIG), qmin = dmin, dmax = _mm256_set1_ps(-BIG), qmax = dmax;
    // along an edge the three linear forms are base + i * step (stand-ins only: an ulp of difference to mapForward's own association of
    // W:38-40 is inside the tolerances, and the candidates are evaluated with that association by the caller)
    __m256 base[3], step[3];
    for (int j = 0; j < 3; ++j) {
        base[j] = _mm256_set1_ps(rk[3 * j] * x0 + rk[3 * j + 1] * y0 + rk[3 * j + 2]);
        step[j] = _mm256_set1_ps(rk[3 * j] * dx + rk[3 * j + 1] * dy);
    }
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 fi = _mm256_add_ps(_mm256_set1_ps((float)i), iota);       // exact: i < 2^24
        const __m256 x_ = _mm256_add_ps(base[0], _mm256_mul_ps(step[0], fi));
        const __m256 y_ = _mm256_add_ps(base[1], _mm256_mul_ps(step[1], fi));
        const __m256 z_ = _mm256_add_ps(base[2], _mm256_mul_ps(step[2], fi));
        const __m256 ax = _mm256_andnot_ps(sign, x_), az = _mm256_andnot_ps(sign, z_);
        const __m256 t = _mm256_mul_ps(ax, rcp_nr(_mm256_add_ps(ax, az)));
        const __m256 zneg = _mm256_cmp_ps(z_, zero, _CMP_NGE_UQ);               // !(z_ >= 0)
        const __m256 m = _mm256_blendv_ps(t, _mm256_sub_ps(two, t), zneg);
        const __m256 dd = _mm256_or_ps(_mm256_andnot_ps(sign, m), _mm256_and_ps(sign, x_));       // copysign(m, x_)
        __m256 qq;
        if (sph) {
            const __m256 r2 = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(x_, x_), _mm256_mul_ps(y_, y_)), _mm256_mul_ps(z_, z_));
            const __m256 w = _mm256_mul_ps(y_, rsqrt_nr(r2));
            qq = _mm256_and_ps(w, _mm256_cmp_ps(w, w, _CMP_EQ_OQ));             // NaN -> 0
        } else {
            const __m256 r2 = _mm256_add_ps(_mm256_mul_ps(x_, x_), _mm256_mul_ps(z_, z_));
            qq = _mm256_mul_ps(y_, rsqrt_nr(r2));
        }
        _mm256_storeu_ps(d + i, dd); _mm256_storeu_ps(q + i, qq);
        // min_ps / max_ps return their SECOND operand when either is NaN: a NaN stand-in never replaces the running extremum
        dmin = _mm256_min_ps(dd, dmin); qmin = _mm256_min_ps(qq, qmin); dmax = _mm256_max_ps(dd, dmax); qmax = _mm256_max_ps(qq, qmax);
    }
    float a[8], b[8], c[8], f[8];
    _mm256_storeu_ps(a, dmin); _mm256_storeu_ps(b, qmin); _mm256_storeu_ps(c, dmax); _mm256_storeu_ps(f, qmax);
    for (int j = 0; j < 8; ++j) {
        e.dmin = (a[j] < e.dmin) ? a[j] : e.dmin; e.qmin = (b[j] < e.qmin) ? b[j] : e.qmin;
        e.dmax = (e.dmax < c[j]) ? c[j] : e.dmax; e.qmax = (e.qmax < f[j]) ? f[j] : e.qmax;
    }
    if (i < n) edge_scalar(rk, sph, x0 + dx * (float)i, y0 + dy * (float)i, dx, dy, n - i, d + i, q + i, e);
}

// the same sixteen points at a time where the CPU has AVX-512F (the GPU boxes' EPYC 9575F does): vrcp14ps / vrsqrt14ps + one Newton step
__attribute__((target("avx512f"))) inline __m512 rcp_nr512(__m512 a) {
    const __m512 r = _mm512_rcp14_ps(a);
    return _mm512_mul_ps(r, _mm512_sub_ps(_mm512_set1_ps(2.f), _mm512_mul_ps(a, r)));
}
__attribute__((target("avx512f"))) inline __m512 rsqrt_nr512(__m512 a) {
    const __m512 r = _mm512_rsqrt14_ps(a);
    return _mm512_mul_ps(_mm512_mul_ps(_mm512_set1_ps(0.5f), r), _mm512_sub_ps(_mm512_set1_ps(3.f), _mm512_mul_ps(_mm512_mul_ps(a, r), r)));
}
__attribute__((target("avx512f"))) void edge_avx512(const float* rk, bool sph, float x0, float y0, float dx, float dy, int n, float* d, float* q, Ext& e) {
    const __m512 iota = _mm512_setr_ps(0.f, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f, 9.f, 10.f, 11.f, 12.f, 13.f, 14.f, 15.f);
    const __m512i sign = _mm512_set1_epi32((int)0x80000000u);
    const __m512 two = _mm512_set1_ps(2.f), zero = _mm512_setzero_ps();
    __m512 dmin = _mm512_set1_ps(BIG), qmin = dmin, dmax = _mm512_set1_ps(-BIG), qmax = dmax;
    __m512 base[3], step[3];
    for (int j = 0; j < 3; ++j) {
        base[j] = _mm512_set1_ps(rk[3 * j] * x0 + rk[3 * j + 1] * y0 + rk[3 * j + 2]);
        step[j] = _mm512_set1_ps(rk[3 * j] * dx + rk[3 * j + 1] * dy);
    }
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512 fi = _mm512_add_ps(_mm512_set1_ps((float)i), iota);
        const __m512 x_ = _mm512_add_ps(base[0], _mm512_mul_ps(step[0], fi));
        const __m512 y_ = _mm512_add_ps(base[1], _mm512_mul_ps(step[1], fi));
        const __m512 z_ = _mm512_add_ps(base[2], _mm512_mul_ps(step[2], fi));
        const __m512 ax = _mm512_castsi512_ps(_mm512_andnot_si512(sign, _mm512_castps_si512(x_)));
        const __m512 az = _mm512_castsi512_ps(_mm512_andnot_si512(sign, _mm512_castps_si512(z_)));
        const __m512 t = _mm512_mul_ps(ax, rcp_nr512(_mm512_add_ps(ax, az)));
        const __mmask16 zneg = _mm512_cmp_ps_mask(z_, zero, _CMP_NGE_UQ);              // !(z_ >= 0)
        const __m512 m = _mm512_mask_blend_ps(zneg, t, _mm512_sub_ps(two, t));
        const __m512 dd = _mm512_castsi512_ps(_mm512_or_si512(_mm512_andnot_si512(sign, _mm512_castps_si512(m)), _mm512_and_si512(sign, _mm512_castps_si512(x_))));
        __m512 qq;
        if (sph) {
            const __m512 r2 = _mm512_add_ps(_mm512_add_ps(_mm512_mul_ps(x_, x_), _mm512_mul_ps(y_, y_)), _mm512_mul_ps(z_, z_));
            const __m512 w = _mm512_mul_ps(y_, rsqrt_nr512(r2));
            qq = _mm512_maskz_mov_ps(_mm512_cmp_ps_mask(w, w, _CMP_EQ_OQ), w);         // NaN -> 0
        } else {
            const __m512 r2 = _mm512_add_ps(_mm512_mul_ps(x_, x_), _mm512_mul_ps(z_, z_));
            qq = _mm512_mul_ps(y_, rsqrt_nr512(r2));
        }
        _mm512_storeu_ps(d + i, dd); _mm512_storeu_ps(q + i, qq);
        // a NaN stand-in never replaces the running extremum: compare-and-blend in the scalar code's own order ((d < dmin) ? d : dmin)
        dmin = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(dd, dmin, _CMP_LT_OQ), dmin, dd);
        qmin = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(qq, qmin, _CMP_LT_OQ), qmin, qq);
        dmax = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(dmax, dd, _CMP_LT_OQ), dmax, dd);
        qmax = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(qmax, qq, _CMP_LT_OQ), qmax, qq);
    }
    float a[16], b[16], c[16], f[16];
    _mm512_storeu_ps(a, dmin); _mm512_storeu_ps(b, qmin); _mm512_storeu_ps(c, dmax); _mm512_storeu_ps(f, qmax);
    for (int j = 0; j < 16; ++j) {
        e.dmin = (a[j] < e.dmin) ? a[j] : e.dmin; e.qmin = (b[j] < e.qmin) ? b[j] : e.qmin;
        e.dmax = (e.dmax < c[j]) ? c[j] : e.dmax; e.qmax = (e.qmax < f[j]) ? f[j] : e.qmax;
    }
    if (i < n) edge_scalar(rk, sph, x0 + dx * (float)i, y0 + dy * (float)i, dx, dy, n - i, d + i, q + i, e);
}

// the second pass, eight stand-ins per test: groups that hold a candidate (a handful of 1 500) are handed to `take` point by point
template <class F>
__attribute__((target("avx2"))) int hits_avx2(const float* d, const float* q, int n, float d_lo, float d_hi, float q_lo, float q_hi, F&& take) {
    const __m256 vdl = _mm256_set1_ps(d_lo), vdh = _mm256_set1_ps(d_hi), vql = _mm256_set1_ps(q_lo), vqh = _mm256_set1_ps(q_hi);
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 dd = _mm256_loadu_ps(d + i), qq = _mm256_loadu_ps(q + i);
        const __m256 m = _mm256_or_ps(_mm256_or_ps(_mm256_cmp_ps(dd, vdl, _CMP_LE_OQ), _mm256_cmp_ps(dd, vdh, _CMP_GE_OQ)),
                                      _mm256_or_ps(_mm256_cmp_ps(qq, vql, _CMP_LE_OQ), _mm256_cmp_ps(qq, vqh, _CMP_GE_OQ)));
        if (_mm256_movemask_ps(m))
            for (int j = 0; j < 8; ++j) take(i + j);
    }
    return i;
}
#endif

}  // namespace

// The border pixels of an sw x sh source in the order of k_roi_border_pin's border_point (top, bottom, left, right; corners twice), ranked by
// their stand-ins; writes the (x, y) of every pixel within the tolerance of one of the four extrema to cand_xy (at most cap pairs) and returns
// their number (> cap: the list is incomplete, use another path), 0 when no stand-in is finite.  scratch: 2 * (2 sw + 2 sh) floats.
// isa: 0 = best available (AVX-512F, else AVX2, else scalar), 1 = scalar code, 2 = AVX2 where the CPU has it (tests compare them).
extern "C" int isx_roi_border_host(const float r_kinv[9], int spherical, int sw, int sh, int* cand_xy, int cap, float* scratch, int isa) {
    const int n = 2 * sw + 2 * sh;
    float* d = scratch;
    float* q = scratch + n;
    const bool sph = spherical != 0;
    Ext e = {BIG, BIG, -BIG, -BIG};
    using EdgeFn = void (*)(const float*, bool, float, float, float, float, int, float*, float*, Ext&);
    EdgeFn edge = edge_scalar;
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2"), have_avx512 = __builtin_cpu_supports("avx512f");
    if ((isa == 0 || isa == 2) && have_avx2) edge = edge_avx2;
    if (isa == 0 && have_avx512 && have_avx2) edge = edge_avx512;      // (isa 2: the AVX2 form on a CPU that has both - tests)
#endif
    edge(r_kinv, sph, 0.f, 0.f, 1.f, 0.f, sw, d, q, e);
    edge(r_kinv, sph, 0.f, (float)(sh - 1), 1.f, 0.f, sw, d + sw, q + sw, e);
    edge(r_kinv, sph, 0.f, 0.f, 0.f, 1.f, sh, d + 2 * sw, q + 2 * sw, e);
    edge(r_kinv, sph, (float)(sw - 1), 0.f, 0.f, 1.f, sh, d + 2 * sw + sh, q + 2 * sw + sh, e);
    if (!(e.dmin <= e.dmax) || !(e.qmin <= e.qmax)) return 0;      // nothing finite anywhere
    const float tol_d = 7.62939453125e-06f;                                        // as k_roi_candidates: 2^-17 of a (-2, 2] range
    const float tol_q = 4e-6f * std::fmax(std::fabs(e.qmin), std::fabs(e.qmax)) + 1e-9f;
    const float d_lo = e.dmin + tol_d, d_hi = e.dmax - tol_d, q_lo = e.qmin + tol_q, q_hi = e.qmax - tol_q;
    int cnt = 0;
    auto take = [&](int i) {
        if (d[i] <= d_lo || d[i] >= d_hi || q[i] <= q_lo || q[i] >= q_hi) {
            if (cnt < cap) {
                int x, y;
                if (i < sw) { x = i; y = 0; }
                else if (i < 2 * sw) { x = i - sw; y = sh - 1; }
                else if (i < 2 * sw + sh) { x = 0; y = i - 2 * sw; }
                else { x = sw - 1; y = i - 2 * sw - sh; }
                cand_xy[2 * cnt] = x; cand_xy[2 * cnt + 1] = y;
            }
            ++cnt;
        }
    };
    int i = 0;
#if defined(__x86_64__)
    if (edge == edge_avx2 || edge == edge_avx512) i = hits_avx2(d, q, n, d_lo, d_hi, q_lo, q_hi, take);
#endif
    for (; i < n; ++i) take(i);
    return cnt;
}
