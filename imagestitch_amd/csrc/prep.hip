// prep.hip — the cheap per-pixel stages either side of the blender on gfx950:
//   isx_mask_dilate_and  dilate(masks_seam[k], MORPH_RECT 20x20) & masks_warped[k]      W:286-301
//   isx_gain_apply       GainCompensator::apply = multiply(image, gain, image)          W:241-244
#include "isx_device.hpp"
#include "isx_internal.hpp"

#include <climits>

using namespace isx;
using namespace isxd;

namespace {

// N3  dilate(mask, MORPH_RECT kw x kh) [& other]  (W:286-301), anchor (kw/2, kh/2), pixels outside the image do
// not take part.  Structuring elements up to 33 x 33 (the reference uses 20 x 20): one fused kernel, a 64 x 128
// output tile per block — input tile + halo staged in LDS, row maxima into a second LDS plane, column maxima from
// there; every thread produces 4 adjacent outputs from one run of k + 3 bytes (the k - 3 bytes common to the four
// windows are reduced once).  Larger elements: the two generic kernels below through a temporary plane.
constexpr int DIL_MAXK = 33, DIL_TW = 64, DIL_TH = 128, DIL_NT = 512;
constexpr int DIL_AWD = (DIL_TW + DIL_MAXK - 1 + 3 + 3) / 4;   // dwords per staged row: 96 bytes + up to 3 of misalignment
constexpr int DIL_AH = DIL_TH + DIL_MAXK - 1;   // 160

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
// four bytes as two pairs of 16-bit lanes (even bytes, odd bytes): v_pk_max_u16 then is a 4-way byte maximum
struct B4 { unsigned e, o; };
__device__ __forceinline__ B4 b4_split(unsigned d) { B4 r; r.e = d & 0x00ff00ffu; r.o = (d >> 8) & 0x00ff00ffu; return r; }
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    us2_t x, y;
    __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    const us2_t z = __builtin_elementwise_max(x, y);
    unsigned r; __builtin_memcpy(&r, &z, 4);
    return r;
}
__device__ __forceinline__ B4 b4_max(B4 a, B4 b) { B4 r; r.e = pk_max_u16(a.e, b.e); r.o = pk_max_u16(a.o, b.o); return r; }
__device__ __forceinline__ unsigned b4_join(B4 a) { return a.e | (a.o << 8); }

__global__ __launch_bounds__(DIL_NT) void k_dilate_and(const unsigned char* src, size_t sstep, int rows, int cols, int kw, int kh,
                                                    const unsigned char* other, size_t ostep, unsigned char* dst, size_t dstep) {
    // A: input tile + halo; LDS dword (r, j) = the 4 pixels of image row Y0 - ay + r from column X0 - ax + 4 j on, built
    // from the two ALIGNED source dwords around them (v_alignbyte; 8 loads in flight per thread), outside pixels zeroed.
    __shared__ unsigned A[DIL_AH][DIL_AWD];
    __shared__ unsigned B[DIL_AH][DIL_TW / 4];     // row maxima, 4 columns per dword
    const int X0 = blockIdx.x * DIL_TW, Y0 = blockIdx.y * DIL_TH, ax = kw / 2, ay = kh / 2;
    const int th = min(DIL_TH, rows - Y0), ah = th + kh - 1;
    const uintptr_t base = (uintptr_t)src + (intptr_t)(X0 - ax);
    const unsigned* safe = (const unsigned*)((uintptr_t)src & ~(uintptr_t)3);
    for (int i0 = threadIdx.x; i0 < ah * DIL_AWD; i0 += 4 * DIL_NT) {
        unsigned lo[4], hi[4], keep[4], sh[4];
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * DIL_NT, ah * DIL_AWD - 1);
            const int r = i / DIL_AWD, j = i - r * DIL_AWD;
            const int gy = Y0 - ay + r;
            const uintptr_t p = base + (uintptr_t)((intptr_t)gy * (intptr_t)sstep) + (uintptr_t)(4 * j);   // address of the first pixel
            const int mis = (int)(p & 3);
            const int gx0 = X0 - ax + 4 * j;                      // its image column
            const bool rowok = (unsigned)gy < (unsigned)rows;
            // an aligned dword may be read when it holds at least one byte of row gy
            const bool any0 = rowok && gx0 - mis + 3 >= 0 && gx0 - mis < cols;
            const bool any1 = rowok && mis != 0 && gx0 - mis + 7 >= 0 && gx0 - mis + 4 < cols;
            unsigned m = 0xffffffffu;                             // bytes of the dword that are image pixels
            if (gx0 < 0) m = gx0 <= -4 ? 0u : m << (8 * -gx0);
            if (gx0 + 3 >= cols) m = gx0 >= cols ? 0u : m & (0xffffffffu >> (8 * (gx0 + 4 - cols)));
            keep[u] = rowok ? m : 0u;
            lo[u] = *(any0 ? (const unsigned*)(p - mis) : safe);
            hi[u] = *(any1 ? (const unsigned*)(p - mis + 4) : safe);
            sh[u] = (unsigned)mis;
            idx[u] = i;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) (&A[0][0])[idx[u]] = __builtin_amdgcn_alignbyte(hi[u], lo[u], sh[u]) & keep[u];
    }
    __syncthreads();
    // row pass: one task = 4 adjacent outputs (one dword) of one staged row.  With e(j) = (byte j, byte j + 2) as two 16-bit
    // lanes, the even output bytes are max e(j) over [0, kw) and the odd ones max e(j) over [1, kw + 1): one packed running
    // maximum over [1, kw) serves both.  e(4n) / e(4n+1) are the even / odd bytes of dword n, e(4n+2) / e(4n+3) those
    // shifted by one 16-bit lane into dword n + 1.
    for (int i = threadIdx.x; i < ah * (DIL_TW / 4); i += DIL_NT) {
        const int r = i / (DIL_TW / 4), g = i % (DIL_TW / 4);
        const unsigned* ar = &A[r][g];
        const B4 first = b4_split(ar[0]);
        B4 prev = first;
        unsigned common = 0, ekw = 0;
        for (int n = 0; 4 * n <= kw; ++n) {              // taps 4n .. 4n + 3 (uniform trip count)
            const B4 next = b4_split(ar[n + 1]);
            const unsigned t[4] = {prev.e, prev.o, __builtin_amdgcn_alignbyte(next.e, prev.e, 2), __builtin_amdgcn_alignbyte(next.o, prev.o, 2)};
            if (n > 0 && 4 * n + 3 < kw) {
                common = pk_max_u16(pk_max_u16(common, t[0]), pk_max_u16(t[1], pk_max_u16(t[2], t[3])));
            } else {
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) {
                    const int j = 4 * n + sft;
                    if (j >= 1 && j < kw) common = pk_max_u16(common, t[sft]);
                    if (j == kw) ekw = t[sft];
                }
            }
            prev = next;
        }
        B4 o;
        o.e = pk_max_u16(common, first.e);
        o.o = pk_max_u16(common, ekw);
        B[r][g] = b4_join(o);
    }
    __syncthreads();
    // column pass: one task = 4 adjacent columns x 4 adjacent output rows, byte maxima as packed 16-bit maxima
    for (int i = threadIdx.x; i < (DIL_TW / 4) * (DIL_TH / 4); i += DIL_NT) {
        const int xg = i % (DIL_TW / 4), q = i / (DIL_TW / 4);
        const int gx = X0 + 4 * xg;
        if (4 * q >= th || gx >= cols) continue;
        B4 o[4];
        const B4 zero = {0u, 0u};
        if (kh >= 4) {
            B4 mid = zero;
#pragma unroll 4
            for (int k = 3; k < kh; ++k) mid = b4_max(mid, b4_split(B[4 * q + k][xg]));
            const B4 a0 = b4_split(B[4 * q][xg]), a1 = b4_split(B[4 * q + 1][xg]), a2 = b4_split(B[4 * q + 2][xg]);
            const B4 b0 = b4_split(B[4 * q + kh][xg]), b1 = b4_split(B[4 * q + kh + 1][xg]), b2 = b4_split(B[4 * q + kh + 2][xg]);
            o[0] = b4_max(mid, b4_max(a0, b4_max(a1, a2)));      // rows past the staged ones only reach outputs past the image
            o[1] = b4_max(mid, b4_max(a1, b4_max(a2, b0)));
            o[2] = b4_max(mid, b4_max(a2, b4_max(b0, b1)));
            o[3] = b4_max(mid, b4_max(b0, b4_max(b1, b2)));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = zero;
                for (int k = 0; k < kh; ++k) o[j] = b4_max(o[j], b4_split(B[4 * q + j + k][xg]));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = Y0 + 4 * q + j;
            if (gy >= rows) continue;
            unsigned m = b4_join(o[j]);
            unsigned char* dp = dst + (size_t)gy * dstep + gx;
            const unsigned char* op = other ? other + (size_t)gy * ostep + gx : nullptr;
            if (gx + 3 < cols && ((uintptr_t)dp & 3) == 0 && ((uintptr_t)op & 3) == 0) {
                if (op) m &= *(const unsigned*)op;
                *(unsigned*)dp = m;
            } else {
                for (int k = 0; k < 4 && gx + k < cols; ++k) {
                    unsigned v = (m >> (8 * k)) & 255u;
                    if (op) v &= op[k];
                    dp[k] = (unsigned char)v;
                }
            }
        }
    }
}

// generic element sizes: separable running max through a temporary plane
__global__ __launch_bounds__(256) void k_dilate_rows(const unsigned char* src, size_t sstep, int rows, int cols, int kw, unsigned char* tmp) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const unsigned char* s = src + (size_t)y * sstep;
    const int lo = max(x - kw / 2, 0), hi = min(x - kw / 2 + kw, cols);
    int m = 0;
    for (int k = lo; k < hi; ++k) m = max(m, (int)s[k]);
    tmp[(size_t)y * cols + x] = (unsigned char)m;
}
__global__ __launch_bounds__(256) void k_dilate_cols_and(const unsigned char* tmp, int rows, int cols, int kh, const unsigned char* other, size_t ostep,
                                                         unsigned char* dst, size_t dstep) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const int lo = max(y - kh / 2, 0), hi = min(y - kh / 2 + kh, rows);
    int m = 0;
    for (int k = lo; k < hi; ++k) m = max(m, (int)tmp[(size_t)k * cols + x]);
    if (other) m &= other[(size_t)y * ostep + x];
    dst[(size_t)y * dstep + x] = (unsigned char)m;
}

// N3  GainCompensator::apply (W:241-244): multiply(image, gains_(index, 0), image) on a CV_8U image.  cv::multiply with
// a double scalar works in CV_64F (arithm_op: muldiv => depth2 = CV_64F, wtype = CV_64F): every byte becomes
// saturate_cast<uchar>(cvRound((double)byte * gain)), cvRound = cvtsd2si (round-half-even, NaN / overflow -> INT_MIN -> 0).
__device__ __forceinline__ unsigned gain_byte(unsigned v, double gain) {
    const double t = __builtin_rint((double)v * gain);
    const int iv = (t >= -2147483648.0 && t <= 2147483647.0) ? (int)t : INT_MIN;
    return (unsigned)iv <= 255u ? (unsigned)iv : (iv > 0 ? 255u : 0u);
}
template <bool VEC>
__global__ __launch_bounds__(256) void k_gain_apply(const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep, int rows, int row_bytes, double gain) {
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= rows) return;
    if constexpr (VEC) {
        const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
        if (x >= row_bytes) return;
        const unsigned v = *(const unsigned*)(src + (size_t)y * sstep + x);   // row_bytes is padded to the step: the last dword stays inside the row
        const unsigned o = gain_byte(v & 255u, gain) | (gain_byte((v >> 8) & 255u, gain) << 8) | (gain_byte((v >> 16) & 255u, gain) << 16) | (gain_byte(v >> 24, gain) << 24);
        if (x + 4 <= row_bytes) *(unsigned*)(dst + (size_t)y * dstep + x) = o;
        else for (int k = 0; x + k < row_bytes; ++k) dst[(size_t)y * dstep + x + k] = (unsigned char)((o >> (8 * k)) & 255u);
    } else {
        const int x = blockIdx.x * 64 + (threadIdx.x & 63);
        if (x >= row_bytes) return;
        dst[(size_t)y * dstep + x] = (unsigned char)gain_byte(src[(size_t)y * sstep + x], gain);
    }
}


// A14  the glue conversions of the reference's main(): images_warped[i].convertTo(images_warped_f[i], CV_32F) (W:261),
// images_warped_f[k].convertTo(images_warped_s[k], CV_16S) (W:294), result.convertTo(CV_8U) (imwrite's input, W:315) - Mat::convertTo with
// alpha = 1, beta = 0: widening is exact, narrowing is saturate_cast (float -> short / uchar: cvRound = round-half-even with cvtss2si's
// NaN / overflow -> INT_MIN, then the clamp).  One element per thread over the rows' channel values, dword stores where the rows allow.
template <class S, class D>
__device__ __forceinline__ D convert_one(S v) {
    if constexpr (sizeof(D) >= sizeof(S) && !(sizeof(S) == 4 && sizeof(D) == 4)) return (D)v;               // u8 -> s16 / f32, s16 -> f32: exact
    else if constexpr (sizeof(S) == 4 && sizeof(D) == 2) return (D)isxd::sat_s16(isxd::cvround_x86(v));       // f32 -> s16
    else if constexpr (sizeof(S) == 4 && sizeof(D) == 1) return (D)isxd::sat_u8(isxd::cvround_x86(v));        // f32 -> u8
    else return (D)isxd::sat_u8((int)v);                                                                        // s16 -> u8
}
// N values per thread as ONE vector load and ONE vector store, at whatever alignment the rows have (a dense cv::Mat row of 3425 CV_16SC3
// pixels starts on a 2-byte boundary; unaligned global access is legal on this part).  The row's last, partial group goes value by value.
template <class S, class D, int N>
__global__ __launch_bounds__(256) void k_convert(const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep, int rows, int n) {
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= rows) return;
    const S* s = (const S*)(src + (size_t)y * sstep);
    D* d = (D*)(dst + (size_t)y * dstep);
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * N;
    if (x >= n) return;
    if (x + N <= n) {
        typedef S svn __attribute__((ext_vector_type(N), aligned(1)));
        typedef D dvn __attribute__((ext_vector_type(N), aligned(1)));
        const svn v = *(const svn*)(s + x);
        dvn o;
#pragma unroll
        for (int k = 0; k < N; ++k) o[k] = convert_one<S, D>(v[k]);
        *(dvn*)(d + x) = o;
    } else for (int k = x; k < n; ++k) d[k] = convert_one<S, D>(s[k]);
}

}  // namespace

// dilate(mask, MORPH_RECT kw x kh) [& other] between DEVICE buffers with the one-kernel path (elements up to 33 a side): what
// isx_mask_dilate_and launches, callable from the blender (isx_blender_feed_dilated writes the result straight into the mask it keeps)
namespace isx {
int dilate_and_device(const unsigned char* mask, size_t mstep, const unsigned char* other, size_t ostep, int rows, int cols, int kw, int kh,
                      unsigned char* dst, size_t dstep, hipStream_t st) {
    ISX_CHECK_ARG(kw >= 1 && kh >= 1 && kw <= DIL_MAXK && kh <= DIL_MAXK, ISX_ERR_UNSUPPORTED, "dilate: a %d x %d element exceeds the fused kernel's %d a side", kw, kh, DIL_MAXK);
    ISX_LAUNCH("dilate_and", (double)rows * cols * (other ? 3.0 : 2.0), st, k_dilate_and, dim3(cdiv(cols, DIL_TW), cdiv(rows, DIL_TH)), dim3(DIL_NT), 0,
               mask, mstep, rows, cols, kw, kh, other, other ? ostep : (size_t)0, dst, dstep);
    return ISX_OK;
}
}  // namespace isx

extern "C" {

int isx_mask_dilate_and(const isx_mat* mask, const isx_mat* other, int kw, int kh, isx_mat* out, int device, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_TRY(check_mat(mask, "dilate: mask"));
    ISX_TRY(check_mat(out, "dilate: out"));
    ISX_CHECK_ARG(mask->type == ISX_8UC1 && out->type == ISX_8UC1, ISX_ERR_TYPE, "dilate: masks must be CV_8U");
    ISX_CHECK_ARG(out->rows == mask->rows && out->cols == mask->cols, ISX_ERR_SIZE, "dilate: out is %dx%d, mask is %dx%d", out->cols, out->rows, mask->cols, mask->rows);
    ISX_CHECK_ARG(kw >= 1 && kh >= 1 && kw <= 4096 && kh <= 4096, ISX_ERR_INVALID, "dilate: bad structuring element %dx%d", kw, kh);
    if (other) {
        ISX_TRY(check_mat(other, "dilate: other"));
        ISX_CHECK_ARG(other->type == ISX_8UC1 && other->rows == mask->rows && other->cols == mask->cols, ISX_ERR_SIZE, "dilate: the AND operand must be a CV_8U mask of the same size");
    }
    ISX_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    MatStage sm, so, sd;
    DevBuf tmp;
    ISX_TRY(sm.use_in(mask, st, "dilate: mask"));
    if (other) ISX_TRY(so.use_in(other, st, "dilate: other"));
    ISX_TRY(sd.use_out(out, st, "dilate: out"));
    const int rows = mask->rows, cols = mask->cols;
    if (kw <= DIL_MAXK && kh <= DIL_MAXK) {
        ISX_TRY(dilate_and_device((const unsigned char*)sm.d.data, sm.d.step, other ? (const unsigned char*)so.d.data : nullptr, other ? so.d.step : 0, rows, cols, kw, kh,
                                  (unsigned char*)sd.d.data, sd.d.step, st));
        ISX_TRY(sd.finish_out(st));
        if (mask->device < 0 || out->device < 0 || (other && other->device < 0)) ISX_HIP(hipStreamSynchronize(st));   // the staging buffers are freed on return
        return ISX_OK;
    }
    ISX_TRY(tmp.reserve((size_t)rows * cols));
    dim3 grid(cdiv(cols, 64), cdiv(rows, 4));
    ISX_LAUNCH("dilate_rows", (double)rows * cols * 2.0, st, k_dilate_rows, grid, dim3(256), 0, (const unsigned char*)sm.d.data, sm.d.step, rows, cols, kw, (unsigned char*)tmp.p);
    ISX_LAUNCH("dilate_cols_and", (double)rows * cols * 3.0, st, k_dilate_cols_and, grid, dim3(256), 0, (const unsigned char*)tmp.p, rows, cols, kh,
               other ? (const unsigned char*)so.d.data : nullptr, other ? so.d.step : 0, (unsigned char*)sd.d.data, sd.d.step);
    ISX_TRY(sd.finish_out(st));
    ISX_HIP(hipStreamSynchronize(st));   // tmp is freed on return
    return ISX_OK;
} ISX_EXIT("isx_mask_dilate_and")

int isx_gain_apply(isx_mat* image, double gain, int device, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_TRY(check_mat(image, "gain_apply: image"));
    ISX_CHECK_ARG(image->type == ISX_8UC3 || image->type == ISX_8UC1, ISX_ERR_TYPE, "gain_apply: image must be CV_8UC3 or CV_8UC1, got %s", type_name(image->type));
    ISX_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    MatStage si, so;
    ISX_TRY(si.use_in(image, st, "gain_apply: image"));
    ISX_TRY(so.use_out(image, st, "gain_apply: image"));
    const int rows = image->rows, row_bytes = image->cols * (image->type == ISX_8UC3 ? 3 : 1);
    // dword path: aligned rows whose last (possibly partial) dword still lies inside the row pitch
    const bool vec = ((uintptr_t)si.d.data % 4 == 0) && (si.d.step % 4 == 0) && ((uintptr_t)so.d.data % 4 == 0) && (so.d.step % 4 == 0) &&
                     (size_t)((row_bytes + 3) & ~3) <= si.d.step;
    const double bytes = 2.0 * rows * row_bytes;
    if (vec) ISX_LAUNCH("gain_apply", bytes, st, (k_gain_apply<true>), dim3(cdiv(cdiv(row_bytes, 4), 64), cdiv(rows, 4)), dim3(256), 0,
                        (const unsigned char*)si.d.data, si.d.step, (unsigned char*)so.d.data, so.d.step, rows, row_bytes, gain);
    else ISX_LAUNCH("gain_apply", bytes, st, (k_gain_apply<false>), dim3(cdiv(row_bytes, 64), cdiv(rows, 4)), dim3(256), 0,
                    (const unsigned char*)si.d.data, si.d.step, (unsigned char*)so.d.data, so.d.step, rows, row_bytes, gain);
    ISX_TRY(so.finish_out(st));
    if (image->device < 0) ISX_HIP(hipStreamSynchronize(st));   // the staging buffers are freed on return
    return ISX_OK;
} ISX_EXIT("isx_gain_apply")

int isx_convert_to(const isx_mat* src, isx_mat* dst, int device, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_TRY(check_mat(src, "convertTo: src"));
    ISX_TRY(check_mat(dst, "convertTo: dst"));
    ISX_CHECK_ARG(dst->rows == src->rows && dst->cols == src->cols && mat_cn(dst->type) == mat_cn(src->type), ISX_ERR_SIZE,
                  "convertTo: dst %dx%dx%d does not match src %dx%dx%d", dst->cols, dst->rows, mat_cn(dst->type), src->cols, src->rows, mat_cn(src->type));
    const int sd = mat_depth(src->type), dd = mat_depth(dst->type);      // CV_8U 0, CV_16S 3, CV_32F 5
    ISX_CHECK_ARG((sd == 0 || sd == 3 || sd == 5) && (dd == 0 || dd == 3 || dd == 5) && sd != dd, ISX_ERR_TYPE,
                  "convertTo: %s -> %s (CV_8U, CV_16S and CV_32F depths, different ones)", type_name(src->type), type_name(dst->type));
    ISX_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    MatStage si, so;
    ISX_TRY(si.use_in(src, st, "convertTo: src"));
    ISX_TRY(so.use_out(dst, st, "convertTo: dst"));
    const int rows = src->rows, n = src->cols * mat_cn(src->type);
    const size_t ss = sd == 0 ? 1 : (sd == 3 ? 2 : 4), ds = dd == 0 ? 1 : (dd == 3 ? 2 : 4);
    const double bytes = (double)rows * n * (double)(ss + ds);
    const unsigned char* sp = (const unsigned char*)si.d.data;
    unsigned char* dp = (unsigned char*)so.d.data;
    // values per thread: 16 bytes of the wider side (8 where the wider side is 2 bytes)
#define ISX_CVT(S, D, N) ISX_LAUNCH("convert_to", bytes, st, (k_convert<S, D, N>), dim3(cdiv(cdiv(n, N), 64), cdiv(rows, 4)), dim3(256), 0, sp, si.d.step, dp, so.d.step, rows, n)
    if (sd == 0 && dd == 3) ISX_CVT(unsigned char, short, 8);
    else if (sd == 0 && dd == 5) ISX_CVT(unsigned char, float, 4);
    else if (sd == 3 && dd == 5) ISX_CVT(short, float, 4);
    else if (sd == 5 && dd == 3) ISX_CVT(float, short, 4);
    else if (sd == 5 && dd == 0) ISX_CVT(float, unsigned char, 4);
    else ISX_CVT(short, unsigned char, 8);
#undef ISX_CVT
    ISX_TRY(so.finish_out(st));
    if (src->device < 0 || dst->device < 0) ISX_HIP(hipStreamSynchronize(st));   // the staging buffers are freed on return
    return ISX_OK;
} ISX_EXIT("isx_convert_to")

}  // extern "C"
