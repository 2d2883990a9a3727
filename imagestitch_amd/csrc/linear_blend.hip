// linear_blend.hip — the reference's in-tree single-band seam-ramp pair blend (B:141-717) on gfx950.
// Stages (all row-major f32, as in the reference):
//   k_lin_cost     costV SSD map                                  B:207-261
//   k_lin_seam     greedy seam walk (sequential by nature)        B:268-307
//   k_lin_classify gray (cvtColor RGB2GRAY) + overlap classes     B:313-470
//   k_lin_rows     per-row left/right scan, ramp weights, cleanup B:483-572   (one wave per row)
//   k_lin_compose  left-only / right-only / weighted overlap      B:579-711
// Where the reference indexes out of bounds (cost rows past images1.rows for tiles of different
// height, a seam walking off the cost map) rows are skipped / columns clamped, exactly as the
// oracle documents; all in-bounds arithmetic is literal (the ramp expressions are double).
#include "isx_device.hpp"
#include "isx_internal.hpp"

#include <algorithm>

using namespace isx;
using namespace isxd;

namespace {

struct LinGeom {
    int rows1, cols1, rows2, cols2;
    int dx2, dy, dy1, dy2;
    int panoBr, panoHe, width, height, iBr, iHe;
    size_t step1, step2, pstep;  // bytes
};

__device__ __forceinline__ const float* rowp(const unsigned char* base, size_t step, int y) { return (const float*)(base + (size_t)y * step); }
__device__ __forceinline__ float sqrf(float v) { return v * v; }

__global__ __launch_bounds__(256) void k_lin_cost(LinGeom g, const unsigned char* img1, const unsigned char* img2, float* costV) {
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    int cw = g.iBr + 2;
    if (x >= cw) return;
    float c = 0.f;
    int y0, y1, off;
    if (g.dy > 0) { y0 = g.dy2; y1 = g.iHe - g.dy2; off = g.dy2; }
    else if (g.dy < 0) { y0 = g.dy1; y1 = g.iHe - g.dy1; off = g.dy1; }
    else { y0 = 0; y1 = min(g.rows1, g.rows2); off = 0; }
    if (y >= y0 && y < y1 && y < g.rows1 && y - off >= 0 && y - off < g.rows2 && x >= 1 && x < g.iBr - 1) {
        const float* p1 = rowp(img1, g.step1, y);
        const float* p2 = rowp(img2, g.step2, y - off);
        int a = (x + g.dx2) * 3, b = x * 3;
        c = ((sqrf(p1[a] - p2[b]) + sqrf(p1[a + 1] - p2[b + 1]) + sqrf(p1[a + 2] - p2[b + 2])) +
             (sqrf(p1[a + 3] - p2[b - 3]) + sqrf(p1[a + 4] - p2[b - 2]) + sqrf(p1[a + 5] - p2[b - 1]))) / 2;
    }
    costV[(size_t)y * cw + x] = c;
}

// Five waves.  The seam moves at most one column per row (B:268-307), and which way it moves from column x of row y depends on
// costV[y + 1][x - 1 .. x + 1] alone.  So waves 1-4 (the producers) turn a window of SEAM_ROWS x 256 costs into a table of steps
// (-1 / 0 / +1 per cell, every lane four columns of its rows, the reference's comparisons and tie order), and wave 0 (the walker)
// only follows the table: one dependent LDS byte per row.  Chunks are double-buffered in LDS; the producer fetches a chunk's costs
// two chunks before it is walked, centred where the seam stands then - by the time it is walked the seam has moved at most
// 3 * SEAM_ROWS = 72 columns, well inside the 128 the window reaches to either side - so neither the memory latency nor the
// comparisons are on the walker's path.  (First version: one wave, a 64-column window per 24 rows loaded and then walked with three
// LDS reads and the comparisons per row: 483 us for the 2170 rows of a 4K pair.)
constexpr int SEAM_ROWS = 24;
constexpr int SEAM_W = 256;
static_assert(3 * SEAM_ROWS + 2 < SEAM_W / 2, "a window fetched three chunks ahead still holds the seam and its neighbours");

__device__ __forceinline__ int seam_dir(float a, float b, float c) {     // B:283-300: left / stay / right from the three costs below
    // if (a == b && a == c) stay; else if (a <= b && a <= c) left; else if (b <= a && b <= c) stay; else if (c <= a && c <= b) right
    const bool all = (a == b) & (a == c), la = (a <= b) & (a <= c), lb = (b <= a) & (b <= c), lc = (c <= a) & (c <= b);
    int d = lc ? 1 : 0;
    d = lb ? 0 : d;
    d = la ? -1 : d;
    return all ? 0 : d;
}

constexpr int SEAM_PRODUCERS = 4;                        // producer waves: each fetches and tabulates every 4th row of a chunk
constexpr int SEAM_PR = SEAM_ROWS / SEAM_PRODUCERS;      // rows of a chunk per producer wave
static_assert(SEAM_ROWS % SEAM_PRODUCERS == 0, "rows of a chunk are dealt evenly to the producer waves");

__global__ __launch_bounds__(64 * (1 + SEAM_PRODUCERS)) void k_lin_seam(LinGeom g, const float* costV, int* seam) {
    __shared__ signed char dir[2][SEAM_ROWS][SEAM_W];
    // s_px[j & 1]: the column the walker reached at the end of chunk j.  Two slots: the walker goes straight on into chunk j + 1 and writes
    // the other slot while slower waves may still be reading this one (a single slot was a formal race: ADVICE r2)
    __shared__ int s_x0[2], s_px[2];
    const int cw = g.iBr + 2, lane = threadIdx.x & 63, last = g.iHe - 1;     // rows 1 .. last are chosen by the walk
    const bool producer = threadIdx.x >= 64;
    const int pw = (int)(threadIdx.x >> 6) - 1;          // producer wave index (rows pw, pw + 4, ...)
    int px = g.iBr / 2;
    if (threadIdx.x == 0) seam[0] = px;
    if (last < 1) return;
    const int nchunks = (last + SEAM_ROWS - 1) / SEAM_ROWS;
    float4 S0[SEAM_PR], S1[SEAM_PR];         // producer: its rows of two chunks in flight (even chunks in S0, odd ones in S1; registers)
    int x0_0 = 0, x0_1 = 0;
    auto fetch = [&](float4 (&S)[SEAM_PR], int& x0, int j, int centre) {   // chunk j = rows j * SEAM_ROWS + 1 ..., columns centre - 128 .. centre + 127
        x0 = centre - SEAM_W / 2;
        const int py = j * SEAM_ROWS;
        const bool inside = x0 >= 0 && x0 + SEAM_W <= cw;                  // wave-uniform: no clamping needed
#pragma unroll
        for (int i = 0; i < SEAM_PR; ++i) {
            const int r = pw + SEAM_PRODUCERS * i;
            const float* q = costV + (size_t)min(py + 1 + r, last) * cw;
            const int c = x0 + 4 * lane;
            float4 v;
            if (inside) { v.x = q[c]; v.y = q[c + 1]; v.z = q[c + 2]; v.w = q[c + 3]; }
            else { v.x = q[min(max(c, 0), cw - 1)]; v.y = q[min(max(c + 1, 0), cw - 1)]; v.z = q[min(max(c + 2, 0), cw - 1)]; v.w = q[min(max(c + 3, 0), cw - 1)]; }
            S[i] = v;
        }
    };
    auto publish = [&](const float4 (&S)[SEAM_PR], int x0, int b) {       // the step table of a chunk into LDS buffer b
#pragma unroll
        for (int i = 0; i < SEAM_PR; ++i) {
            const int r = pw + SEAM_PRODUCERS * i;
            const float4 v = S[i];
            const float left = __shfl_up(v.w, 1), right = __shfl_down(v.x, 1);     // the window's outermost columns are never reached
            const int d0 = seam_dir(left, v.x, v.y), d1 = seam_dir(v.x, v.y, v.z), d2 = seam_dir(v.y, v.z, v.w), d3 = seam_dir(v.z, v.w, right);
            *(unsigned*)&dir[b][r][4 * lane] = (unsigned)(d0 & 255) | ((unsigned)(d1 & 255) << 8) | ((unsigned)(d2 & 255) << 16) | ((unsigned)(d3 & 255) << 24);
        }
        if (lane == 0 && pw == 0) s_x0[b] = x0;
    };
    auto walk = [&](int j) {                    // wave 0, lane 0: follow the table of chunk j
        const int b = j & 1, x0 = s_x0[b], py = j * SEAM_ROWS, nr = min(SEAM_ROWS, last - py);
        for (int r = 0; r < nr; ++r) {
            px = min(max(px + (int)dir[b][r][px - x0], 0), cw - 1);          // xl = max(px - 1, 0), xr = min(px + 1, cw - 1)
            seam[py + 1 + r] = px;
        }
        s_px[b] = px;
    };
    if (producer) {
        fetch(S0, x0_0, 0, px);
        if (nchunks > 1) fetch(S1, x0_1, 1, px);
        publish(S0, x0_0, 0);
        if (nchunks > 2) fetch(S0, x0_0, 2, px);          // S0 is free again
    }
    __syncthreads();
    for (int j = 0; j < nchunks; j += 2) {
        // even chunk j (LDS buffer 0) is walked while the table of chunk j + 1 (S1 -> buffer 1) is written
        if (!producer) { if (lane == 0) walk(j); }
        else if (j + 1 < nchunks) publish(S1, x0_1, 1);
        __syncthreads();
        px = s_px[0];
        if (producer && j + 3 < nchunks) fetch(S1, x0_1, j + 3, px);
        if (j + 1 >= nchunks) break;
        // odd chunk j + 1 (buffer 1) is walked while the table of chunk j + 2 (S0 -> buffer 0) is written
        if (!producer) { if (lane == 0) walk(j + 1); }
        else if (j + 2 < nchunks) publish(S0, x0_0, 0);
        __syncthreads();
        px = s_px[1];
        if (producer && j + 4 < nchunks) fetch(S0, x0_0, j + 4, px);
    }
}

__device__ __forceinline__ float gray_at(const float* p, int x) {
    // cvtColor(CV_RGB2GRAY) on CV_32FC3: c0*0.299f + c1*0.587f + c2*0.114f
    return p[3 * x] * 0.299f + p[3 * x + 1] * 0.587f + p[3 * x + 2] * 0.114f;
}

__global__ __launch_bounds__(256) void k_lin_classify(LinGeom g, const unsigned char* img1, const unsigned char* img2, float* m1, float* m2) {
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    int mw = g.width + 2;
    if (x >= mw) return;
    float a1, a2;
    if (x == 0 || x == g.width + 1) { a1 = 128.f; a2 = 128.f; }
    else {
        const float* p1 = rowp(img1, g.step1, g.dy > 0 ? y + g.dy2 : y);
        const float* p2 = rowp(img2, g.step2, g.dy < 0 ? y + g.dy1 : y);
        float ga = gray_at(p1, x + g.dx2 - 1), gb = gray_at(p2, x - 1);
        float thr = g.dy == 0 ? 10.f : 20.f;
        a1 = 0.f; a2 = 0.f;
        if (ga >= thr && gb >= thr) { a1 = 255.f; a2 = 255.f; }
        if (ga >= thr && gb < thr) { a1 = 1.f; a2 = 0.f; }
        if (ga < thr && gb >= thr) { a1 = 0.f; a2 = 1.f; }
        if (ga < thr && gb < thr) { a1 = 1.f; a2 = 1.f; }
    }
    m1[(size_t)y * mw + x] = a1;
    m2[(size_t)y * mw + x] = a2;
}

// one wave per overlap row
__global__ __launch_bounds__(64) void k_lin_rows(LinGeom g, const int* seam, float* m1, float* m2) {
    const int y = blockIdx.x, lane = threadIdx.x, mw = g.width + 2;
    float* p1 = m1 + (size_t)y * mw;
    float* p2 = m2 + (size_t)y * mw;
    int left = 0, right = 0;
    for (int x = 1 + lane; x < g.width + 1; x += 64) {
        float c = p2[x], l = p2[x - 1], r = p2[x + 1];
        if (c == 255.f && l == 0.f && r == 1.f) left = x;
        if ((c == 255.f && l == 0.f && r == 255.f) ||
            (l == 128.f && c == 255.f && r == 255.f && (x + 2 < mw ? p2[x + 2] : 0.f) == 255.f && (x + 3 < mw ? p2[x + 3] : 0.f) == 255.f)) left = x;
        if (l == 0.f && c == 255.f && r == 1.f) right = x;
        if (l == 255.f && c == 255.f && (r == 1.f || r == 128.f)) right = x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { left = max(left, __shfl_xor(left, o)); right = max(right, __shfl_xor(right, o)); }
    const int sx = seam[y + g.dy2 + g.dy1];
    for (int x = lane; x < g.width + 1; x += 64) {
        float a = p1[x], b = p2[x];
        if (x >= 1 && b == 255.f) {
            if (left && left == right) { a = 1.f; b = 0.f; }
            else if (x <= sx + 1) { a = (float)(1 - 0.5 * (x - left) / (sx + 1 - left)); b = 1 - a; }        // B:542
            else if (x > sx + 1 && x <= right) { a = (float)(0.5 * (right - x) / (right - sx - 1)); b = 1 - a; }   // B:548
        }
        if (a == 255.f) { a = 1.f; b = 0.f; }   // B:560-572
        p1[x] = a; p2[x] = b;
    }
}

__global__ __launch_bounds__(256) void k_lin_compose(LinGeom g, const unsigned char* img1, const unsigned char* img2,
                                                     const float* m1, const float* m2, unsigned char* pano) {
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.panoBr) return;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (x < g.dx2) {                         // image 1 only: pano row y + dy1 <- images1 row y
        int sy = y - g.dy1;
        if (sy >= 0 && sy < g.rows1) { const float* p = rowp(img1, g.step1, sy); o0 = p[3 * x]; o1 = p[3 * x + 1]; o2 = p[3 * x + 2]; }
    } else if (x >= g.cols1) {               // image 2 only
        int sy = g.dy > 0 ? y - g.dy2 : y;
        bool ok = g.dy > 0 ? (y >= g.dy2 && y < g.rows2) : (y < g.rows2);
        if (ok) { const float* p = rowp(img2, g.step2, sy); int xx = 3 * (x - g.dx2); o0 = p[xx]; o1 = p[xx + 1]; o2 = p[xx + 2]; }
    } else if (x < g.dx2 + g.width) {        // overlap: pano row y <- overlap row y - dy2 - dy1
        int oy = y - g.dy2 - g.dy1;
        if (oy >= 0 && oy < g.height) {
            const float* p1 = rowp(img1, g.step1, g.dy > 0 ? oy + g.dy2 : oy);
            const float* p2 = rowp(img2, g.step2, g.dy < 0 ? oy + g.dy1 : oy);
            int mw = g.width + 2;
            float w1 = m1[(size_t)oy * mw + (x - g.dx2 + 1)], w2 = m2[(size_t)oy * mw + (x - g.dx2 + 1)];
            int xx = 3 * (x - g.dx2);
            o0 = p1[3 * x] * w1 + p2[xx] * w2; o1 = p1[3 * x + 1] * w1 + p2[xx + 1] * w2; o2 = p1[3 * x + 2] * w1 + p2[xx + 2] * w2;
        }
    }
    float* q = (float*)(pano + (size_t)y * g.pstep) + 3 * x;
    q[0] = o0; q[1] = o1; q[2] = o2;
}

void geom_sizes(int rows1, int cols1, int rows2, int cols2, int tl1x, int tl1y, int tl2x, int tl2y, int* pr, int* pc) {
    (void)cols1;
    *pc = tl2x - tl1x + cols2;                                                           // B:152
    *pr = std::max(tl1y + rows1, tl2y + rows2) - std::min(tl1y, tl2y);                   // B:153
}

}  // namespace

namespace {
struct LinScratch { DevBuf buf; int device = -1; };
LinScratch& lin_scratch() {
    static thread_local LinScratch* s = new LinScratch();   // never destroyed at thread exit (the HIP runtime may be gone by then)
    return *s;
}
}  // namespace

extern "C" {

int isx_blend_pair_linear_release(void) ISX_ENTRY {
    clear_error();
    LinScratch& ls = lin_scratch();
    ls.buf.release();      // (hipFree needs no current device: the caller's is left as it is)
    ls.device = -1;
    return ISX_OK;
} ISX_EXIT("isx_blend_pair_linear_release")

int isx_blend_pair_linear_size(int rows1, int cols1, int rows2, int cols2, int tl1_x, int tl1_y, int tl2_x, int tl2_y,
                               int* pano_rows, int* pano_cols) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(pano_rows != nullptr && pano_cols != nullptr, ISX_ERR_INVALID, "blend_pair_linear_size: null argument");
    ISX_CHECK_ARG(rows1 > 0 && cols1 > 0 && rows2 > 0 && cols2 > 0, ISX_ERR_INVALID, "blend_pair_linear_size: empty tile");
    geom_sizes(rows1, cols1, rows2, cols2, tl1_x, tl1_y, tl2_x, tl2_y, pano_rows, pano_cols);
    return ISX_OK;
} ISX_EXIT("isx_blend_pair_linear_size")

int isx_blend_pair_linear(const isx_mat* images1, const isx_mat* images2, int tl1_x, int tl1_y, int tl2_x, int tl2_y,
                          isx_mat* pano, int* seam_x, int device, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_TRY(check_mat(images1, "blend_pair_linear: images1"));
    ISX_TRY(check_mat(images2, "blend_pair_linear: images2"));
    ISX_TRY(check_mat(pano, "blend_pair_linear: pano"));
    ISX_CHECK_ARG(images1->type == ISX_32FC3 && images2->type == ISX_32FC3 && pano->type == ISX_32FC3, ISX_ERR_TYPE,
                  "blend_pair_linear: images and pano must be CV_32FC3 (B:143-145)");
    LinGeom g;
    g.rows1 = images1->rows; g.cols1 = images1->cols; g.rows2 = images2->rows; g.cols2 = images2->cols;
    geom_sizes(g.rows1, g.cols1, g.rows2, g.cols2, tl1_x, tl1_y, tl2_x, tl2_y, &g.panoHe, &g.panoBr);
    ISX_CHECK_ARG(pano->rows == g.panoHe && pano->cols == g.panoBr, ISX_ERR_SIZE, "blend_pair_linear: pano is %dx%d, expected %dx%d",
                  pano->cols, pano->rows, g.panoBr, g.panoHe);
    g.dx2 = tl2_x - tl1_x;                                                               // B:158
    g.dy = tl2_y - tl1_y; g.dy1 = g.dy < 0 ? -g.dy : 0; g.dy2 = g.dy > 0 ? g.dy : 0;     // B:159-172
    int itlx = std::max(tl1_x, tl2_x), itly = std::max(tl1_y, tl2_y);
    int ibrx = std::min(tl1_x + g.cols1, tl2_x + g.cols2), ibry = std::min(tl1_y + g.rows1, tl2_y + g.rows2);
    ISX_CHECK_ARG(itlx < ibrx && itly < ibry, ISX_ERR_INVALID, "blend_pair_linear: the tiles do not overlap (B:182-183 returns without a result)");
    g.height = ibry - itly; g.width = ibrx - itlx;                                       // B:185-186
    g.iBr = g.cols1 - g.dx2; g.iHe = g.panoHe;                                           // B:191-192
    ISX_CHECK_ARG(g.dx2 >= 0 && g.iBr == g.width && g.iBr >= 3 && g.iBr <= g.cols2, ISX_ERR_UNSUPPORTED,
                  "blend_pair_linear: the demo assumes tile 2 lies to the right of tile 1 and extends past it");
    ISX_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    MatStage s1, s2, sp;
    ISX_TRY(s1.use_in(images1, st, "blend_pair_linear: images1"));
    ISX_TRY(s2.use_in(images2, st, "blend_pair_linear: images2"));
    ISX_TRY(sp.use_out(pano, st, "blend_pair_linear: pano"));
    g.step1 = s1.d.step; g.step2 = s2.d.step; g.pstep = sp.d.step;
    const int cw = g.iBr + 2, mw = g.width + 2;
    // work buffers (cost map, seam, the two weight maps: 38 MB for a 4K pair) persist per host thread and device, grow-only: a
    // hipMalloc / hipFree pair per call cost 0.24 ms of a 0.55 ms call (isx_blend_pair_linear_release returns them)
    LinScratch& ls = lin_scratch();
    if (ls.device != device) { ls.buf.release(); ls.device = device; }
    DevBuf& scratch = ls.buf;
    size_t cost_b = ((size_t)g.iHe * cw * 4 + 255) & ~(size_t)255, seam_b = ((size_t)g.iHe * 4 + 255) & ~(size_t)255,
           m_b = ((size_t)g.height * mw * 4 + 255) & ~(size_t)255;
    ISX_TRY(scratch.reserve(cost_b + seam_b + 2 * m_b));
    float* costV = (float*)scratch.p;
    int* seam = (int*)((char*)scratch.p + cost_b);
    float* m1 = (float*)((char*)scratch.p + cost_b + seam_b);
    float* m2 = (float*)((char*)m1 + m_b);
    const unsigned char* i1 = (const unsigned char*)s1.d.data;
    const unsigned char* i2 = (const unsigned char*)s2.d.data;
    ISX_LAUNCH("lin_cost", 0.0, st, k_lin_cost, dim3(cdiv(cw, 256), g.iHe), dim3(256), 0, g, i1, i2, costV);
    ISX_LAUNCH("lin_seam", 0.0, st, k_lin_seam, dim3(1), dim3(64 * (1 + SEAM_PRODUCERS)), 0, g, costV, seam);
    ISX_LAUNCH("lin_classify", 0.0, st, k_lin_classify, dim3(cdiv(mw, 256), g.height), dim3(256), 0, g, i1, i2, m1, m2);
    ISX_LAUNCH("lin_rows", 0.0, st, k_lin_rows, dim3(g.height), dim3(64), 0, g, seam, m1, m2);
    ISX_LAUNCH("lin_compose", 0.0, st, k_lin_compose, dim3(cdiv(g.panoBr, 256), g.panoHe), dim3(256), 0, g, i1, i2, m1, m2, (unsigned char*)sp.d.data);
    if (seam_x) ISX_HIP(hipMemcpyAsync(seam_x, seam, (size_t)g.iHe * 4, hipMemcpyDeviceToHost, st));
    ISX_TRY(sp.finish_out(st));
    ISX_HIP(hipStreamSynchronize(st));   // the seam and a host pano are complete when the call returns (cv::Mat semantics)
    return ISX_OK;
} ISX_EXIT("isx_blend_pair_linear")

}  // extern "C"
