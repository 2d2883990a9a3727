// linear_blend.hip — the reference's in-tree single-band seam-ramp pair blend (B:141-717) on gfx950.
// Stages (all row-major f32, as in the reference):
//   k_lin_cost     costV SSD map                                  B:207-261
//   k_lin_seam_*   greedy seam walk as a composition of chunk maps B:268-307   (maps in parallel, 91 dependent steps, rows in parallel)
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

// The greedy seam walk (B:268-307) without its serial chain.  The walk is a composition of per-row maps  x -> x + dir[y][x]  (dir = -1 / 0 / +1
// from the three costs below, the reference's comparisons and tie order; columns clamped into the cost map), and a composition of maps can be
// bracketed any way one likes:
//   k_lin_seam_maps  every chunk of SEAM_ROWS rows, every start column, in parallel: where does a walk that ENTERS the chunk at column x leave
//                    it?  One thread per (chunk, column) walks its 24 rows out of a cost tile in LDS; the answer is a signed byte (|x' - x| <= 24).
//   k_lin_seam_entry one block walks the chunk maps instead of the rows: 91 dependent steps for the 2170 rows of a 4K pair, the maps of 46 chunks
//                    at a time staged in LDS on the +-1104 columns the seam can reach within them.
//   k_lin_seam_fill  every chunk in parallel again: from its entry column the 24 rows of the seam itself.
// Every step evaluates the same function of (row, column) the sequential walk evaluates: the same seam, point for point (tests/test_gpu_blend.py
// against the oracle's walk, incl. a 4K pair).  Round 4's form - one walker wave following step tables that four producer waves built two chunks
// ahead - paid an LDS latency per row: 196 us of the 0.33 ms a 4K pair took (first version: 483 us).
constexpr int SEAM_ROWS = 24;          // rows per chunk map (|exit - entry| <= 24 fits a signed byte with room)
constexpr int SEAM_TILE = 256;         // start columns per block of k_lin_seam_maps
constexpr int SEAM_HALO = 32;          // columns either side of them that a 24-row walk can reach (>= SEAM_ROWS + 1), a multiple of the loads' width
constexpr int SEAM_GROUP = 46;         // chunk maps per LDS stage of k_lin_seam_entry (46 x 2212 bytes = 99 KB of the CU's 160 KB; a 4K pair's 91 chunks: two stages)
constexpr int SEAM_REACH = SEAM_ROWS * SEAM_GROUP;      // columns the seam can move within one stage
static_assert(SEAM_HALO >= SEAM_ROWS + 1 && SEAM_ROWS < 127, "a chunk's walk stays inside its tile + halo and its displacement inside a signed byte");

__device__ __forceinline__ int seam_dir(float a, float b, float c) {     // B:283-300: left / stay / right from the three costs below
    // if (a == b && a == c) stay; else if (a <= b && a <= c) left; else if (b <= a && b <= c) stay; else if (c <= a && c <= b) right
    const bool all = (a == b) & (a == c), la = (a <= b) & (a <= c), lb = (b <= a) & (b <= c), lc = (c <= a) & (c <= b);
    int d = lc ? 1 : 0;
    d = lb ? 0 : d;
    d = la ? -1 : d;
    return all ? 0 : d;
}
// one row of the walk at column px (tile coordinates: LDS column = px - x_lo): xl = max(px - 1, 0), xr = min(px + 1, cw - 1), the oracle's clamps
__device__ __forceinline__ int seam_step(const float* row, int px, int x_lo, int cw) {
    const int xl = max(px - 1, 0), xr = min(px + 1, cw - 1);
    const int d = seam_dir(row[xl - x_lo], row[px - x_lo], row[xr - x_lo]);
    return d < 0 ? xl : (d > 0 ? xr : px);
}

// grid (ceil(cw / SEAM_TILE), chunks): maps[chunk][x] = (column a walk entering the chunk at x leaves it at) - x
__global__ __launch_bounds__(SEAM_TILE + 2 * SEAM_HALO) void k_lin_seam_maps(LinGeom g, const float* costV, signed char* maps) {
    __shared__ float tile[SEAM_ROWS][SEAM_TILE + 2 * SEAM_HALO];
    const int cw = g.iBr + 2, last = g.iHe - 1;
    const int j = blockIdx.y, py = j * SEAM_ROWS, nr = min(SEAM_ROWS, last - py);
    const int x_lo = (int)blockIdx.x * SEAM_TILE - SEAM_HALO, t = threadIdx.x, x = x_lo + t;
    const int xc = min(max(x, 0), cw - 1);
    for (int r = 0; r < nr; ++r) tile[r][t] = costV[(size_t)(py + 1 + r) * cw + xc];      // (a column outside the map is never stepped on: the clamps keep the walk inside)
    __syncthreads();
    if (t < SEAM_HALO || t >= SEAM_HALO + SEAM_TILE || x >= cw) return;
    int px = x;
    for (int r = 0; r < nr; ++r) px = seam_step(tile[r], px, x_lo, cw);
    maps[(size_t)j * cw + x] = (signed char)(px - x);
}

// one block: entry[j] = the seam's column on entering chunk j (entry[0] = iBr / 2, B:269).  A stage = the maps of SEAM_GROUP chunks on the columns
// the seam can reach within them, fetched as unaligned dwords, eight in flight per thread (fetched byte by byte, one after the other, the two
// stages of a 4K pair took 105 us), then SEAM_GROUP dependent LDS bytes.
typedef unsigned lin_u32_a1 __attribute__((aligned(1)));
constexpr int SEAM_WROW = (2 * SEAM_REACH + 1 + 3) / 4;      // dwords of one staged map row
__global__ __launch_bounds__(1024) void k_lin_seam_entry(LinGeom g, const signed char* maps, int nchunks, int* entry) {
    __shared__ unsigned win[SEAM_GROUP][SEAM_WROW];
    __shared__ int s_px;
    const int cw = g.iBr + 2;
    int px = g.iBr / 2;
    for (int j0 = 0; j0 < nchunks; j0 += SEAM_GROUP) {
        const int nj = min(SEAM_GROUP, nchunks - j0), w0 = max(px - SEAM_REACH, 0), w1 = min(px + SEAM_REACH, cw - 1);
        const int nd = (w1 - w0 + 1 + 3) / 4, total = nj * nd;      // (a row's last dword may read up to 3 bytes of the next row: inside the buffer's padding)
        for (int base = (int)threadIdx.x; base < total; base += 8 * 1024) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 1024;
                v[u] = 0u;
                if (idx < total) { const int i = idx / nd, k = idx - i * nd; v[u] = *(const lin_u32_a1*)(maps + (size_t)(j0 + i) * cw + w0 + 4 * k); }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 1024;
                if (idx < total) { const int i = idx / nd, k = idx - i * nd; win[i][k] = v[u]; }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 0; i < nj; ++i) { entry[j0 + i] = px; px += (int)((const signed char*)win[i])[px - w0]; }
            s_px = px;
        }
        __syncthreads();
        px = s_px;
        __syncthreads();
    }
}

// grid (chunks): seam[py + 1 + r], r < rows of the chunk, from the chunk's entry column
__global__ __launch_bounds__(64) void k_lin_seam_fill(LinGeom g, const float* costV, const int* entry, int* seam) {
    __shared__ float tile[SEAM_ROWS][64];
    const int cw = g.iBr + 2, last = g.iHe - 1;
    const int j = blockIdx.x, py = j * SEAM_ROWS, nr = min(SEAM_ROWS, last - py), lane = threadIdx.x;
    const int e = entry[j], x_lo = e - 32;
    const int xc = min(max(x_lo + lane, 0), cw - 1);
    for (int r = 0; r < nr; ++r) tile[r][lane] = costV[(size_t)(py + 1 + r) * cw + xc];
    __syncthreads();
    if (lane != 0) return;
    if (j == 0) seam[0] = e;
    int px = e;
    for (int r = 0; r < nr; ++r) { px = seam_step(tile[r], px, x_lo, cw); seam[py + 1 + r] = px; }
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
    const int nchunks = std::max(cdiv(g.iHe - 1, SEAM_ROWS), 1);
    const int half_ibr = g.iBr / 2;
    size_t cost_b = ((size_t)g.iHe * cw * 4 + 255) & ~(size_t)255, seam_b = ((size_t)g.iHe * 4 + 255) & ~(size_t)255,
           m_b = ((size_t)g.height * mw * 4 + 255) & ~(size_t)255, maps_b = (((size_t)nchunks * cw + 255) & ~(size_t)255) + 256, entry_b = ((size_t)nchunks * 4 + 255) & ~(size_t)255;
    ISX_TRY(scratch.reserve(cost_b + seam_b + 2 * m_b + maps_b + entry_b));
    float* costV = (float*)scratch.p;
    int* seam = (int*)((char*)scratch.p + cost_b);
    float* m1 = (float*)((char*)scratch.p + cost_b + seam_b);
    float* m2 = (float*)((char*)m1 + m_b);
    signed char* maps = (signed char*)((char*)m2 + m_b);
    int* entry = (int*)((char*)maps + maps_b);
    const unsigned char* i1 = (const unsigned char*)s1.d.data;
    const unsigned char* i2 = (const unsigned char*)s2.d.data;
    // algorithmic bytes of the launches (bench.py --a13): every input read once, every output written once
    const double cells = (double)g.iHe * cw, ocells = (double)g.height * mw, ppx = (double)g.panoHe * g.panoBr;
    const double left_px = (double)g.rows1 * g.dx2, right_px = (double)g.rows2 * std::max(g.panoBr - g.cols1, 0), ov_px = (double)g.height * g.width;
    ISX_LAUNCH("lin_cost", cells * 28.0, st, k_lin_cost, dim3(cdiv(cw, 256), g.iHe), dim3(256), 0, g, i1, i2, costV);
    if (g.iHe < 2) ISX_HIP(hipMemcpyAsync(seam, &half_ibr, sizeof(int), hipMemcpyHostToDevice, st));      // a one-row overlap: seam[0] alone (B:269)
    else {
        ISX_LAUNCH("lin_seam_maps", cells * 5.0, st, k_lin_seam_maps, dim3(cdiv(cw, SEAM_TILE), nchunks), dim3(SEAM_TILE + 2 * SEAM_HALO), 0, g, (const float*)costV, maps);
        ISX_LAUNCH("lin_seam_entry", 0.0, st, k_lin_seam_entry, dim3(1), dim3(1024), 0, g, (const signed char*)maps, nchunks, entry);
        ISX_LAUNCH("lin_seam_fill", 0.0, st, k_lin_seam_fill, dim3(nchunks), dim3(64), 0, g, (const float*)costV, (const int*)entry, seam);
    }
    ISX_LAUNCH("lin_classify", ocells * 32.0, st, k_lin_classify, dim3(cdiv(mw, 256), g.height), dim3(256), 0, g, i1, i2, m1, m2);
    ISX_LAUNCH("lin_rows", ocells * 16.0, st, k_lin_rows, dim3(g.height), dim3(64), 0, g, seam, m1, m2);
    ISX_LAUNCH("lin_compose", ppx * 12.0 + (left_px + right_px) * 12.0 + ov_px * 32.0, st, k_lin_compose, dim3(cdiv(g.panoBr, 256), g.panoHe), dim3(256), 0, g, i1, i2, m1, m2,
               (unsigned char*)sp.d.data);
    if (seam_x) ISX_HIP(hipMemcpyAsync(seam_x, seam, (size_t)g.iHe * 4, hipMemcpyDeviceToHost, st));
    ISX_TRY(sp.finish_out(st));
    ISX_HIP(hipStreamSynchronize(st));   // the seam and a host pano are complete when the call returns (cv::Mat semantics)
    return ISX_OK;
} ISX_EXIT("isx_blend_pair_linear")

}  // extern "C"
