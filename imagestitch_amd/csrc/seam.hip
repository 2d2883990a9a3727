// seam.hip — SURVEY §8(f) N1, the data-parallel part of the in-tree DP seam finder
// (S = 动态规划法寻找最佳缝合线/.../动态规划法寻找最佳缝合线.cpp) on gfx950:
//   k_seam_costs  computeCosts S:733-803 (costFunc_ COLOR): costV / costH of one intersection component
//   k_seam_dp     the dynamic programme of estimateSeam S:858-916: one wavefront step per row (column) of the
//                 component's ROI, the cells of a step in parallel, previous step's cost / reachability in LDS
//   host          seam direction + swap S:825-842, backtracking through the control map S:918-953
// Component analysis (findComponents / findEdges / resolveConflicts / getSeamTips / updateLabelsUsingSeam) stays with
// the caller: it produces labels_, the component's bounding rectangle and the seam tips p1, p2 consumed here.
// A labels_ read outside the union counts as "not this component" (the reference reads past the row, S:763).
#include "isx_device.hpp"
#include "isx_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace isx;
using namespace isxd;

namespace {

struct SeamGeom {
    const unsigned char* img1; size_t step1;
    const unsigned char* img2; size_t step2;
    const unsigned char* labels; size_t lstep;
    int uh, uw, label;
    int rx, ry, rw, rh;
    int dx1, dy1, dx2, dy2;
};

__device__ __forceinline__ int seam_label(const SeamGeom& g, int y, int x) {
    return ((unsigned)y < (unsigned)g.uh && (unsigned)x < (unsigned)g.uw) ? ((const int*)(g.labels + (size_t)y * g.lstep))[x] : 0;
}

// diffL2Square3<T> S:712-718
template <bool U8>
__device__ __forceinline__ float seam_diff(const SeamGeom& g, int y1, int x1, int y2, int x2) {
    if constexpr (U8) {
        const unsigned char* a = g.img1 + (size_t)y1 * g.step1 + (size_t)x1 * 3;
        const unsigned char* b = g.img2 + (size_t)y2 * g.step2 + (size_t)x2 * 3;
        const int d0 = (int)a[0] - (int)b[0], d1 = (int)a[1] - (int)b[1], d2 = (int)a[2] - (int)b[2];
        return (float)(d0 * d0 + d1 * d1 + d2 * d2);
    } else {
        const float* a = (const float*)(g.img1 + (size_t)y1 * g.step1) + (size_t)x1 * 3;
        const float* b = (const float*)(g.img2 + (size_t)y2 * g.step2) + (size_t)x2 * 3;
        const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
        float s = d0 * d0;
        s = s + d1 * d1;
        s = s + d2 * d2;
        return s;
    }
}

template <bool U8>
__global__ __launch_bounds__(256) void k_seam_costs(SeamGeom g, float* costV, float* costH) {
    const int cx = blockIdx.x * 64 + (threadIdx.x & 63), cy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cx > g.rw || cy > g.rh) return;
    const int x = g.rx + cx, y = g.ry + cy;
    const float bad = 3.f * 255.f * 255.f;   // normL2(Point3f(255, 255, 255), Point3f(0, 0, 0)), S:754
    const bool here = seam_label(g, y, x) == g.label;
    if (cy < g.rh) {   // S:757-779
        float c = bad;
        if (here && x > 0 && seam_label(g, y, x - 1) == g.label)
            c = (seam_diff<U8>(g, y + g.dy1, x + g.dx1 - 1, y + g.dy2, x + g.dx2) + seam_diff<U8>(g, y + g.dy1, x + g.dx1, y + g.dy2, x + g.dx2 - 1)) / 2;
        costV[(size_t)cy * (g.rw + 1) + cx] = c;
    }
    if (cx < g.rw) {   // S:784-802
        float c = bad;
        if (here && y > 0 && seam_label(g, y - 1, x) == g.label)
            c = (seam_diff<U8>(g, y + g.dy1 - 1, x + g.dx1, y + g.dy2, x + g.dx2) + seam_diff<U8>(g, y + g.dy1, x + g.dx1, y + g.dy2 - 1, x + g.dx2)) / 2;
        costH[(size_t)cy * g.rw + cx] = c;
    }
}

// One workgroup walks the ROI from the source row (column) to the destination row (column).  n = cells per step,
// LDS: cost[2][n] floats + reach[2][n] bytes (ping-pong).  control: rh x rw bytes, only reached cells are written
// (backtracking only follows written cells).  A cell has three candidate predecessors i, i - 1, i + 1 in the previous
// step: cost[i] + c1, cost[i - 1] + c2a + c2b, cost[i + 1] + c3a + c3b (S:866-875 / S:896-905, additions left to right).
constexpr int SEAM_NT = 1024;
struct SeamStep { int lab; float c1, c2a, c2b, c3a, c3b; };

__device__ __forceinline__ SeamStep seam_load_step(const SeamGeom& g, const float* costV, const float* costH, int horiz, int s, int i) {
    SeamStep t;
    const int rw = g.rw, rh = g.rh;
    if (horiz) {   // x = s, y = i: the seam follows along the upper side of pixels
        const int x = s, y = i;
        t.lab = seam_label(g, y + g.ry, x + g.rx) == g.label;
        t.c1 = costH[(size_t)y * rw + (x - 1)];
        t.c2a = y > 0 ? costH[(size_t)(y - 1) * rw + (x - 1)] : 0.f;
        t.c2b = y > 0 ? costV[(size_t)(y - 1) * (rw + 1) + x] : 0.f;
        t.c3a = y < rh - 1 ? costH[(size_t)(y + 1) * rw + (x - 1)] : 0.f;
        t.c3b = costV[(size_t)y * (rw + 1) + x];
    } else {       // y = s, x = i: the seam follows along the left side of pixels
        const int x = i, y = s;
        t.lab = seam_label(g, y + g.ry, x + g.rx) == g.label;
        t.c1 = costV[(size_t)(y - 1) * (rw + 1) + x];
        t.c2a = x > 0 ? costV[(size_t)(y - 1) * (rw + 1) + (x - 1)] : 0.f;
        t.c2b = x > 0 ? costH[(size_t)y * rw + (x - 1)] : 0.f;
        t.c3a = x < rw - 1 ? costV[(size_t)(y - 1) * (rw + 1) + (x + 1)] : 0.f;
        t.c3b = costH[(size_t)y * rw + x];
    }
    return t;
}

// Step-major records for the dynamic programme: everything cell (s, i) needs besides the previous step's LDS state, so
// that the single-workgroup walk issues two coalesced loads per cell and step instead of seven scattered ones.
__global__ __launch_bounds__(256) void k_seam_pack(SeamGeom g, const float* costV, const float* costH, int horiz, int first, int nsteps, int n,
                                                   float4* ra, float2* rb) {
    const int i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (i >= n || k >= nsteps) return;
    const SeamStep t = seam_load_step(g, costV, costH, horiz, first + k, i);
    ra[(size_t)k * n + i] = make_float4(t.c1, t.c2a, t.c2b, t.c3a);
    rb[(size_t)k * n + i] = make_float2(t.c3b, __int_as_float(t.lab));
}

// One cell of one step from the previous step's LDS state and the cell's record; branch-free (clamped neighbour reads).
// min_element over pair<float, int> (S:878,908): the first minimum; equal costs keep the smaller step code, which is the
// order the candidates are tried in (strict <).
__device__ __forceinline__ void seam_cell(const float* pc, const unsigned char* pr, int i, int n, const float4& ta, const float2& tb, float& best, int& dir) {
    const int im = max(i - 1, 0), ip = min(i + 1, n - 1);
    // all six LDS reads first (one wait), then bitwise logic: no short-circuit branches around the reads
    const unsigned q1 = pr[i], q2 = pr[im], q3 = pr[ip];
    const float p1 = pc[i], p2 = pc[im], p3 = pc[ip];
    const bool lab = __float_as_int(tb.y) != 0;
    const bool r1 = lab & (q1 != 0), r2 = lab & (i > 0) & (q2 != 0), r3 = lab & (i < n - 1) & (q3 != 0);
    const float c1 = p1 + ta.x;
    const float c2 = p2 + ta.y + ta.z;
    const float c3 = p3 + ta.w + tb.x;
    dir = r1 ? 1 : 0;
    best = r1 ? c1 : 0.f;
    const bool t2 = r2 & ((dir == 0) | (c2 < best));
    best = t2 ? c2 : best; dir = t2 ? 2 : dir;
    const bool t3 = r3 & ((dir == 0) | (c3 < best));
    best = t3 ? c3 : best; dir = t3 ? 3 : dir;
}

// E = cells per thread whose records of the NEXT step are fetched while the current step is computed: unconditional
// loads at clamped indices, so that the compiler can count them (s_waitcnt vmcnt(k)) and the step itself only waits for
// LDS.  Cells beyond E * 1024 take the plain path at the end of each step.
template <int E>
__global__ __launch_bounds__(SEAM_NT) void k_seam_dp(const float4* ra, const float2* rb, int rw, int n, int horiz, int sx, int sy, int dx, int dy,
                                                     unsigned char* control, int* found) {
    extern __shared__ unsigned char smem[];
    // plain offset arithmetic on the LDS base (an array of pointers selected by `cur` turns into generic pointers: flat
    // loads, which count against vmcnt as well and would make every LDS read wait for the prefetched records)
    float* const cost = (float*)smem;                  // [2][n]
    unsigned char* const reach = smem + (size_t)8 * n;  // [2][n]
    for (int i = threadIdx.x; i < n; i += SEAM_NT) { cost[i] = 0.f; reach[i] = (i == (horiz ? sy : sx)) ? 1 : 0; }   // S:850-851
    const int first = (horiz ? sx : sy) + 1, last = horiz ? dx : dy;
    if (first > last) {
        __syncthreads();
        if (threadIdx.x == 0) *found = reach[horiz ? dy : dx] ? 1 : 0;
        return;
    }
    int ci[E];
    float4 na[E];
    float2 nb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        ci[e] = min((int)threadIdx.x + e * SEAM_NT, n - 1);
        na[e] = ra[ci[e]]; nb[e] = rb[ci[e]];
    }
    __syncthreads();
    int cur = 0;
    for (int s = first; s <= last; ++s) {
        const float* pc = cost + cur * n;
        const unsigned char* pr = reach + cur * n;
        float* nc = cost + (cur ^ 1) * n;
        unsigned char* nr = reach + (cur ^ 1) * n;
        const size_t row = (size_t)(s - first) * n, nrow = (size_t)(min(s + 1, last) - first) * n;
        float4 wa[E];
        float2 wb[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { wa[e] = na[e]; wb[e] = nb[e]; }
#pragma unroll
        for (int e = 0; e < E; ++e) { na[e] = ra[nrow + ci[e]]; nb[e] = rb[nrow + ci[e]]; }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = ci[e];
            float best; int dir;
            seam_cell(pc, pr, i, n, wa[e], wb[e], best, dir);
            if ((int)threadIdx.x + e * SEAM_NT < n) {
                if (dir) control[horiz ? (size_t)i * rw + s : (size_t)s * rw + i] = (unsigned char)dir;
                nc[i] = best;
                nr[i] = dir ? 255 : 0;
            }
        }
        for (int i = threadIdx.x + E * SEAM_NT; i < n; i += SEAM_NT) {
            float best; int dir;
            seam_cell(pc, pr, i, n, ra[row + i], rb[row + i], best, dir);
            if (dir) control[horiz ? (size_t)i * rw + s : (size_t)s * rw + i] = (unsigned char)dir;
            nc[i] = best;
            nr[i] = dir ? 255 : 0;
        }
        // LDS-only barrier: __syncthreads() would also drain the global queue (workgroup-scope release = vmcnt(0)), i.e.
        // wait every step for the control stores and for the NEXT step's records that were just requested
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur ^= 1;
    }
    if (threadIdx.x == 0) *found = reach[cur * n + (horiz ? dy : dx)] ? 1 : 0;   // S:918
}

// The same programme for wavefront steps too long for LDS (more than 15 360 cells: a component side beyond 15 K pixels): the previous
// step's cost / reachability ping-pong in global memory, one __syncthreads() (a workgroup-scope release / acquire) per step.  The
// reference has no such limit (S:806-957); this path keeps the call working, at global-memory speed.
__global__ __launch_bounds__(SEAM_NT) void k_seam_dp_global(const float4* ra, const float2* rb, int rw, int n, int horiz, int sx, int sy, int dx, int dy,
                                                            unsigned char* control, int* found, float* cost, unsigned char* reach) {
    for (int i = threadIdx.x; i < n; i += SEAM_NT) { cost[i] = 0.f; reach[i] = (i == (horiz ? sy : sx)) ? 1 : 0; }   // S:850-851
    const int first = (horiz ? sx : sy) + 1, last = horiz ? dx : dy;
    __syncthreads();
    int cur = 0;
    for (int s = first; s <= last; ++s) {
        const float* pc = cost + (size_t)cur * n;
        const unsigned char* pr = reach + (size_t)cur * n;
        float* nc = cost + (size_t)(cur ^ 1) * n;
        unsigned char* nr = reach + (size_t)(cur ^ 1) * n;
        const size_t row = (size_t)(s - first) * n;
        for (int i = threadIdx.x; i < n; i += SEAM_NT) {
            float best; int dir;
            seam_cell(pc, pr, i, n, ra[row + i], rb[row + i], best, dir);
            if (dir) control[horiz ? (size_t)i * rw + s : (size_t)s * rw + i] = (unsigned char)dir;
            nc[i] = best;
            nr[i] = dir ? 255 : 0;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (threadIdx.x == 0) *found = reach[(size_t)cur * n + (horiz ? dy : dx)] ? 1 : 0;   // S:918
}

struct SeamScratch { MatStage stages[3]; DevBuf scratch; int device = -1; };
SeamScratch& seam_scratch() {
    static thread_local SeamScratch* s = new SeamScratch();   // never destroyed at thread exit (the HIP runtime may be gone by then)
    return *s;
}

}  // namespace

namespace isx {
void seam_scratch_release() {
    SeamScratch& ss = seam_scratch();
    for (int i = 0; i < 3; ++i) ss.stages[i].buf.release();
    ss.scratch.release();
    ss.device = -1;
}
}  // namespace isx

extern "C" {

int isx_seam_estimate(const isx_mat* image1, const isx_mat* image2, int tl1_x, int tl1_y, int tl2_x, int tl2_y, int union_tl_x, int union_tl_y,
                      const isx_mat* labels, int label, const int roi[4], int p1_x, int p1_y, int p2_x, int p2_y,
                      int* seam_xy, int cap, int* seam_len, int* is_horizontal, int device, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_TRY(check_mat(image1, "seam_estimate: image1"));
    ISX_TRY(check_mat(image2, "seam_estimate: image2"));
    ISX_TRY(check_mat(labels, "seam_estimate: labels"));
    ISX_CHECK_ARG(roi != nullptr && seam_xy != nullptr && seam_len != nullptr && cap > 0, ISX_ERR_INVALID, "seam_estimate: null argument");
    ISX_CHECK_ARG((image1->type == ISX_32FC3 && image2->type == ISX_32FC3) || (image1->type == ISX_8UC3 && image2->type == ISX_8UC3), ISX_ERR_TYPE,
                  "seam_estimate: both images must have CV_32FC3 or CV_8UC3 type (S:745-746), got %s / %s", type_name(image1->type), type_name(image2->type));
    ISX_CHECK_ARG(labels->type == ISX_32SC1, ISX_ERR_TYPE, "seam_estimate: labels must be CV_32SC1, got %s", type_name(labels->type));
    const int rx = roi[0], ry = roi[1], rw = roi[2], rh = roi[3];
    ISX_CHECK_ARG(rw > 0 && rh > 0 && rx >= 0 && ry >= 0 && rx + rw <= labels->cols && ry + rh <= labels->rows, ISX_ERR_INVALID,
                  "seam_estimate: component rectangle (%d,%d) %dx%d lies outside the %dx%d label image", rx, ry, rw, rh, labels->cols, labels->rows);
    int sx = p1_x - rx, sy = p1_y - ry, dx = p2_x - rx, dy = p2_y - ry;                   // S:816-817
    ISX_CHECK_ARG(sx >= 0 && sx < rw && sy >= 0 && sy < rh && dx >= 0 && dx < rw && dy >= 0 && dy < rh, ISX_ERR_INVALID,
                  "seam_estimate: the seam tips must lie inside the component rectangle");
    // the component is an intersection component: its rectangle lies inside both images
    const int dx1 = union_tl_x - tl1_x, dy1 = union_tl_y - tl1_y, dx2 = union_tl_x - tl2_x, dy2 = union_tl_y - tl2_y;   // S:750-751
    ISX_CHECK_ARG(rx + dx1 >= 0 && ry + dy1 >= 0 && rx + rw + dx1 <= image1->cols && ry + rh + dy1 <= image1->rows && rx + dx2 >= 0 && ry + dy2 >= 0 &&
                      rx + rw + dx2 <= image2->cols && ry + rh + dy2 <= image2->rows,
                  ISX_ERR_INVALID, "seam_estimate: the component rectangle must lie inside both images (CV_Assert(states_[comp] & INTERS), S:736)");
    bool swapped = false;
    const bool horiz = std::abs(dx - sx) > std::abs(dy - sy);                                // S:827
    if (horiz ? sx > dx : sy > dy) { std::swap(sx, dx); std::swap(sy, dy); swapped = true; }  // S:829-842
    if (is_horizontal) *is_horizontal = horiz ? 1 : 0;
    const int n = horiz ? rh : rw;
    const bool lds_fits = (size_t)n * 10 <= 150 * 1024;   // one step's cost + reachability, both generations, in one workgroup's LDS

    ISX_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    // staging and scratch persist per host thread (grow-only; isx_dp_seam_release frees them: a finder calls this once per conflict)
    SeamScratch& ss = seam_scratch();
    if (ss.device != device) {   // the buffers live on one device: a call for another one starts afresh
        for (int i = 0; i < 3; ++i) ss.stages[i].buf.release();
        ss.scratch.release();
        ss.device = device;
    }
    MatStage &s1 = ss.stages[0], &s2 = ss.stages[1], &sl = ss.stages[2];
    ISX_TRY(s1.use_in(image1, st, "seam_estimate: image1"));
    ISX_TRY(s2.use_in(image2, st, "seam_estimate: image2"));
    ISX_TRY(sl.use_in(labels, st, "seam_estimate: labels"));
    SeamGeom g;
    g.img1 = (const unsigned char*)s1.d.data; g.step1 = s1.d.step;
    g.img2 = (const unsigned char*)s2.d.data; g.step2 = s2.d.step;
    g.labels = (const unsigned char*)sl.d.data; g.lstep = sl.d.step;
    g.uh = labels->rows; g.uw = labels->cols; g.label = label;
    g.rx = rx; g.ry = ry; g.rw = rw; g.rh = rh; g.dx1 = dx1; g.dy1 = dy1; g.dx2 = dx2; g.dy2 = dy2;
    DevBuf& scratch = ss.scratch;
    const size_t cv_b = ((size_t)rh * (rw + 1) * 4 + 255) & ~(size_t)255, ch_b = ((size_t)(rh + 1) * rw * 4 + 255) & ~(size_t)255,
                 ct_b = ((size_t)rh * rw + 255) & ~(size_t)255;
    const int first = (horiz ? sx : sy) + 1, nsteps = std::max((horiz ? dx : dy) - first + 1, 0);
    const size_t ra_b = ((size_t)std::max(nsteps, 1) * n * 16 + 255) & ~(size_t)255, rb_b = ((size_t)std::max(nsteps, 1) * n * 8 + 255) & ~(size_t)255;
    const size_t gl_b = lds_fits ? 0 : (((size_t)n * 10 + 255) & ~(size_t)255);
    ISX_TRY(scratch.reserve(cv_b + ch_b + ct_b + 256 + ra_b + rb_b + gl_b));
    float* costV = (float*)scratch.p;
    float* costH = (float*)((char*)scratch.p + cv_b);
    unsigned char* control = (unsigned char*)scratch.p + cv_b + ch_b;
    int* found = (int*)(control + ct_b);
    float4* ra = (float4*)((char*)found + 256);
    float2* rb = (float2*)((char*)ra + ra_b);
    dim3 grid(cdiv(rw + 1, 64), cdiv(rh + 1, 4));
    const double cbytes = (double)rw * rh * ((image1->type == ISX_8UC3 ? 6.0 : 24.0) + 4.0 + 8.0);
    if (image1->type == ISX_8UC3) ISX_LAUNCH("seam_costs", cbytes, st, (k_seam_costs<true>), grid, dim3(256), 0, g, costV, costH);
    else ISX_LAUNCH("seam_costs", cbytes, st, (k_seam_costs<false>), grid, dim3(256), 0, g, costV, costH);
#define ISX_SEAM_DP(E)                                                                                                                    \
    do {                                                                                                                                  \
        ISX_HIP(hipFuncSetAttribute((const void*)k_seam_dp<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));                  \
        ISX_LAUNCH("seam_dp", (double)nsteps * n * 25.0, st, (k_seam_dp<E>), dim3(1), dim3(SEAM_NT), (size_t)n * 10, (const float4*)ra,    \
                   (const float2*)rb, rw, n, horiz ? 1 : 0, sx, sy, dx, dy, control, found);                                            \
    } while (0)
    if (nsteps > 0)
        ISX_LAUNCH("seam_pack", (double)nsteps * n * 48.0, st, k_seam_pack, dim3(cdiv(n, 256), nsteps), dim3(256), 0, g, (const float*)costV, (const float*)costH,
                   horiz ? 1 : 0, first, nsteps, n, ra, rb);
    if (!lds_fits) {
        float* gcost = (float*)((char*)rb + rb_b);
        unsigned char* greach = (unsigned char*)gcost + (size_t)8 * n;
        ISX_LAUNCH("seam_dp", (double)nsteps * n * 25.0, st, k_seam_dp_global, dim3(1), dim3(SEAM_NT), 0, (const float4*)ra, (const float2*)rb, rw, n, horiz ? 1 : 0,
                   sx, sy, dx, dy, control, found, gcost, greach);
    } else if (n <= SEAM_NT) ISX_SEAM_DP(1);
    else if (n <= 2 * SEAM_NT) ISX_SEAM_DP(2);
    else ISX_SEAM_DP(4);
#undef ISX_SEAM_DP
    int h_found = 0;
    std::vector<unsigned char> ctl((size_t)rw * rh);
    ISX_HIP(hipMemcpyAsync(&h_found, found, sizeof(int), hipMemcpyDeviceToHost, st));
    ISX_HIP(hipMemcpyAsync(ctl.data(), control, ctl.size(), hipMemcpyDeviceToHost, st));
    ISX_HIP(hipStreamSynchronize(st));
    *seam_len = 0;
    if (!h_found) return ISX_OK;   // `return false`, S:918-919
    // restore the seam, S:921-948 (cells of rows the programme never reached are never visited here)
    std::vector<int> pts;
    int px = dx, py = dy;
    pts.push_back(px + rx); pts.push_back(py + ry);
    if (horiz) {
        for (; px != sx;) {
            const int c = ctl[(size_t)py * rw + px];
            if (c == 2) py--; else if (c == 3) py++;
            px--;
            pts.push_back(px + rx); pts.push_back(py + ry);
        }
    } else {
        for (; py != sy;) {
            const int c = ctl[(size_t)py * rw + px];
            if (c == 2) px--; else if (c == 3) px++;
            py--;
            pts.push_back(px + rx); pts.push_back(py + ry);
        }
    }
    const int len = (int)(pts.size() / 2);
    ISX_CHECK_ARG(len <= cap, ISX_ERR_SIZE, "seam_estimate: the seam has %d points, the buffer holds %d", len, cap);
    for (int i = 0; i < len; ++i) {
        const int j = swapped ? i : len - 1 - i;
        seam_xy[2 * i] = pts[2 * j]; seam_xy[2 * i + 1] = pts[2 * j + 1];
    }
    ISX_CHECK_ARG(seam_xy[0] == p1_x && seam_xy[1] == p1_y && seam_xy[2 * len - 2] == p2_x && seam_xy[2 * len - 1] == p2_y, ISX_ERR_INVALID,
                  "seam_estimate: the restored seam does not join the tips (CV_Assert, S:953-954)");
    *seam_len = len;
    return ISX_OK;
} ISX_EXIT("isx_seam_estimate")

}  // extern "C"
