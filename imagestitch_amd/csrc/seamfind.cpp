// seamfind.cpp — SURVEY §8(f) N1: the reference's in-tree DP seam finder as a whole
// (S = 动态规划法寻找最佳缝合线/.../动态规划法寻找最佳缝合线.cpp, itself a restatement of cv::detail::DpSeamFinder, costFunc_ COLOR):
//   find S:87-124 -> process S:127-193 -> findComponents S:196-308, findEdges S:311-392, resolveConflicts S:395-546
//   (hasOnlyOneNeighbor S:574-582, getSeamTips S:607-706 with closeToContour S:585-604 and cv::partition,
//   estimateSeam S:806-957, updateLabelsUsingSeam S:960-1093).
// The component / contour / graph logic is sequential host code over union-sized label images (a few MB); the cost
// maps and the dynamic programme of every estimateSeam call run on the GPU (isx_seam_estimate, seam.hip) straight from
// the caller's images — device-resident images are never copied.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <utility>
#include <vector>

#include "isx_internal.hpp"

using namespace isx;

namespace {

struct Pt { int x, y; };
enum { FIRST = 1, SECOND = 2, INTERS = 4 };

// first index in [e, x1) whose label differs from l (x1 if none): eight labels per step while they all match
inline int label_run_end(const int* row, int e, int x1, int l) {
    const unsigned long long pat = (unsigned)l | ((unsigned long long)(unsigned)l << 32);
    while (e + 8 <= x1) {
        unsigned long long v[4];
        memcpy(v, row + e, 32);
        if (((v[0] ^ pat) | (v[1] ^ pat) | (v[2] ^ pat) | (v[3] ^ pat)) != 0) break;
        e += 8;
    }
    while (e < x1 && row[e] == l) ++e;
    return e;
}
// first index in [x, x1) whose label IS l (x1 if none): eight labels per step while none matches (zero test on the 32-bit halves of v ^ pattern)
inline int label_skip_to(const int* row, int x, int x1, int l) {
    const unsigned long long pat = (unsigned)l | ((unsigned long long)(unsigned)l << 32);
    auto has_zero32 = [](unsigned long long t) { return ((t - 0x0000000100000001ull) & ~t & 0x8000000080000000ull) != 0; };
    while (x + 8 <= x1) {
        unsigned long long v[4];
        memcpy(v, row + x, 32);
        if (has_zero32(v[0] ^ pat) || has_zero32(v[1] ^ pat) || has_zero32(v[2] ^ pat) || has_zero32(v[3] ^ pat)) break;
        x += 8;
    }
    while (x < x1 && row[x] != l) ++x;
    return x;
}

// dst[i] = 0 wherever cond[i] != 0 (two different masks: no overlap, said so for the vectoriser)
inline void clear_where(unsigned char* __restrict dst, const unsigned char* __restrict cond, int n) {
    for (int i = 0; i < n; ++i) dst[i] = cond[i] ? (unsigned char)0 : dst[i];
}

struct Finder {
    int device = 0;
    hipStream_t stream = nullptr;
    int utlx = 0, utly = 0, uw = 0, uh = 0;
    // the two masks of the pair, read in place through the union's coordinates (zero outside a tile's rectangle): no union-sized copies
    struct MaskView {
        const unsigned char* p = nullptr; size_t step = 0; int ox = 0, oy = 0, rows = 0, cols = 0;   // (ox, oy) = the tile's corner in the union
        bool at(int y, int x) const {
            const int ty = y - oy, tx = x - ox;
            return (unsigned)ty < (unsigned)rows && (unsigned)tx < (unsigned)cols && p[(size_t)ty * step + tx] != 0;
        }
    };
    MaskView mask1_, mask2_;
    int ncomps = 0;
    // labels_ is only ever read (S:311-392, 607-706, 806-1093 and the write-back S:495-523) next to a contour pixel of some component, inside the
    // rectangle of an intersection component, or inside the intersection of the two tiles; everything that is not a contour test of the
    // first pass lies in the intersection rectangle widened by one pixel.  The label IMAGE therefore covers only that window (a quarter of a
    // 4K pair's union); outside it a label is looked up in the row runs of find_components, which nothing ever changes there.
    int wx0 = 0, wy0 = 0, ww = 0, wh = 0;
    std::vector<int> labels;
    std::vector<int> states;
    std::vector<Pt> tls, brs;
    std::vector<std::vector<Pt>> contours;
    std::set<std::pair<int, int>> edges;
    struct Run { int x0, x1, cls, id; };
    std::vector<Run> runs;
    std::vector<int> row_start;

    std::vector<int> seam_mask_;
    std::vector<unsigned short> seam_mask16_;
    int run_label(int y, int x) const {
        int lo = row_start[y], hi = row_start[y + 1];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (runs[mid].x1 <= x) lo = mid + 1; else hi = mid; }
        return (lo < row_start[y + 1] && runs[lo].x0 <= x) ? runs[lo].id : 0;
    }
    bool in_window(int y, int x) const { return (unsigned)(y - wy0) < (unsigned)wh && (unsigned)(x - wx0) < (unsigned)ww; }
    int& LW(int y, int x) { return labels[(size_t)(y - wy0) * ww + (x - wx0)]; }
    int L(int y, int x) const { return in_window(y, x) ? labels[(size_t)(y - wy0) * ww + (x - wx0)] : run_label(y, x); }

    // contour1mask_ / contour2mask_ (S:168-186) are only ever read through closeToContour (S:585-604) at the contour pixels
    // of one component: evaluated on demand from the masks instead of being materialised for the whole union
    bool is_mask_contour(const MaskView& m, int y, int x) const {
        return m.at(y, x) && ((x == 0 || !m.at(y, x - 1)) || (x == uw - 1 || !m.at(y, x + 1)) || (y == 0 || !m.at(y - 1, x)) || (y == uh - 1 || !m.at(y + 1, x)));
    }

    // maximal runs of non-zero bytes of p[0, n), shifted by xoff, appended to out: eight pixels per step on the per-byte "non-zero" flags
    static void nonzero_runs(const unsigned char* p, int n, int xoff, std::vector<std::pair<int, int>>& out) {
        const unsigned long long HI = 0x8080808080808080ull, LO = 0x7f7f7f7f7f7f7f7full;
        auto flags = [&](unsigned long long v) { return (((v & LO) + LO) | v) & HI; };   // bit 7 of every non-zero byte
        int x = 0;
        while (x < n) {
            // first non-zero byte at or after x: 32 bytes per step while all are zero
            while (x + 32 <= n) { unsigned long long v[4]; memcpy(v, p + x, 32); if (v[0] | v[1] | v[2] | v[3]) break; x += 32; }
            while (x + 8 <= n) { unsigned long long v; memcpy(&v, p + x, 8); if (v) { x += __builtin_ctzll(v) >> 3; break; } x += 8; }
            while (x < n && !p[x]) ++x;
            if (x >= n) break;
            // first zero byte after x: 32 bytes per step while none is zero (little-endian: the lowest clear flag is the first zero pixel)
            int e = x + 1;
            while (e + 32 <= n) { unsigned long long v[4]; memcpy(v, p + e, 32); if ((flags(v[0]) & flags(v[1]) & flags(v[2]) & flags(v[3])) != HI) break; e += 32; }
            while (e + 8 <= n) {
                unsigned long long v; memcpy(&v, p + e, 8);
                const unsigned long long nz = flags(v);
                if (nz != HI) { e += __builtin_ctzll(~nz & HI) >> 3; goto have_end; }
                e += 8;
            }
            while (e < n && p[e]) ++e;
        have_end:
            out.push_back({x + xoff, e + xoff});
            x = e;
        }
    }

    // S:196-308.  The reference scans the union in raster order and flood-fills (4-connectivity, equal class) from the
    // first pixel of every component it meets.  Same numbering without touching every pixel: row runs of equal class,
    // union-find over vertically overlapping runs of the same class, components numbered by their first run in raster
    // order (= their first pixel); bounding boxes from the runs; contour pixels = run ends plus the interior pixels not
    // covered by the same component in the row above or below (interval arithmetic on the sorted run lists).
    void find_components() {
        labels.assign((size_t)ww * wh, 0);
        states.clear(); tls.clear(); brs.clear(); contours.clear();
        runs.clear();
        row_start.assign((size_t)uh + 1, 0);
        std::vector<std::pair<int, int>> ra, rb;
        for (int y = 0; y < uh; ++y) {
            row_start[y] = (int)runs.size();
            ra.clear(); rb.clear();
            const int y1 = y - mask1_.oy, y2 = y - mask2_.oy;
            if ((unsigned)y1 < (unsigned)mask1_.rows) nonzero_runs(mask1_.p + (size_t)y1 * mask1_.step, mask1_.cols, mask1_.ox, ra);
            if ((unsigned)y2 < (unsigned)mask2_.rows) nonzero_runs(mask2_.p + (size_t)y2 * mask2_.step, mask2_.cols, mask2_.ox, rb);
            // maximal runs of constant (mask1 != 0, mask2 != 0): sweep over the two sorted run lists
            size_t ia = 0, ib = 0;
            int x = 0;
            while (ia < ra.size() || ib < rb.size()) {
                const int sa = ia < ra.size() ? ra[ia].first : INT_MAX, sb = ib < rb.size() ? rb[ib].first : INT_MAX;
                const bool ca = sa <= x, cb = sb <= x;
                if (!ca && !cb) { x = std::min(sa, sb); continue; }
                // the class holds up to the next boundary of either list
                int e = INT_MAX;
                if (ca) e = std::min(e, ra[ia].second); else e = std::min(e, sa);
                if (cb) e = std::min(e, rb[ib].second); else e = std::min(e, sb);
                const int cls = (ca && cb) ? INTERS : (ca ? FIRST : SECOND);
                if (!runs.empty() && (int)runs.size() > row_start[y] && runs.back().x1 == x && runs.back().cls == cls) runs.back().x1 = e;
                else runs.push_back({x, e, cls, 0});
                x = e;
                if (ca && ra[ia].second == e) ++ia;
                if (cb && rb[ib].second == e) ++ib;
            }
        }
        row_start[uh] = (int)runs.size();
        std::vector<int> parent(runs.size());
        for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int)i;
        auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
        for (int y = 1; y < uh; ++y) {   // runs of the same class that overlap in x are 4-connected
            int i = row_start[y - 1], j = row_start[y];
            const int ie = row_start[y], je = row_start[y + 1];
            while (i < ie && j < je) {
                if (runs[i].cls == runs[j].cls && runs[i].x0 < runs[j].x1 && runs[j].x0 < runs[i].x1) {
                    const int pi = find(i), pj = find(j);
                    if (pi != pj) parent[std::max(pi, pj)] = std::min(pi, pj);
                }
                if (runs[i].x1 <= runs[j].x1) ++i; else ++j;
            }
        }
        std::vector<int> comp_of(runs.size(), 0);
        ncomps = 0;
        for (int y = 0; y < uh; ++y)
            for (int k = row_start[y]; k < row_start[y + 1]; ++k) {
                const int r = find(k);
                if (!comp_of[r]) {   // first run of the component in raster order: its first pixel is the flood-fill seed
                    comp_of[r] = ++ncomps;
                    states.push_back(runs[k].cls);
                    tls.push_back({runs[k].x0, y});
                    brs.push_back({runs[k].x0 + 1, y + 1});
                    contours.emplace_back();
                }
                const int l = comp_of[r];
                runs[k].id = l;
                if ((unsigned)(y - wy0) < (unsigned)wh) {   // the label image covers the window only
                    const int a = std::max(runs[k].x0, wx0), b = std::min(runs[k].x1, wx0 + ww);
                    if (a < b) std::fill(labels.begin() + (size_t)(y - wy0) * ww + (a - wx0), labels.begin() + (size_t)(y - wy0) * ww + (b - wx0), l);
                }
                Pt& tl = tls[l - 1]; Pt& br = brs[l - 1];
                tl.x = std::min(tl.x, runs[k].x0); tl.y = std::min(tl.y, y);
                br.x = std::max(br.x, runs[k].x1); br.y = std::max(br.y, y + 1);
            }
        // contour pixels, per component in raster order (S:249-253)
        std::vector<std::pair<int, int>> cov;   // pixels of the run covered by the same component above AND below
        for (int y = 0; y < uh; ++y)
            for (int k = row_start[y]; k < row_start[y + 1]; ++k) {
                const Run& r = runs[k];
                cov.clear();
                if (y > 0 && y < uh - 1) {
                    int i = row_start[y - 1], j = row_start[y + 1];
                    const int ie = row_start[y], je = row_start[y + 2 <= uh ? y + 2 : uh];
                    while (i < ie && j < je) {
                        if (runs[i].id != r.id || runs[i].x1 <= r.x0) { ++i; continue; }
                        if (runs[j].id != r.id || runs[j].x1 <= r.x0) { ++j; continue; }
                        if (runs[i].x0 >= r.x1 || runs[j].x0 >= r.x1) break;
                        const int lo = std::max(std::max(runs[i].x0, runs[j].x0), r.x0), hi = std::min(std::min(runs[i].x1, runs[j].x1), r.x1);
                        if (lo < hi) cov.push_back({lo, hi});
                        if (runs[i].x1 <= runs[j].x1) ++i; else ++j;
                    }
                }
                std::vector<Pt>& out = contours[r.id - 1];
                int x = r.x0;
                for (const auto& c : cov) {   // covered interior pixels are not contour pixels — except the two run ends
                    const int lo = std::max(c.first, r.x0 + 1), hi = std::min(c.second, r.x1 - 1);
                    if (lo >= hi) continue;
                    for (; x < lo; ++x) out.push_back({x, y});
                    x = hi;
                }
                for (; x < r.x1; ++x) out.push_back({x, y});
            }
    }

    void find_edges() {   // S:311-392
        std::map<std::pair<int, int>, int> wedges;
        for (int ci = 0; ci < ncomps; ++ci)
            for (const Pt& p : contours[ci]) {
                const int x = p.x, y = p.y, l = ci + 1;
                const int nb[4] = {x > 0 ? L(y, x - 1) : 0, y > 0 ? L(y - 1, x) : 0, x < uw - 1 ? L(y, x + 1) : 0, y < uh - 1 ? L(y + 1, x) : 0};
                for (int k = 0; k < 4; ++k)
                    if (nb[k] && nb[k] != l) { wedges[{ci, nb[k] - 1}]++; wedges[{nb[k] - 1, ci}]++; }
            }
        edges.clear();
        for (const auto& e : wedges)
            if (e.second > 0) edges.insert(e.first);
    }

    // phase times of resolveConflicts (ISX_SEAMFIND_TIMING)
    double t_tips = 0, t_est = 0, t_upd = 0, t_rec = 0, t_wb = 0;
    static double tnow() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    bool has_only_one_neighbor(int comp) const {   // S:574-582
        auto b = edges.lower_bound({comp, INT_MIN});
        auto e = edges.upper_bound({comp, INT_MAX});
        return b != e && std::next(b) == e;
    }

    bool close_to_contour(int y, int x, const MaskView& m) const {   // S:585-604 on the contour mask of m
        for (int dy = -2; dy <= 2; ++dy)
            if (y + dy >= 0 && y + dy < uh)
                for (int dx = -2; dx <= 2; ++dx)
                    if (x + dx >= 0 && x + dx < uw && is_mask_contour(m, y + dy, x + dx)) return true;
        return false;
    }

    // cv::partition(points, labels, ClosePoints(10)) S:44-57,629: classes numbered by their first member
    static std::vector<int> partition(const std::vector<Pt>& pts, int min_dist) {
        const int n = (int)pts.size();
        std::vector<int> parent(n), out(n), cls(n, -1);
        for (int i = 0; i < n; ++i) parent[i] = i;
        auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const int d2 = (pts[i].x - pts[j].x) * (pts[i].x - pts[j].x) + (pts[i].y - pts[j].y) * (pts[i].y - pts[j].y);
                if (d2 < min_dist * min_dist) { const int a = find(i), b = find(j); if (a != b) parent[b] = a; }
            }
        int nc = 0;
        for (int i = 0; i < n; ++i) { const int r = find(i); if (cls[r] < 0) cls[r] = nc++; out[i] = cls[r]; }
        return out;
    }

    bool get_seam_tips(int comp1, int comp2, Pt& p1, Pt& p2) const {   // S:607-706
        const int l2 = comp2 + 1;
        std::vector<Pt> special;
        for (const Pt& p : contours[comp1]) {
            const int x = p.x, y = p.y;
            if (close_to_contour(y, x, mask1_) && close_to_contour(y, x, mask2_) &&
                ((x > 0 && L(y, x - 1) == l2) || (y > 0 && L(y - 1, x) == l2) || (x < uw - 1 && L(y, x + 1) == l2) || (y < uh - 1 && L(y + 1, x) == l2)))
                special.push_back(p);
        }
        if (special.size() < 2) return false;
        const std::vector<int> lab = partition(special, 10);
        const int nlabels = *std::max_element(lab.begin(), lab.end()) + 1;
        if (nlabels < 2) return false;
        std::vector<long long> sx(nlabels, 0), sy(nlabels, 0);
        std::vector<std::vector<Pt>> pts(nlabels);
        for (size_t i = 0; i < special.size(); ++i) { sx[lab[i]] += special[i].x; sy[lab[i]] += special[i].y; pts[lab[i]].push_back(special[i]); }
        auto rnd = [](double v) { return std::nearbyint(v); };   // cvRound(double)
        int idx[2] = {-1, -1};
        double max_dist = -1.7976931348623157e308;
        for (int i = 0; i < nlabels - 1; ++i)
            for (int j = i + 1; j < nlabels; ++j) {
                const double s1 = (double)pts[i].size(), s2 = (double)pts[j].size();
                const double cx1 = rnd(sx[i] / s1), cy1 = rnd(sy[i] / s1), cx2 = rnd(sx[j] / s2), cy2 = rnd(sy[j] / s2);
                const double dist = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
                if (dist > max_dist) { max_dist = dist; idx[0] = i; idx[1] = j; }
            }
        Pt p[2];
        for (int i = 0; i < 2; ++i) {
            const std::vector<Pt>& g = pts[idx[i]];
            const double size = (double)g.size(), cx = rnd(sx[idx[i]] / size), cy = rnd(sy[idx[i]] / size);
            size_t closest = g.size();
            double min_dist = 1.7976931348623157e308;
            for (size_t j = 0; j < g.size(); ++j) {
                const double dist = (g[j].x - cx) * (g[j].x - cx) + (g[j].y - cy) * (g[j].y - cy);
                if (dist < min_dist) { min_dist = dist; closest = j; }
            }
            p[i] = g[closest];
        }
        p1 = p[0]; p2 = p[1];
        return true;
    }

    // estimateSeam S:806-957 on the GPU.  Only the component's rectangle of labels_ is handed over (a view into the label
    // image with the union origin shifted accordingly): every labels_ read outside Rect(tls_, brs_) is "not this
    // component" anyway, and the rectangle is a fraction of the union.
    int estimate_seam(const isx_mat* image1, const isx_mat* image2, Pt tl1, Pt tl2, int comp, Pt p1, Pt p2, std::vector<Pt>& seam, bool& horiz, bool& found) {
        const int rx = tls[comp].x, ry = tls[comp].y, rw = brs[comp].x - rx, rh = brs[comp].y - ry;
        const int roi[4] = {0, 0, rw, rh};
        isx_mat lab;
        lab.data = &LW(ry, rx); lab.rows = rh; lab.cols = rw; lab.type = ISX_32SC1; lab.step = (size_t)ww * 4; lab.device = -1;
        std::vector<int> xy((size_t)2 * (rw + rh + 2));
        int len = 0, h = 0;
        ISX_TRY(isx_seam_estimate(image1, image2, tl1.x, tl1.y, tl2.x, tl2.y, utlx + rx, utly + ry, &lab, comp + 1, roi, p1.x - rx, p1.y - ry, p2.x - rx, p2.y - ry,
                                  xy.data(), rw + rh + 2, &len, &h, device, stream));
        seam.resize(len);
        for (int i = 0; i < len; ++i) seam[i] = {xy[2 * i] + rx, xy[2 * i + 1] + ry};
        horiz = h != 0;
        found = len > 0;
        return ISX_OK;
    }

    // S:960-1093.  The work image `mask` (the component's rectangle; 255 = contour / seam pixel, else the number of the pixel's region)
    // holds T = 16-bit values while the regions number fewer than 65 535 - the same numbers, so also the same collision of region 255 with
    // the marker that the reference's int image has; more regions than that: the caller runs it again with T = int.
    template <class T>
    bool update_labels_t(std::vector<T>& mask, int comp1, int comp2, const std::vector<Pt>& seam, bool horiz) {
        const Pt tl = tls[comp1], br = brs[comp1];
        const int h = br.y - tl.y, w = br.x - tl.x;
        mask.assign((size_t)h * w, 0);
        auto M = [&](int y, int x) -> T& { return mask[(size_t)y * w + x]; };
        for (const Pt& p : contours[comp1]) M(p.y - tl.y, p.x - tl.x) = 255;
        for (const Pt& p : seam) M(p.y - tl.y, p.x - tl.x) = 255;
        const int l1 = comp1 + 1, l2 = comp2 + 1;
        // S:976-981: flood fills of the zero pixels of `mask`, seeded in raster order at zero pixels that carry label l1 -
        // done on row runs of zeros (union-find over vertically overlapping runs); a zero region without any l1 pixel is
        // never seeded and stays 0
        int nc = 0;
        struct ZRun { int y, x0, x1; };
        std::vector<ZRun> zr;
        std::vector<int> rs((size_t)h + 1, 0);
        constexpr int PER8 = 8 / (int)sizeof(T);
        for (int y = 0; y < h; ++y) {
            rs[y] = (int)zr.size();
            const T* row = &mask[(size_t)y * w];
            int x = 0;
            while (x < w) {
                while (x < w && row[x]) ++x;
                if (x >= w) break;
                int e = x + 1;
                while (e + PER8 <= w) { unsigned long long v; memcpy(&v, row + e, 8); if (v) break; e += PER8; }   // eight bytes of zeros per step
                while (e < w && !row[e]) ++e;
                zr.push_back({y, x, e});
                x = e;
            }
        }
        rs[h] = (int)zr.size();
        std::vector<int> parent(zr.size());
        for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int)i;
        auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
        for (int y = 1; y < h; ++y) {
            int i = rs[y - 1], j = rs[y];
            const int ie = rs[y], je = rs[y + 1];
            while (i < ie && j < je) {
                if (zr[i].x0 < zr[j].x1 && zr[j].x0 < zr[i].x1) { const int a = find(i), b = find(j); if (a != b) parent[std::max(a, b)] = std::min(a, b); }
                if (zr[i].x1 <= zr[j].x1) ++i; else ++j;
            }
        }
        std::vector<int> number(zr.size(), 0);
        for (size_t k = 0; k < zr.size(); ++k) {   // raster order of the runs = raster order of their first l1 pixels
            const int r = find((int)k);
            if (number[r]) continue;
            const int* lrow = &LW(zr[k].y + tl.y, tl.x);     // an intersection component's rectangle lies inside the window
            for (int x = zr[k].x0; x < zr[k].x1; ++x)
                if (lrow[x] == l1) {
                    if (sizeof(T) < sizeof(int) && nc >= 65534) return false;
                    number[r] = ++nc;
                    break;
                }
        }
        for (size_t k = 0; k < zr.size(); ++k) {
            const int v = number[find((int)k)];
            number[k] = v;                        // from here on: the region number of run k itself
            if (v) std::fill(mask.begin() + (size_t)zr[k].y * w + zr[k].x0, mask.begin() + (size_t)zr[k].y * w + zr[k].x1, (T)v);
        }
        static const int dx[] = {-1, +1, 0, 0, -1, +1, -1, +1}, dy[] = {0, 0, -1, +1, -1, -1, +1, +1};
        for (const Pt& p : contours[comp1]) {
            const int x = p.x - tl.x, y = p.y - tl.y;
            bool ok = false;
            for (int j = 0; j < 8; ++j) {
                const int c = x + dx[j], r = y + dy[j];
                if (c >= 0 && c < w && r >= 0 && r < h && M(r, c) && M(r, c) != 255) { ok = true; M(y, x) = M(r, c); }
            }
            if (!ok) M(y, x) = 0;
        }
        for (const Pt& p : seam) {
            const int x = p.x - tl.x, y = p.y - tl.y;
            if (horiz) M(y, x) = (y < h - 1 && M(y + 1, x) && M(y + 1, x) != 255) ? M(y + 1, x) : (T)0;
            else M(y, x) = (x < w - 1 && M(y, x + 1) && M(y, x + 1) != 255) ? M(y, x + 1) : (T)0;
        }
        std::map<int, int> connect2, connect_other;
        for (int i = 1; i <= nc; ++i) { connect2[i] = 0; connect_other[i] = 0; }
        for (const Pt& p : contours[comp1]) {
            const int x = p.x, y = p.y;
            if ((x > 0 && L(y, x - 1) == l2) || (y > 0 && L(y - 1, x) == l2) || (x < uw - 1 && L(y, x + 1) == l2) || (y < uh - 1 && L(y + 1, x) == l2))
                connect2[M(y - tl.y, x - tl.x)]++;
            if ((x > 0 && L(y, x - 1) != l1 && L(y, x - 1) != l2) || (y > 0 && L(y - 1, x) != l1 && L(y - 1, x) != l2) ||
                (x < uw - 1 && L(y, x + 1) != l1 && L(y, x + 1) != l2) || (y < uh - 1 && L(y + 1, x) != l1 && L(y + 1, x) != l2))
                connect_other[M(y - tl.y, x - tl.x)]++;
        }
        // the reference indexes isAdjComp(ncomps + 1) with every key of connect2, which may hold 0 (a contour pixel that lost
        // its region): sized to hold all keys here
        int maxkey = nc;
        for (const auto& kv : connect2) maxkey = std::max(maxkey, kv.first);
        std::vector<int> is_adj((size_t)maxkey + 1, 0);
        const double len = (double)contours[comp1].size();
        for (const auto& kv : connect2) {
            int res = 0;
            if (kv.second / len > 0.05) {
                auto sub = connect_other.find(kv.first);
                if (sub != connect_other.end() && (sub->second / len < 0.1)) res = 1;
            }
            if (kv.first >= 0) is_adj[kv.first] = res;
        }
        // S:1081-1092 walks the whole rectangle; a pixel of it is either in a zero run (its value = the run's region number) or a contour / seam
        // pixel (its value as the two loops above left it): the same assignments from the runs and the two point lists
        auto adjacent = [&](int m) { return m && m <= maxkey && is_adj[m]; };
        for (size_t k = 0; k < zr.size(); ++k)
            if (adjacent(number[k])) std::fill(&LW(zr[k].y + tl.y, zr[k].x0 + tl.x), &LW(zr[k].y + tl.y, zr[k].x0 + tl.x) + (zr[k].x1 - zr[k].x0), l2);
        for (const Pt& p : contours[comp1])
            if (adjacent(M(p.y - tl.y, p.x - tl.x))) LW(p.y, p.x) = l2;
        for (const Pt& p : seam)
            if (adjacent(M(p.y - tl.y, p.x - tl.x))) LW(p.y, p.x) = l2;
        return true;
    }
    void update_labels_using_seam(int comp1, int comp2, const std::vector<Pt>& seam, bool horiz) {
        if (!update_labels_t<unsigned short>(seam_mask16_, comp1, comp2, seam, horiz)) update_labels_t<int>(seam_mask_, comp1, comp2, seam, horiz);
    }

    // S:457-487: bounding box and contour pixels of label l, scanning the component's OLD rectangle only (pixels of l
    // outside it are ignored, as in the reference) while the neighbour tests look at the whole label image.  Row runs of
    // `== l` over the full width, contour pixels = true run ends + interior pixels not covered by l above and below.
    void recompute_region(int c, int l) {
        const int x0 = tls[c].x, x1 = brs[c].x, y0 = tls[c].y, y1 = brs[c].y;
        tls[c] = {INT_MAX, INT_MAX};
        brs[c] = {INT_MIN, INT_MIN};
        contours[c].clear();
        if (x0 >= x1 || y0 >= y1) return;
        // runs of label l in row y, clipped to the old rectangle widened by one pixel: everything below only asks whether a
        // pixel of [x0, x1) has a same-label neighbour left / right / above / below, which that window decides
        const int qx0 = std::max(x0 - 1, 0), qx1 = std::min(x1 + 1, uw);
        auto runs_of = [&](int y, std::vector<std::pair<int, int>>& out) {
            out.clear();
            if (y < 0 || y >= uh) return;
            if ((unsigned)(y - wy0) < (unsigned)wh && qx0 >= wx0 && qx1 <= wx0 + ww) {   // inside the label image (every intersection component is)
                const int* row = &labels[(size_t)(y - wy0) * ww];
                int x = qx0 - wx0;
                const int xe = qx1 - wx0;
                while (x < xe) {
                    x = label_skip_to(row, x, xe, l);
                    if (x >= xe) break;
                    const int e = label_run_end(row, x + 1, xe, l);
                    out.push_back({x + wx0, e + wx0});
                    x = e;
                }
                return;
            }
            int x = qx0;
            while (x < qx1) {
                while (x < qx1 && L(y, x) != l) ++x;
                if (x >= qx1) break;
                int e = x + 1;
                while (e < qx1 && L(y, e) == l) ++e;
                out.push_back({x, e});
                x = e;
            }
        };
        std::vector<std::pair<int, int>> up, cur, dn, cov;
        runs_of(y0 - 1, up);
        runs_of(y0, cur);
        for (int y = y0; y < y1; ++y) {
            runs_of(y + 1, dn);
            for (const auto& r : cur) {
                const int a = std::max(r.first, x0), b = std::min(r.second, x1);   // the part of the run inside the old rectangle
                if (a >= b) continue;
                tls[c].x = std::min(tls[c].x, a); tls[c].y = std::min(tls[c].y, y);
                brs[c].x = std::max(brs[c].x, b); brs[c].y = std::max(brs[c].y, y + 1);
                cov.clear();
                size_t i = 0, j = 0;
                while (i < up.size() && j < dn.size()) {
                    const int lo = std::max(std::max(up[i].first, dn[j].first), r.first), hi = std::min(std::min(up[i].second, dn[j].second), r.second);
                    if (lo < hi) cov.push_back({lo, hi});
                    if (up[i].second <= dn[j].second) ++i; else ++j;
                }
                int x = a;
                for (const auto& cv : cov) {   // covered pixels that are not a true run end (image border included) are interior
                    const int lo = std::max(std::max(cv.first, r.first + 1), a), hi = std::min(std::min(cv.second, r.second - 1), b);
                    if (lo >= hi) continue;
                    for (; x < lo; ++x) contours[c].push_back({x, y});
                    x = hi;
                }
                for (; x < b; ++x) contours[c].push_back({x, y});
            }
            up.swap(cur);
            cur.swap(dn);
        }
    }

    int resolve_conflicts(const isx_mat* image1, const isx_mat* image2, Pt tl1, Pt tl2, unsigned char* mask1, size_t step1, int rows1, int cols1,
                          unsigned char* mask2, size_t step2, int rows2, int cols2) {   // S:395-546
        bool has_conflict = true;
        while (has_conflict) {
            int c1 = 0, c2 = 0;
            has_conflict = false;
            for (const auto& e : edges) {
                c1 = e.first; c2 = e.second;
                if ((states[c1] & INTERS) && (states[c1] & (~INTERS)) != states[c2]) { has_conflict = true; break; }
            }
            if (!has_conflict) break;
            const int l1 = c1 + 1, l2 = c2 + 1;
            if (has_only_one_neighbor(c1)) {
                for (int y = tls[c1].y; y < brs[c1].y; ++y)
                    for (int x = tls[c1].x; x < brs[c1].x; ++x)
                        if (L(y, x) == l1) LW(y, x) = l2;
                states[c1] = states[c2] == FIRST ? SECOND : FIRST;
            } else {
                Pt p1, p2;
                const double ta = tnow();
                const bool tips = get_seam_tips(c1, c2, p1, p2);
                t_tips += tnow() - ta;
                if (tips) {
                    std::vector<Pt> seam;
                    bool horiz = false, found = false;
                    const double tb = tnow();
                    ISX_TRY(estimate_seam(image1, image2, tl1, tl2, c1, p1, p2, seam, horiz, found));
                    const double tc = tnow();
                    if (found) update_labels_using_seam(c1, c2, seam, horiz);
                    t_est += tc - tb; t_upd += tnow() - tc;
                }
                states[c1] = states[c2] == FIRST ? (INTERS | SECOND) : (INTERS | FIRST);
            }
            // S:457-487 recomputes the rectangle and contour of both labels.  Those of c2 are never read again unless c2 is itself an
            // intersection component: only such a component is ever the first of a conflicting edge (its neighbours are FIRST- or SECOND-only
            // components, whose states never change), and tls_ / brs_ / contours_ are read for the first component only - so the scan of a
            // whole tile's rectangle that the second call would be is skipped, with nothing observable changed
            const double td = tnow();
            recompute_region(c1, l1);
            if (states[c2] & INTERS) recompute_region(c2, l2);
            t_rec += tnow() - td;
            edges.erase({c1, c2});
            edges.erase({c2, c1});
        }
        const int dx1 = utlx - tl1.x, dy1 = utly - tl1.y, dx2 = utlx - tl2.x, dy2 = utly - tl2.y;   // S:495-523
        // Both loops only touch pixels that lie inside BOTH tiles (the other tile's mask is read at the same union position),
        // i.e. the intersection rectangle; per label one flag byte instead of a states[] lookup per pixel.  The second loop
        // reads mask2 as the first one left it, as S:509-523 does.
        std::vector<unsigned char> is_first(states.size() + 1, 0), is_second(states.size() + 1, 0);
        for (size_t i = 0; i < states.size(); ++i) { is_first[i + 1] = (states[i] & FIRST) ? 1 : 0; is_second[i + 1] = (states[i] & SECOND) ? 1 : 0; }
        const int ux0 = std::max(tl1.x, tl2.x) - utlx, ux1 = std::min(tl1.x + cols1, tl2.x + cols2) - utlx;   // intersection in union coordinates
        const int uy0 = std::max(tl1.y, tl2.y) - utly, uy1 = std::min(tl1.y + rows1, tl2.y + rows2) - utly;
        const double te = tnow();
        // one pass: at a pixel the second loop of the reference (S:509-523) reads mask2 as its first loop (S:495-507) left it AT THAT PIXEL.
        // Row by row over the runs of equal label (eight labels per step), so that what is done per pixel is a byte select
        for (int uy = uy0; uy < uy1; ++uy) {
            const int* lrow = &labels[(size_t)(uy - wy0) * ww] - wx0;   // indexed by the union x; the intersection rectangle lies inside the window
            unsigned char* m1 = mask1 + (size_t)(uy + dy1) * step1 + dx1;
            unsigned char* m2 = mask2 + (size_t)(uy + dy2) * step2 + dx2;
            int ux = ux0;
            while (ux < ux1) {
                const int l = lrow[ux];
                const int e = label_run_end(lrow, ux + 1, ux1, l);
                if (is_first[l]) clear_where(m2 + ux, m1 + ux, e - ux);
                if (is_second[l]) clear_where(m1 + ux, m2 + ux, e - ux);
                ux = e;
            }
        }
        t_wb += tnow() - te;
        return ISX_OK;
    }

    int process(const isx_mat* image1, const isx_mat* image2, Pt tl1, Pt tl2, unsigned char* mask1, size_t step1, unsigned char* mask2, size_t step2) {   // S:127-193
        const int r1 = image1->rows, c1 = image1->cols, r2 = image2->rows, c2 = image2->cols;
        const int itlx = std::max(tl1.x, tl2.x), itly = std::max(tl1.y, tl2.y);
        const int ibrx = std::min(tl1.x + c1, tl2.x + c2), ibry = std::min(tl1.y + r1, tl2.y + r2);
        if (itlx >= ibrx || itly >= ibry) return ISX_OK;   // there are no conflicts
        utlx = std::min(tl1.x, tl2.x); utly = std::min(tl1.y, tl2.y);
        uw = std::max(tl1.x + c1, tl2.x + c2) - utlx;
        uh = std::max(tl1.y + r1, tl2.y + r2) - utly;
        mask1_.p = mask1; mask1_.step = step1; mask1_.ox = tl1.x - utlx; mask1_.oy = tl1.y - utly; mask1_.rows = r1; mask1_.cols = c1;
        mask2_.p = mask2; mask2_.step = step2; mask2_.ox = tl2.x - utlx; mask2_.oy = tl2.y - utly; mask2_.rows = r2; mask2_.cols = c2;
        // the label image's window: the intersection rectangle widened by one pixel, clipped to the union
        wx0 = std::max(itlx - utlx - 1, 0); wy0 = std::max(itly - utly - 1, 0);
        ww = std::min(ibrx - utlx + 1, uw) - wx0; wh = std::min(ibry - utly + 1, uh) - wy0;
        static const bool tm = getenv("ISX_SEAMFIND_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        t_tips = t_est = t_upd = t_rec = t_wb = 0;
        double t1 = now();
        find_components();
        double t2 = now();
        find_edges();
        double t3 = now();
        int rc = resolve_conflicts(image1, image2, tl1, tl2, mask1, step1, r1, c1, mask2, step2, r2, c2);
        if (tm) fprintf(stderr, "seamfind: components %.2f ms, edges %.2f ms, resolve %.2f ms (tips %.2f, estimateSeam %.2f, label update %.2f, region recompute %.2f, mask write-back %.2f)\n",
                        t2 - t1, t3 - t2, now() - t3, t_tips, t_est, t_upd, t_rec, t_wb);
        return rc;
    }
};

Finder& finder() {
    static thread_local Finder* f = new Finder();
    return *f;
}
struct FinderStages { std::vector<std::unique_ptr<MatStage>> img; int device = -1; };
FinderStages& finder_stages() {
    static thread_local FinderStages* s = new FinderStages();
    return *s;
}

}  // namespace

namespace isx { void seam_scratch_release(); }

extern "C" {

int isx_dp_seam_release(void) ISX_ENTRY {
    clear_error();
    Finder& f = finder();
    f = Finder();                       // labels, union masks, contours, the seam mask: back to empty vectors
    finder_stages().img.clear();        // the staged images
    finder_stages().device = -1;
    isx::seam_scratch_release();        // cost maps, DP records and the staging of isx_seam_estimate
    return ISX_OK;
} ISX_EXIT("isx_dp_seam_release")

int isx_dp_seam_find(int num_images, const isx_mat* images, const int* corners_xy, isx_mat* masks, int device, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(num_images >= 0 && (num_images == 0 || (images && corners_xy && masks)), ISX_ERR_INVALID, "dp_seam_find: null argument");
    if (num_images == 0) return ISX_OK;   // S:95-96
    for (int i = 0; i < num_images; ++i) {
        ISX_TRY(check_mat(&images[i], "dp_seam_find: image"));
        ISX_TRY(check_mat(&masks[i], "dp_seam_find: mask"));
        ISX_CHECK_ARG(images[i].type == images[0].type && (images[i].type == ISX_32FC3 || images[i].type == ISX_8UC3), ISX_ERR_TYPE,
                      "dp_seam_find: all images must have CV_32FC3 or CV_8UC3 type (S:745-746)");
        ISX_CHECK_ARG(masks[i].type == ISX_8UC1, ISX_ERR_TYPE, "dp_seam_find: masks must be CV_8U");
        ISX_CHECK_ARG(masks[i].rows == images[i].rows && masks[i].cols == images[i].cols, ISX_ERR_SIZE, "dp_seam_find: image %d and its mask differ in size (S:133-134)", i);
    }
    ISX_HIP(hipSetDevice(device));
    // the logic below edits the masks on the host: device masks are brought down and written back
    std::vector<std::vector<unsigned char>> hostm(num_images);
    std::vector<unsigned char*> mp(num_images);
    std::vector<size_t> ms(num_images);
    bool any_dev = false;
    for (int i = 0; i < num_images; ++i) {
        if (masks[i].device >= 0) {
            hostm[i].resize((size_t)masks[i].rows * masks[i].cols);
            // on the caller's stream: the masks may still be in production there (a non-blocking stream is not ordered
            // against the legacy null stream a plain hipMemcpy2D would use)
            ISX_HIP(hipMemcpy2DAsync(hostm[i].data(), masks[i].cols, masks[i].data, masks[i].step, masks[i].cols, masks[i].rows, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
            mp[i] = hostm[i].data(); ms[i] = (size_t)masks[i].cols;
            any_dev = true;
        } else { mp[i] = (unsigned char*)masks[i].data; ms[i] = masks[i].step; }
    }
    if (any_dev) ISX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    // host images are staged in HBM once per call, not once per conflict (estimateSeam runs for every conflicting pair of
    // components and reads both images each time: ~100 MB per CV_32FC3 4K image and upload otherwise)
    std::vector<isx_mat> dimg(images, images + num_images);
    FinderStages& fs = finder_stages();
    if (fs.device != device) { fs.img.clear(); fs.device = device; }
    if ((int)fs.img.size() < num_images) fs.img.resize(num_images);
    for (int i = 0; i < num_images; ++i)
        if (images[i].device < 0) {
            if (!fs.img[i]) fs.img[i].reset(new MatStage());
            ISX_TRY(fs.img[i]->use_in(&images[i], (hipStream_t)hip_stream, "dp_seam_find: image"));
            dimg[i] = fs.img[i]->d; dimg[i].device = device;
        }
    images = dimg.data();
    // the union-sized work images (labels, the two masks: ~6 B per union pixel) keep their storage between calls of a thread:
    // a fresh 70 MB of vectors per 4K pair spent a third of the call in page faults (isx_dp_seam_release returns it)
    Finder& f = finder();
    f.device = device;
    f.stream = (hipStream_t)hip_stream;
    std::vector<std::pair<int, int>> pairs;   // S:98-113
    for (int i = 0; i + 1 < num_images; ++i)
        for (int j = i + 1; j < num_images; ++j) pairs.push_back({i, j});
    std::reverse(pairs.begin(), pairs.end());
    for (const auto& pr : pairs) {
        const int i0 = pr.first, i1 = pr.second;
        ISX_TRY(f.process(&images[i0], &images[i1], {corners_xy[2 * i0], corners_xy[2 * i0 + 1]}, {corners_xy[2 * i1], corners_xy[2 * i1 + 1]}, mp[i0], ms[i0], mp[i1], ms[i1]));
    }
    for (int i = 0; i < num_images; ++i)
        if (masks[i].device >= 0)
            ISX_HIP(hipMemcpy2DAsync(masks[i].data, masks[i].step, hostm[i].data(), masks[i].cols, masks[i].cols, masks[i].rows, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
    if (any_dev) ISX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));   // hostm dies with this frame
    return ISX_OK;
} ISX_EXIT("isx_dp_seam_find")

}  // extern "C"
