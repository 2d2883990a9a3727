"""CPU restatement of the reference's in-tree DP seam finder as a whole (S = 动态规划法寻找最佳缝合线.cpp):
find S:87-124, process S:127-193, findComponents S:196-308, findEdges S:311-392, resolveConflicts S:395-546,
hasOnlyOneNeighbor S:574-582, closeToContour S:585-604, getSeamTips S:607-706, updateLabelsUsingSeam S:960-1093;
estimateSeam / computeCosts (S:733-957) are the C oracle's orc_seam_estimate.  costFunc_ = COLOR (S:71-72).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  NumPy / SciPy for the image-sized steps, plain Python for the small
graph logic.  cv::floodFill (4-connectivity, zero tolerance) = connected components of equal-valued pixels, numbered in
the raster order in which S:228-240 meets their first pixel; cv::partition = union-find classes numbered in the order
of their first member."""
import numpy as np
from scipy import ndimage

from . import capi as O

FIRST, SECOND, INTERS = 1, 2, 4
_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)


def _contour_of(lab_eq):
    """pixels of a region with a 4-neighbour outside it (or on the image border): the test of S:176-186 / S:249-253"""
    p = np.pad(lab_eq, 1, constant_values=False)
    inner = p[1:-1, :-2] & p[1:-1, 2:] & p[:-2, 1:-1] & p[2:, 1:-1]
    return lab_eq & ~inner


def _partition(points, min_dist):
    """cv::partition(points, labels, ClosePoints(min_dist)) S:44-57, 629"""
    n = len(points)
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    pts = np.asarray(points, np.int64)
    for i in range(n):
        d2 = ((pts - pts[i]) ** 2).sum(1)
        for j in np.nonzero(d2 < min_dist * min_dist)[0]:
            a, b = find(i), find(int(j))
            if a != b:
                parent[b] = a
    labels, seen = [], {}
    for i in range(n):
        r = find(i)
        if r not in seen:
            seen[r] = len(seen)
        labels.append(seen[r])
    return labels


class DpSeamFinder:
    def find(self, src, corners, masks):
        """find(src, corners, masks) S:87-124: masks are modified in place (uint8 arrays)."""
        n = len(src)
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
        pairs.reverse()                                                   # S:113
        for i0, i1 in pairs:
            self.process(src[i0], src[i1], corners[i0], corners[i1], masks[i0], masks[i1])
        return masks

    # ---------------------------------------------------------------------------------------------------------
    def process(self, image1, image2, tl1, tl2, mask1, mask2):
        assert image1.shape[:2] == mask1.shape and image2.shape[:2] == mask2.shape
        itl = (max(tl1[0], tl2[0]), max(tl1[1], tl2[1]))
        ibr = (min(tl1[0] + image1.shape[1], tl2[0] + image2.shape[1]), min(tl1[1] + image1.shape[0], tl2[1] + image2.shape[0]))
        if itl[0] >= ibr[0] or itl[1] >= ibr[1]:
            return                                                        # there are no conflicts, S:142-143
        self.utl = (min(tl1[0], tl2[0]), min(tl1[1], tl2[1]))
        ubr = (max(tl1[0] + image1.shape[1], tl2[0] + image2.shape[1]), max(tl1[1] + image1.shape[0], tl2[1] + image2.shape[0]))
        self.uw, self.uh = ubr[0] - self.utl[0], ubr[1] - self.utl[1]
        self.mask1_ = np.zeros((self.uh, self.uw), np.uint8)
        self.mask2_ = np.zeros((self.uh, self.uw), np.uint8)
        o1 = (tl1[0] - self.utl[0], tl1[1] - self.utl[1])
        o2 = (tl2[0] - self.utl[0], tl2[1] - self.utl[1])
        self.mask1_[o1[1]:o1[1] + mask1.shape[0], o1[0]:o1[0] + mask1.shape[1]] = mask1
        self.mask2_[o2[1]:o2[1] + mask2.shape[0], o2[0]:o2[0] + mask2.shape[1]] = mask2
        self.contour1mask_ = _contour_of(self.mask1_ > 0)                 # S:168-186
        self.contour2mask_ = _contour_of(self.mask2_ > 0)
        self.find_components()
        self.find_edges()
        self.resolve_conflicts(image1, image2, tl1, tl2, mask1, mask2)

    # ---------------------------------------------------------------------------------------------------------
    def _region_info(self, l):
        ys, xs = np.nonzero(self.labels == l)
        if len(ys) == 0:
            return (np.iinfo(np.int32).max,) * 2, (np.iinfo(np.int32).min,) * 2, []
        tl, br = (int(xs.min()), int(ys.min())), (int(xs.max()) + 1, int(ys.max()) + 1)
        cy, cx = np.nonzero(_contour_of(self.labels == l))                # raster order, as S:249-253 pushes them
        return tl, br, list(zip(cx.tolist(), cy.tolist()))

    def find_components(self):
        m1, m2 = self.mask1_ > 0, self.mask2_ > 0
        cls = np.zeros((self.uh, self.uw), np.int32)                      # S:207-219
        cls[m2] = SECOND; cls[m1] = FIRST; cls[m1 & m2] = INTERS
        comps = []
        for state in (INTERS, FIRST, SECOND):
            lab, n = ndimage.label(cls == state, structure=_CROSS)
            if n:
                first = ndimage.minimum(np.arange(lab.size).reshape(lab.shape), lab, index=np.arange(1, n + 1))
                comps += [(int(f), state, lab, k + 1) for k, f in enumerate(np.atleast_1d(first))]
        comps.sort(key=lambda c: c[0])                                    # the raster order of the seeds, S:224-234
        self.labels = np.zeros((self.uh, self.uw), np.int32)
        self.states = []
        for i, (_, state, lab, k) in enumerate(comps):
            self.labels[lab == k] = i + 1
            self.states.append(state)
        self.ncomps = len(comps)
        self.tls, self.brs, self.contours = [], [], []
        for ci in range(self.ncomps):
            tl, br, cont = self._region_info(ci + 1)
            self.tls.append(tl); self.brs.append(br); self.contours.append(cont)

    def find_edges(self):
        lab = self.labels                                                 # S:311-392: components that touch (4-neighbourhood)
        edges = set()
        for a, b in ((lab[:, :-1], lab[:, 1:]), (lab[:-1, :], lab[1:, :])):
            m = (a != b) & (a > 0) & (b > 0)
            for p, q in set(zip(a[m].tolist(), b[m].tolist())):
                edges.add((p - 1, q - 1)); edges.add((q - 1, p - 1))
        self.edges = edges

    def has_only_one_neighbor(self, comp):
        return sum(1 for e in self.edges if e[0] == comp) == 1            # S:574-582

    # ---------------------------------------------------------------------------------------------------------
    def resolve_conflicts(self, image1, image2, tl1, tl2, mask1, mask2):
        while True:
            conflict = None
            for c1, c2 in sorted(self.edges):                             # std::set order, S:416-426
                if (self.states[c1] & INTERS) and (self.states[c1] & ~INTERS) != self.states[c2]:
                    conflict = (c1, c2)
                    break
            if conflict is None:
                break
            c1, c2 = conflict
            l1, l2 = c1 + 1, c2 + 1
            if self.has_only_one_neighbor(c1):                            # S:432-441
                tl, br = self.tls[c1], self.brs[c1]
                sub = self.labels[tl[1]:br[1], tl[0]:br[0]]
                sub[sub == l1] = l2
                self.states[c1] = SECOND if self.states[c2] == FIRST else FIRST
            else:                                                         # S:442-455
                tips = self.get_seam_tips(c1, c2)
                if tips is not None:
                    seam, horiz = self.estimate_seam(image1, image2, tl1, tl2, c1, tips[0], tips[1])
                    if len(seam):
                        self.update_labels_using_seam(c1, c2, seam, horiz)
                self.states[c1] = (INTERS | SECOND) if self.states[c2] == FIRST else (INTERS | FIRST)
            for c in (c1, c2):                                            # S:457-487: bounding boxes / contours of both, within the old boxes
                x0, x1, y0, y1 = self.tls[c][0], self.brs[c][0], self.tls[c][1], self.brs[c][1]
                win = np.zeros_like(self.labels, bool)
                win[y0:y1, x0:x1] = True
                eq = (self.labels == c + 1) & win
                ys, xs = np.nonzero(eq)
                if len(ys):
                    self.tls[c] = (int(xs.min()), int(ys.min())); self.brs[c] = (int(xs.max()) + 1, int(ys.max()) + 1)
                else:
                    self.tls[c] = (np.iinfo(np.int32).max,) * 2; self.brs[c] = (np.iinfo(np.int32).min,) * 2
                full = self.labels == c + 1
                cy, cx = np.nonzero(_contour_of(full) & win)
                self.contours[c] = list(zip(cx.tolist(), cy.tolist()))
            self.edges.discard((c1, c2)); self.edges.discard((c2, c1))    # S:489-490
        dx1, dy1 = self.utl[0] - tl1[0], self.utl[1] - tl1[1]             # S:495-523
        dx2, dy2 = self.utl[0] - tl2[0], self.utl[1] - tl2[1]
        st = np.array([0] + self.states, np.int32)[self.labels]
        lab2 = self.labels[-dy2:-dy2 + mask2.shape[0], -dx2:-dx2 + mask2.shape[1]]
        kill2 = (lab2 > 0) & ((st[-dy2:-dy2 + mask2.shape[0], -dx2:-dx2 + mask2.shape[1]] & FIRST) != 0) & \
                (self.mask1_[-dy2:-dy2 + mask2.shape[0], -dx2:-dx2 + mask2.shape[1]] > 0)
        m1_before = self.mask1_.copy()
        mask2[kill2] = 0
        m2u = np.zeros_like(self.mask2_)
        o2 = (-dx2, -dy2)
        m2u[o2[1]:o2[1] + mask2.shape[0], o2[0]:o2[0] + mask2.shape[1]] = mask2      # mask2 as already modified (S:512-523 reads it)
        lab1 = self.labels[-dy1:-dy1 + mask1.shape[0], -dx1:-dx1 + mask1.shape[1]]
        kill1 = (lab1 > 0) & ((st[-dy1:-dy1 + mask1.shape[0], -dx1:-dx1 + mask1.shape[1]] & SECOND) != 0) & \
                (m2u[-dy1:-dy1 + mask1.shape[0], -dx1:-dx1 + mask1.shape[1]] > 0)
        mask1[kill1] = 0
        del m1_before

    # ---------------------------------------------------------------------------------------------------------
    def close_to_contour(self, y, x, cm):
        return bool(cm[max(y - 2, 0):y + 3, max(x - 2, 0):x + 3].any())   # S:585-604, rad = 2

    def get_seam_tips(self, comp1, comp2):
        l2 = comp2 + 1
        lab = self.labels
        special = []
        for x, y in self.contours[comp1]:                                 # S:614-630
            if self.close_to_contour(y, x, self.contour1mask_) and self.close_to_contour(y, x, self.contour2mask_) and (
                    (x > 0 and lab[y, x - 1] == l2) or (y > 0 and lab[y - 1, x] == l2) or
                    (x < self.uw - 1 and lab[y, x + 1] == l2) or (y < self.uh - 1 and lab[y + 1, x] == l2)):
                special.append((x, y))
        if len(special) < 2:
            return None
        labels = _partition(special, 10)
        nl = max(labels) + 1
        if nl < 2:
            return None
        pts = [[] for _ in range(nl)]
        for p, l in zip(special, labels):
            pts[l].append(p)
        sums = [(sum(p[0] for p in g), sum(p[1] for p in g)) for g in pts]
        rnd = lambda v: float(np.rint(v))                                 # cvRound(double)
        idx, best = (-1, -1), -np.inf
        for i in range(nl - 1):                                           # S:649-667: the two clusters farthest apart
            for j in range(i + 1, nl):
                cx1, cy1 = rnd(sums[i][0] / len(pts[i])), rnd(sums[i][1] / len(pts[i]))
                cx2, cy2 = rnd(sums[j][0] / len(pts[j])), rnd(sums[j][1] / len(pts[j]))
                d = (cx1 - cx2) ** 2 + (cy1 - cy2) ** 2
                if d > best:
                    best, idx = d, (i, j)
        tips = []
        for k in idx:                                                     # S:669-693: the cluster member closest to its centroid
            cx, cy = rnd(sums[k][0] / len(pts[k])), rnd(sums[k][1] / len(pts[k]))
            dist = [(p[0] - cx) ** 2 + (p[1] - cy) ** 2 for p in pts[k]]
            tips.append(pts[k][int(np.argmin(dist))])
        return tips[0], tips[1]

    def estimate_seam(self, image1, image2, tl1, tl2, comp, p1, p2):
        roi = (self.tls[comp][0], self.tls[comp][1], self.brs[comp][0] - self.tls[comp][0], self.brs[comp][1] - self.tls[comp][1])
        seam, horiz = O.seam_estimate(image1, image2, tl1, tl2, self.utl, self.labels, comp + 1, roi, p1, p2)
        return [tuple(int(v) for v in p) for p in seam], horiz

    def update_labels_using_seam(self, comp1, comp2, seam, horiz):
        tl, br = self.tls[comp1], self.brs[comp1]
        h, w = br[1] - tl[1], br[0] - tl[0]
        mask = np.zeros((h, w), np.int32)
        for x, y in self.contours[comp1]:
            mask[y - tl[1], x - tl[0]] = 255
        for x, y in seam:
            mask[y - tl[1], x - tl[0]] = 255
        l1, l2 = comp1 + 1, comp2 + 1
        sub = self.labels[tl[1]:br[1], tl[0]:br[0]]
        # S:976-981: flood fills of the zero pixels of `mask`, seeded (raster order) at zero pixels that carry label l1
        zlab, nz = ndimage.label(mask == 0, structure=_CROSS)
        ncomps = 0
        order = {}
        seeds = np.nonzero((mask == 0) & (sub == l1))
        for y, x in zip(*seeds):                                          # raster order
            k = int(zlab[y, x])
            if k not in order:
                ncomps += 1
                order[k] = ncomps
        for k, v in order.items():
            mask[zlab == k] = v
        dxs = (-1, +1, 0, 0, -1, +1, -1, +1)
        dys = (0, 0, -1, +1, -1, -1, +1, +1)
        for cx, cy in self.contours[comp1]:                               # S:983-1006 (sequential: later pixels see earlier updates)
            x, y = cx - tl[0], cy - tl[1]
            ok = False
            for j in range(8):
                c, r = x + dxs[j], y + dys[j]
                if 0 <= c < w and 0 <= r < h and mask[r, c] and mask[r, c] != 255:
                    ok = True
                    mask[y, x] = mask[r, c]
            if not ok:
                mask[y, x] = 0
        for sx, sy in seam:                                               # S:1008-1033
            x, y = sx - tl[0], sy - tl[1]
            if horiz:
                mask[y, x] = mask[y + 1, x] if (y < h - 1 and mask[y + 1, x] and mask[y + 1, x] != 255) else 0
            else:
                mask[y, x] = mask[y, x + 1] if (x < w - 1 and mask[y, x + 1] and mask[y, x + 1] != 255) else 0
        connect2 = {i: 0 for i in range(1, ncomps + 1)}                   # S:1035-1066
        connect_other = {i: 0 for i in range(1, ncomps + 1)}
        lab = self.labels
        for x, y in self.contours[comp1]:
            nb = []
            if x > 0: nb.append(lab[y, x - 1])
            if y > 0: nb.append(lab[y - 1, x])
            if x < self.uw - 1: nb.append(lab[y, x + 1])
            if y < self.uh - 1: nb.append(lab[y + 1, x])
            m = int(mask[y - tl[1], x - tl[0]])
            if any(v == l2 for v in nb):
                connect2[m] = connect2.get(m, 0) + 1
            if any(v != l1 and v != l2 for v in nb):
                connect_other[m] = connect_other.get(m, 0) + 1
        length = float(len(self.contours[comp1]))
        is_adj = {}
        for k, v in connect2.items():                                     # S:1068-1084
            is_adj[k] = 1 if (v / length > 0.05 and k in connect_other and connect_other[k] / length < 0.1) else 0
        adj = np.zeros(max(max(is_adj) + 1, int(mask.max()) + 1), bool)
        for k, v in is_adj.items():
            if v:
                adj[k] = True
        sel = (mask > 0) & adj[np.clip(mask, 0, len(adj) - 1)] & (mask < len(adj))
        sub[sel] = l2                                                     # S:1086-1092
