"""An emulation of Qhull's 2-D incremental hull (qh_initbuild, qh_partitionall, qh_buildhull with qh_findhorizon,
qh_makenew_simplicial, qh_partitionvisible as libqhull 2020.2 runs them for scipy's ConvexHull with default
options) far enough to tell the order of its facet list, and with it the vertex that
``scipy.spatial.ConvexHull(xy).vertices`` lists first (build container only; experiments).

python tools/qhull2d.py [slices] compares with scipy on random slices."""
import math
import sys

import numpy as np


class Facet:
    __slots__ = ("v", "top", "n", "off", "out", "far", "nb", "visible", "new", "visit", "seen", "id")

    def __init__(self, fid, v, top):
        self.id, self.v, self.top = fid, v, top
        self.out, self.far, self.nb = None, 0.0, [None, None]
        self.visible = self.new = self.seen = False
        self.visit = 0


def emulate(xy, minoutside=None, variant=0, trace=False):
    """Facets (pairs of point indices, qh facet_list order at the end) and the first vertex scipy would list."""
    pts = [(float(a), float(b)) for a, b in xy]
    n = len(pts)
    # qh_maxmin: per coordinate the first point with the least / greatest value; MINoutside from the roundoff
    # model (DISTround ~ a few 1e-15 at these scales; only its order of magnitude matters here)
    maxabs = max(max(abs(p[0]), abs(p[1])) for p in pts)
    eps = 2.220446049250313e-16
    sumabs = sum(max(abs(p[k]) for p in pts) for k in range(2))
    distround = eps * (2 * min(math.sqrt(2.0) * maxabs, sumabs) * 1.01 + maxabs)       # qh_distround
    minvisible = 2 * distround          # premerge centrum of C-0
    if minoutside is None:
        minoutside = 2 * minvisible
    searchdist = 2 * distround + minvisible                                             # qh_SEARCHdist, max_outside 0
    distoutside = 2 * minoutside                                                         # qh_DISToutside when merging
    ext = []
    for k in range(2):
        lo = hi = 0
        for i in range(1, n):
            if pts[i][k] > pts[hi][k]:
                hi = i
            elif pts[i][k] < pts[lo][k]:
                lo = i
        ext += [lo, hi]
    # qh_maxsimplex: least and greatest x of the extreme points, then the point of largest |det|
    minx = min(ext, key=lambda i: (pts[i][0], ext.index(i)))
    maxx = max(ext, key=lambda i: (pts[i][0], -ext.index(i)))
    simplex = [minx]
    if maxx != minx:
        simplex.append(maxx)

    def det(i):
        a, b = pts[simplex[0]], pts[simplex[1]]
        p = pts[i]
        return abs((a[0] - p[0]) * (b[1] - p[1]) - (b[0] - p[0]) * (a[1] - p[1]))

    best, bestdet = None, -1.0
    for i in ext:
        if i not in simplex:
            d = det(i)
            if d > bestdet:
                best, bestdet = i, d
    if best is None or bestdet < 1e-3 * abs(pts[simplex[1]][0] - pts[simplex[0]][0]) ** 2:
        # (2020.2 searches all points when the extreme points give a poor simplex)
        for i in range(n):
            if i not in simplex and i != best:
                d = det(i)
                if d > bestdet:
                    best, bestdet = i, d
    simplex.append(best)
    # vertices: ids in the order of the simplex, sets kept by decreasing id
    vid = {}
    order = []          # vertex ids by creation: vid[point] = id

    def newvertex(p):
        vid[p] = len(order)
        order.append(p)

    for p in simplex:
        newvertex(p)
    vs = sorted(simplex, key=lambda p: -vid[p])         # [v2, v1, v0]
    facets = []
    nextid = [0]

    def setplane(f):
        p0, p1 = pts[f.v[0]], pts[f.v[1]]
        nx, ny = p1[1] - p0[1], p0[0] - p1[0]
        norm = math.sqrt(nx * nx + ny * ny)
        if not f.top:
            norm = -norm
        f.n = (nx / norm, ny / norm)
        f.off = -(p0[0] * f.n[0] + p0[1] * f.n[1])

    def dist(p, f):
        q = pts[p]
        return f.off + q[0] * f.n[0] + q[1] * f.n[1]

    top = True
    for i in range(3):
        f = Facet(nextid[0], [v for k, v in enumerate(vs) if k != i], top)
        nextid[0] += 1
        facets.append(f)
        top = not top
    for f in facets:
        f.nb = [g for g in facets if g is not f][:2]
    cx = sum(pts[p][0] for p in simplex) / 3.0
    cy = sum(pts[p][1] for p in simplex) / 3.0
    setplane(facets[0])
    if facets[0].off + cx * facets[0].n[0] + cy * facets[0].n[1] > 0:
        for f in facets:
            f.top = not f.top
    for f in facets:
        setplane(f)
    # qh_partitionall: every point to the first facet it lies above; the farthest last
    rest = [i for i in range(n) if i not in simplex]
    for f in facets:
        keep, outs, bestp, bestd = [], [], None, 0.0
        for p in rest:
            d = dist(p, f)
            if d < distoutside:
                keep.append(p)
            elif bestp is None:
                bestp, bestd = p, d
            elif d > bestd:
                outs.append(bestp)
                bestp, bestd = p, d
            else:
                outs.append(p)
        if bestp is not None:
            outs.append(bestp)
            f.out, f.far = outs, bestd
        rest = keep
    # the facet list with its tail sentinel and qh facet_next, kept as qh_removefacet / qh_appendfacet keep them
    TAIL = object()
    flist = facets + [TAIL]
    state = {"next": flist[0]}

    def remove(f):
        k0 = flist.index(f)
        if state["next"] is f:
            state["next"] = flist[k0 + 1]
        del flist[k0]

    def append(f):
        if state["next"] is TAIL:
            state["next"] = f
        flist.insert(len(flist) - 1, f)

    flags = {"findbestnew": False, "notsharp": False}
    newlist = []                    # qh newfacet_list as a Python list, in list order

    def findbesthorizon(p, start, bestdist):
        best, seen, facet, copl, size, nextfacet = start, {id(start)}, start, [], 0, None
        minsearch = bestdist - searchdist
        while True:
            for nb in facet.nb:
                if id(nb) in seen:
                    continue
                seen.add(id(nb))
                d = dist(p, nb)
                if d > bestdist:
                    minsearch = d - searchdist
                    if d > bestdist + searchdist:
                        size = 0
                    best, bestdist = nb, d
                elif d < minsearch:
                    continue
                if nextfacet is not None:
                    if size == 0:
                        copl, size = [nextfacet], 1
                    else:
                        copl.append(nextfacet)
                        size += 1
                nextfacet = nb
            facet = nextfacet
            if facet is not None:
                nextfacet = None
            elif size == 0:
                break
            elif size == 1:
                size, facet, copl = 0, copl[0], []
            else:
                size -= 1
                facet = copl.pop()
        return best, bestdist

    def findbestnew(p, start):
        best, bestdist = None, -1e300
        k0 = flist.index(start)
        first_new = flist.index(newlist[0]) if newlist else 0
        for f in flist[k0:-1] + flist[first_new:k0]:
            if f.visible:
                continue
            d = dist(p, f)
            if d > bestdist:
                best = f
                if d >= distoutside:
                    return best, d, True
                bestdist = d
        best, bestdist = findbesthorizon(p, best if best is not None else start, bestdist)
        return best, bestdist, bestdist >= minoutside

    def sharpnewfacets():
        q = [newlist[0].n[0] > 0, newlist[0].n[1] > 0]
        for f in newlist[1:]:
            if q[1] != (f.n[1] > 0) or q[0] != (f.n[0] > 0):
                return True
        return False

    def findbest(p, start):
        d = dist(p, start)
        if d >= minoutside:
            return start, d, True
        best, bestdist, seen, facet = start, d, {id(start)}, start
        while facet is not None:
            nxt = None
            for nb in facet.nb:
                if not nb.new or id(nb) in seen:
                    continue
                seen.add(id(nb))
                d = dist(p, nb)
                if d > bestdist:
                    if d >= minoutside:
                        return nb, d, True
                    best, bestdist, nxt = nb, d, nb
                    break
            facet = nxt
        testhorizon = True
        if not flags["notsharp"] and bestdist < -distround:
            if sharpnewfacets():
                best, d2, out2 = findbestnew(p, best)
                testhorizon = False
                flags["findbestnew"] = True
                if variant & 2:
                    bestdist = d2
            else:
                flags["notsharp"] = True
        if testhorizon:
            best, bestdist = findbesthorizon(p, best, bestdist)
        return best, bestdist, bestdist >= minoutside

    def partitionpoint(p, start):
        f, d, isout = findbestnew(p, start) if flags["findbestnew"] else findbest(p, start)
        if not isout:
            return
        if not f.out:
            f.out = [p]
            if not f.new:
                remove(f)
                append(f)
            f.far = d
        elif f.far < d:
            f.out.append(p)
            f.far = d
        else:
            f.out.insert(len(f.out) - 1, p)

    # the rest of qh_partitionall: the points no facet took, through qh_findbestnew
    for f in flist[:-1]:
        f.new = True
    newlist = flist[:-1]
    flags["findbestnew"] = True
    for p in rest:
        partitionpoint(p, flist[0])
    flags["findbestnew"] = False
    for f in flist[:-1]:
        f.new = False
    # qh_furthestnext: the initial facet with the farthest outside point goes to the front of the list
    bestf, bestd = None, -1e300
    for f in flist[:-1]:
        if f.out and f.far > bestd:
            bestf, bestd = f, f.far
    if bestf is not None:
        flist.remove(bestf)
        flist.insert(0, bestf)
    state["next"] = flist[0]
    visit = 0
    while True:
        # qh_nextfurthest
        while state["next"] is not TAIL and not state["next"].out:
            state["next"] = flist[flist.index(state["next"]) + 1]
        if state["next"] is TAIL:
            break
        facet = state["next"]
        apex = facet.out.pop()
        if trace: print('step facet',facet.v,'apex',apex,'list',[tuple(f.v) for f in flist[:-1]], 'outs',[len(f.out or []) for f in flist[:-1]])
        # qh_findhorizon: visible facets move to the end of the list, in the order they are found
        remove(facet)
        append(facet)
        facet.visible = True
        vis = [facet]
        visit += 1
        k = 0
        while k < len(vis):
            v = vis[k]
            v.visit = visit
            for g in v.nb:
                if g.visit == visit:
                    continue
                g.visit = visit
                if dist(apex, g) > minvisible:
                    remove(g)
                    append(g)
                    g.visible = True
                    vis.append(g)
            k += 1
        # qh_makenewfacets / qh_makenew_simplicial
        newvertex(apex)
        new = []
        for v in vis:
            for g in v.nb:
                g.seen = False
            for g in v.nb:
                if not g.seen and not g.visible:
                    hskip = g.nb.index(v)                       # where the horizon facet points to the visible one
                    ridge = [w for k2, w in enumerate(g.v) if k2 != hskip]
                    topo = bool(hskip & 1) if g.top else not bool(hskip & 1)
                    f = Facet(nextid[0], [apex] + ridge, topo)
                    nextid[0] += 1
                    f.new = True
                    f.nb = [g, None]
                    g.nb[hskip] = f
                    setplane(f)
                    append(f)
                    new.append(f)
        # qh_matchnewfacets: the two new facets share the apex
        assert len(new) == 2, len(new)
        new[0].nb[1], new[1].nb[1] = new[1], new[0]
        # qh_partitionvisible: the outside points of the visible facets, through qh_partitionpoint
        newlist = new
        flags["findbestnew"] = flags["notsharp"] = False
        for v in vis:
            for p in (v.out or []):
                partitionpoint(p, new[0])
        flags["findbestnew"] = flags["notsharp"] = False
        # qh_deletevisible
        for v in vis:
            remove(v)
        for f in new:
            f.new = False
    facets = flist[:-1]
    first = facets[0]
    a = first.v[0] if (first.top ^ False) else first.v[1]
    return [tuple(f.v) for f in facets], a, [f.top for f in facets]


if __name__ == "__main__":
    from scipy.spatial import ConvexHull
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    g = np.random.default_rng(1)
    for variant in (0, 2):
        okf = okv = oka = 0
        for r in range(rounds):
            n = int(g.integers(20, 400))
            th = g.uniform(0, 2 * np.pi, n)
            rad = 10 + g.normal(0, 0.05, n)
            bad = g.random(n) < 0.05
            rad[bad] = g.uniform(0, 14, int(bad.sum()))
            xy = np.unique(np.stack([3 + rad * np.cos(th), -2 + rad * np.sin(th)], 1), axis=0)
            h = ConvexHull(xy)
            try:
                fl, a, tops = emulate(xy, variant=variant)
            except AssertionError:
                continue
            want = [tuple(s) for s in h.simplices.tolist()]
            okf += [set(f) for f in fl] == [set(s) for s in want]
            okv += fl == want
            oka += a == int(h.vertices[0])
        print(f"variant {variant}: facet list (as sets) {okf}/{rounds}, with vertex order {okv}/{rounds}, first vertex {oka}/{rounds}")
