// The order in which scipy.spatial.ConvexHull lists the vertices of a 2-D hull (circle_fitting.py:189-190 takes
// `hull.vertices` as the RANSAC candidates, so the vertex the walk starts at decides which triples the sampled
// indices name).  scipy walks Qhull's facet list from its head (each facet an edge; the first vertex is the
// head facet's, in the facet's orientation), and the head of the list is whatever Qhull's incremental
// construction left there.  This file replays that construction for points in the plane - the initial simplex
// from the coordinate extremes, the first partition of the points into outside sets, the farthest-point
// insertions in facet-list order with the list manipulations that go with them - on host cores, in the same
// double arithmetic (no contraction), far enough to know the final facet list.  Libqhull itself is not linked
// or copied: scipy vendors it and the reference only reaches it through scipy.  What is replayed is the
// behaviour of libqhull 2020.2 under scipy's options for ConvexHull (triangulated output, default merging
// with centrum 0), for inputs in general position; every decision that falls inside Qhull's own roundoff
// bands (coplanar horizon facets, merges of nearly collinear edges, points within the outside threshold, a
// flat initial simplex) clears the slice's `sure` flag, because merging is not replayed.
#include "pcfit_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

namespace pcfit {
namespace {

struct Facet {
    int v[2];                       // point indices, the vertex created later first
    int nb[2];                      // nb[k] lies opposite v[k]
    int prev, next;                 // facet list
    double n0, n1, off, far;        // outward unit normal, offset, distance of the farthest outside point
    std::vector<int> out;           // outside set; the farthest point last
    unsigned visit;
    bool top, visible, isnew;
};

class Replay {
  public:
    Replay(const double *x, const double *y, int64_t n) : x_(x), y_(y), n_(n) {}
    // false when the construction cannot be started (fewer than 3 points, no simplex)
    bool run();
    bool sure() const { return sure_; }
    int64_t count() const { return n_facets_; }
    // the vertices in scipy's order: from the head facet, each facet's first vertex in its orientation
    void walk(int64_t *out, int64_t cap, int64_t *first);

  private:
    const double *x_, *y_;
    int64_t n_;
    std::vector<Facet> f_;          // f_[0] is the list's tail sentinel
    int head_ = 0, next_ = 0;       // qh facet_list, qh facet_next
    unsigned visit_ = 0;
    double distround_ = 0, minvisible_ = 0, minoutside_ = 0, searchdist_ = 0, distoutside_ = 0, band_ = 0;
    bool sure_ = true, findbestnew_ = false, notsharp_ = false;
    int new_[2] = {0, 0}, n_new_ = 0, first_new_ = 0;
    int64_t n_facets_ = 0;
    std::vector<int> copl_, vis_;

    double dist(int p, int f) const { return f_[f].off + x_[p] * f_[f].n0 + y_[p] * f_[f].n1; }
    void near(double d) { if (std::fabs(d) < band_) sure_ = false; }
    int make(int v0, int v1, bool top)
    {
        Facet f;
        f.v[0] = v0; f.v[1] = v1; f.nb[0] = f.nb[1] = 0; f.prev = f.next = 0;
        f.n0 = f.n1 = f.off = f.far = 0.0;
        f.visit = 0; f.top = top; f.visible = f.isnew = false;
        f_.push_back(std::move(f));
        return (int)f_.size() - 1;
    }
    void setplane(int k)
    {
        Facet &f = f_[k];
        const double nx = y_[f.v[1]] - y_[f.v[0]], ny = x_[f.v[0]] - x_[f.v[1]];
        double norm = std::sqrt(nx * nx + ny * ny);
        if (!f.top) norm = -norm;
        f.n0 = nx / norm;
        f.n1 = ny / norm;
        f.off = -(x_[f.v[0]] * f.n0 + y_[f.v[0]] * f.n1);
    }
    // the list keeps qh facet_next the way qh_removefacet and qh_appendfacet do
    void remove(int k)
    {
        Facet &f = f_[k];
        if (next_ == k) next_ = f.next;
        if (first_new_ == k) first_new_ = f.next;
        if (head_ == k) { head_ = f.next; f_[f.next].prev = -1; }
        else { f_[f.prev].next = f.next; f_[f.next].prev = f.prev; }
    }
    void append(int k)
    {
        if (next_ == 0) next_ = k;
        if (first_new_ == 0) first_new_ = k;
        Facet &f = f_[k];
        f.prev = f_[0].prev;
        f.next = 0;
        if (head_ == 0) { head_ = k; f.prev = -1; }
        else f_[f_[0].prev].next = k;
        f_[0].prev = k;
    }
    int horizon(int p, int start, double &bestdist);
    int bestnew(int p, int start, double &d, bool &isout);
    bool sharp() const;
    int best(int p, int start, double &d, bool &isout);
    void partition(int p, int start);
};

// qh_findbesthorizon: from a facet, over neighbours whose distance is no more than the search distance below
// the best one found
int Replay::horizon(int p, int start, double &bestdist)
{
    int bestf = start, facet = start, nextfacet = 0;
    const unsigned id = ++visit_;
    int size = 0;
    double minsearch = bestdist - searchdist_;
    f_[start].visit = id;
    while (true) {
        for (int k = 0; k < 2; k++) {
            const int nb = f_[facet].nb[k];
            if (f_[nb].visit == id) continue;
            f_[nb].visit = id;
            const double d = dist(p, nb);
            if (d > bestdist) {
                minsearch = d - searchdist_;
                if (d > bestdist + searchdist_) size = 0;
                bestf = nb;
                bestdist = d;
            } else if (d < minsearch)
                continue;
            if (nextfacet) {
                if (size == 0) { copl_.assign(1, nextfacet); size = 1; }
                else { copl_.push_back(nextfacet); size++; }
            }
            nextfacet = nb;
        }
        facet = nextfacet;
        if (facet) nextfacet = 0;
        else if (size == 0) break;
        else if (size == 1) { size = 0; facet = copl_[0]; copl_.clear(); }
        else { size--; facet = copl_.back(); copl_.pop_back(); }
    }
    return bestf;
}

// qh_findbestnew: every facet from `start` to the end of the list, then the new facets before it
int Replay::bestnew(int p, int start, double &dout, bool &isout)
{
    int bestf = 0;
    double bestdist = -1e300;
    ++visit_;
    for (int pass = 0; pass < 2; pass++) {
        for (int f = pass ? first_new_ : start; f != 0; f = f_[f].next) {
            if (pass && f == start) break;
            if (f_[f].visible) continue;
            const double d = dist(p, f);
            near(d - minoutside_);
            near(d - distoutside_);
            if (d > bestdist) {
                bestf = f;
                if (d >= distoutside_) { dout = d; isout = true; return bestf; }
                bestdist = d;
            }
        }
    }
    bestf = horizon(p, bestf ? bestf : start, bestdist);
    near(bestdist - minoutside_);
    dout = bestdist;
    isout = bestdist >= minoutside_;
    return bestf;
}

// qh_sharpnewfacets: do the new facets' normals leave the first one's quadrant
bool Replay::sharp() const
{
    const Facet &a = f_[new_[0]];
    for (int k = 1; k < n_new_; k++) {
        const Facet &b = f_[new_[k]];
        if ((a.n1 > 0) != (b.n1 > 0) || (a.n0 > 0) != (b.n0 > 0)) return true;
    }
    return false;
}

// qh_findbest among the new facets: the first one the point lies outside of, walking to better neighbours
int Replay::best(int p, int start, double &dout, bool &isout)
{
    double d = dist(p, start);
    near(d - minoutside_);
    if (d >= minoutside_) { dout = d; isout = true; return start; }
    int bestf = start, facet = start;
    double bestdist = d;
    const unsigned id = ++visit_;
    f_[start].visit = id;
    while (facet) {
        int go = 0;
        for (int k = 0; k < 2; k++) {
            const int nb = f_[facet].nb[k];
            if (!f_[nb].isnew || f_[nb].visit == id) continue;
            f_[nb].visit = id;
            d = dist(p, nb);
            near(d - minoutside_);
            if (d > bestdist) {
                if (d >= minoutside_) { dout = d; isout = true; return nb; }
                bestf = nb;
                bestdist = d;
                go = nb;
                break;
            }
        }
        facet = go;
    }
    bool testhorizon = true;
    if (!notsharp_ && bestdist < -distround_) {
        if (sharp()) {
            double d2;
            bool o2;
            bestf = bestnew(p, bestf, d2, o2);
            testhorizon = false;
            findbestnew_ = true;
        } else
            notsharp_ = true;
    }
    near(bestdist + distround_);
    if (testhorizon) bestf = horizon(p, bestf, bestdist);
    near(bestdist - minoutside_);
    dout = bestdist;
    isout = bestdist >= minoutside_;
    return bestf;
}

// qh_partitionpoint: into the outside set of the facet found; a facet that is not new and had no outside
// points goes to the end of the list
void Replay::partition(int p, int start)
{
    double d;
    bool isout;
    const int k = findbestnew_ ? bestnew(p, start, d, isout) : best(p, start, d, isout);
    if (!isout) return;
    Facet &f = f_[k];
    if (f.out.empty()) {
        f.out.push_back(p);
        if (!f.isnew) { remove(k); append(k); }
        f.far = d;
    } else if (f.far < d) {
        f.out.push_back(p);
        f.far = d;
    } else
        f.out.insert(f.out.end() - 1, p);
}

bool Replay::run()
{
    const int n = (int)n_;
    if (n < 3) return false;
    // qh_maxmin: per coordinate the first point of least and of greatest value; the roundoff of a distance
    int ext[4];
    double maxabs = 0.0, sumabs = 0.0, maxwidth = 0.0;
    for (int k = 0; k < 2; k++) {
        const double *c = k ? y_ : x_;
        int lo = 0, hi = 0;
        for (int i = 1; i < n; i++) {
            if (c[i] > c[hi]) hi = i;
            else if (c[i] < c[lo]) lo = i;
        }
        ext[2 * k] = lo;
        ext[2 * k + 1] = hi;
        const double a = std::fmax(std::fabs(c[lo]), std::fabs(c[hi]));
        maxabs = std::fmax(maxabs, a);
        sumabs += a;
        maxwidth = std::fmax(maxwidth, c[hi] - c[lo]);
    }
    const double eps = 2.220446049250313e-16;
    distround_ = eps * (2 * std::fmin(std::sqrt(2.0) * maxabs, sumabs) * 1.01 + maxabs);
    minvisible_ = 2 * distround_;
    minoutside_ = 2 * minvisible_;
    searchdist_ = 2 * distround_ + minvisible_;
    distoutside_ = 2 * minoutside_;
    band_ = 64 * distround_;
    // qh_maxsimplex: least and greatest x among the extremes, then the extreme point spanning the largest triangle
    int minx = ext[0], maxx = ext[0];
    for (int k = 0; k < 4; k++) {
        if (x_[ext[k]] > x_[maxx]) maxx = ext[k];
        if (x_[ext[k]] < x_[minx]) minx = ext[k];
    }
    if (minx == maxx) return false;
    int third = -1;
    double bestdet = -1.0;
    for (int k = 0; k < 4; k++) {
        const int i = ext[k];
        if (i == minx || i == maxx) continue;
        const double det = std::fabs((x_[minx] - x_[i]) * (y_[maxx] - y_[i]) - (x_[maxx] - x_[i]) * (y_[minx] - y_[i]));
        if (det > bestdet) { bestdet = det; third = i; }
    }
    // a flat simplex (under a thousandth of what the extents allow, or a determinant Qhull calls near zero
    // for coordinates of this size) makes Qhull look through the other points, in their order, until one
    // spans a hundredth
    const double target = (x_[maxx] - x_[minx]) * maxwidth, mindet = 10 * 1e-3 * target;
    const double nearzero = 10 * (80 * sumabs * eps);
    if (third < 0 || bestdet < nearzero || bestdet / target < 1e-3) {
        for (int i = 0; i < n; i++) {
            if (i == ext[0] || i == ext[1] || i == ext[2] || i == ext[3]) continue;
            const double det =
                std::fabs((x_[minx] - x_[i]) * (y_[maxx] - y_[i]) - (x_[maxx] - x_[i]) * (y_[minx] - y_[i]));
            if (det > bestdet) {
                bestdet = det;
                third = i;
                if (!(det < nearzero) && bestdet > mindet) break;
            }
        }
        if (third < 0 || !(bestdet > 0.0)) return false;
    }
    // three facets, each the vertex list without one vertex; orientation alternates, then all flip if the
    // interior point lies above the first
    f_.clear();
    f_.reserve(64);
    make(-1, -1, false);                                   // tail sentinel
    f_[0].prev = -1;
    head_ = next_ = first_new_ = 0;
    const int vs[3] = {third, maxx, minx};                 // by decreasing vertex id
    int fs[3];
    bool top = true;
    for (int i = 0; i < 3; i++) {
        fs[i] = make(vs[i == 0 ? 1 : 0], vs[i == 2 ? 1 : 2], top);
        append(fs[i]);
        top = !top;
    }
    for (int i = 0; i < 3; i++) {
        int c = 0;
        for (int j = 0; j < 3; j++)
            if (j != i) f_[fs[i]].nb[c++] = fs[j];
    }
    const double cx = (x_[minx] + x_[maxx] + x_[third]) / 3.0, cy = (y_[minx] + y_[maxx] + y_[third]) / 3.0;
    setplane(fs[0]);
    if (f_[fs[0]].off + cx * f_[fs[0]].n0 + cy * f_[fs[0]].n1 > 0)
        for (int i = 0; i < 3; i++) f_[fs[i]].top = !f_[fs[i]].top;
    for (int i = 0; i < 3; i++) {
        setplane(fs[i]);
        if (!(f_[fs[i]].off + cx * f_[fs[i]].n0 + cy * f_[fs[i]].n1 < -band_)) sure_ = false;
    }
    // qh_partitionall: every point to the first facet it lies outside of, the farthest last; what no facet
    // took goes through qh_findbestnew once more
    std::vector<int> rest, keep;
    rest.reserve(n);
    for (int i = 0; i < n; i++)
        if (i != minx && i != maxx && i != third) rest.push_back(i);
    for (int i = 0; i < 3; i++) {
        Facet &f = f_[fs[i]];
        keep.clear();
        int bestp = -1;
        double bestd = 0.0;
        for (int p : rest) {
            const double d = dist(p, fs[i]);
            near(d - distoutside_);
            if (d < distoutside_) keep.push_back(p);
            else if (bestp < 0) { bestp = p; bestd = d; }
            else if (d > bestd) { f.out.push_back(bestp); bestp = p; bestd = d; }
            else f.out.push_back(p);
        }
        if (bestp >= 0) { f.out.push_back(bestp); f.far = bestd; }
        rest.swap(keep);
    }
    for (int i = 0; i < 3; i++) f_[fs[i]].isnew = true;
    first_new_ = head_;
    findbestnew_ = true;
    for (int p : rest) partition(p, head_);
    findbestnew_ = false;
    for (int i = 0; i < 3; i++) f_[fs[i]].isnew = false;
    // qh_furthestnext: the facet with the farthest outside point goes to the front
    {
        int bestf = 0;
        double bestd = -1e300;
        for (int f = head_; f != 0; f = f_[f].next)
            if (!f_[f].out.empty() && f_[f].far > bestd) { bestf = f; bestd = f_[f].far; }
        if (bestf && bestf != head_) {
            remove(bestf);
            Facet &f = f_[bestf];
            f.prev = -1;
            f.next = head_;
            f_[head_].prev = bestf;
            head_ = bestf;
        }
    }
    next_ = head_;
    // qh_buildhull
    while (true) {
        while (next_ != 0 && f_[next_].out.empty()) next_ = f_[next_].next;     // qh_nextfurthest
        if (next_ == 0) break;
        const int facet = next_;
        const int apex = f_[facet].out.back();
        f_[facet].out.pop_back();
        // qh_findhorizon: visible facets move to the end of the list as they are found
        first_new_ = 0;
        remove(facet);
        append(facet);
        first_new_ = 0;
        f_[facet].visible = true;
        vis_.assign(1, facet);
        const unsigned id = ++visit_;
        f_[facet].visit = id;
        for (size_t k = 0; k < vis_.size(); k++) {
            const int v = vis_[k];
            for (int j = 0; j < 2; j++) {
                const int g = f_[v].nb[j];
                if (f_[g].visit == id) continue;
                f_[g].visit = id;
                const double d = dist(apex, g);
                if (d > minvisible_) {
                    remove(g);
                    append(g);
                    first_new_ = 0;
                    f_[g].visible = true;
                    vis_.push_back(g);
                } else if (d >= -band_)
                    sure_ = false;                          // a coplanar horizon facet
                near(d - minvisible_);
            }
        }
        // qh_makenew_simplicial: one facet per horizon ridge, the apex first
        n_new_ = 0;
        for (size_t k = 0; k < vis_.size(); k++) {
            const int v = vis_[k];
            for (int j = 0; j < 2; j++) {
                const int g = f_[v].nb[j];
                if (f_[g].visible) continue;
                const int hskip = f_[g].nb[0] == v ? 0 : 1;
                const int ridge = f_[g].v[1 - hskip];
                const bool t = f_[g].top ? (hskip & 1) != 0 : (hskip & 1) == 0;
                if (n_new_ == 2) { sure_ = false; return false; }
                const int nf = make(apex, ridge, t);
                f_[nf].isnew = true;
                f_[nf].nb[0] = g;
                f_[g].nb[hskip] = nf;
                setplane(nf);
                append(nf);
                new_[n_new_++] = nf;
            }
        }
        if (n_new_ != 2) { sure_ = false; return false; }
        f_[new_[0]].nb[1] = new_[1];
        f_[new_[1]].nb[1] = new_[0];
        first_new_ = new_[0];
        // where Qhull's default merging would look twice: a new facet nearly in line with a neighbour
        for (int k = 0; k < 2; k++) {
            const int a = new_[k], g = f_[a].nb[0], b = new_[1 - k];
            const int w = f_[g].v[0] == f_[a].v[1] ? f_[g].v[1] : f_[g].v[0];
            if (dist(w, a) > -band_ || dist(apex, g) > -band_ || dist(f_[b].v[1], a) > -band_) sure_ = false;
        }
        // qh_partitionvisible
        findbestnew_ = notsharp_ = false;
        for (size_t k = 0; k < vis_.size(); k++) {
            std::vector<int> pts;
            pts.swap(f_[vis_[k]].out);
            for (int p : pts) partition(p, new_[0]);
        }
        findbestnew_ = notsharp_ = false;
        // qh_deletevisible
        for (size_t k = 0; k < vis_.size(); k++) remove(vis_[k]);
        f_[new_[0]].isnew = f_[new_[1]].isnew = false;
        first_new_ = 0;
    }
    n_facets_ = 0;
    for (int f = head_; f != 0; f = f_[f].next) n_facets_++;
    return true;
}

void Replay::walk(int64_t *out, int64_t cap, int64_t *first)
{
    // the facets from the head of the list on, each giving its first vertex in its orientation
    std::vector<int> w;
    int f = head_;
    for (int64_t i = 0; i < n_facets_; i++) {
        w.push_back(f);
        f = f_[f].top ? f_[f].nb[0] : f_[f].nb[1];
        if (f == head_) break;
    }
    const size_t m = w.size();
    // Facets in one line (points exactly on a hull edge) are one facet in Qhull's result, which merged them: the
    // vertex between two facets is left out when the far end of the second lies on the line of the first (Qhull's
    // centrum test for two edges: half that distance against twice the roundoff of a distance), and the merged
    // facet stands in the list where its newest part stands (it was appended when that part was made).
    std::vector<char> joined(m, 0);
    size_t n_joined = 0;
    // (Qhull's test as it makes it: each facet's centrum - the mean of its vertices, brought onto its own line -
    // against the other facet's line, within the centrum radius of twice the roundoff of a distance)
    auto centrum_dist = [&](int of, int to) {
        const Facet &a = f_[of], &b = f_[to];
        double cx = (x_[a.v[0]] + x_[a.v[1]]) / 2, cy = (y_[a.v[0]] + y_[a.v[1]]) / 2;
        const double own = a.off + cx * a.n0 + cy * a.n1;
        cx -= own * a.n0;
        cy -= own * a.n1;
        return b.off + cx * b.n0 + cy * b.n1;
    };
    for (size_t i = 0; i < m; i++) {
        const int fa = w[(i + m - 1) % m], fb = w[i];
        const double d1 = centrum_dist(fb, fa), d2 = centrum_dist(fa, fb), radius = 2 * distround_;
        joined[i] = (d1 >= -radius && d1 <= radius) || (d2 >= -radius && d2 <= radius);
        n_joined += joined[i];
    }
    if (m - n_joined < 3) std::fill(joined.begin(), joined.end(), 0);          // (no area left: nothing to merge)
    std::vector<int64_t> place(f_.size(), 0);
    int64_t k = 0;
    for (int g = head_; g != 0; g = f_[g].next) place[(size_t)g] = k++;
    size_t start = 0;
    while (joined[start]) start++;
    std::vector<std::pair<size_t, int64_t>> runs;                              // (first facet in the walk, place in the list)
    for (size_t t = 0, i = start; t < m; t++, i = (i + 1) % m) {
        if (!joined[i]) runs.push_back({i, place[(size_t)w[i]]});
        else runs.back().second = std::max(runs.back().second, place[(size_t)w[i]]);
    }
    size_t head = 0;
    for (size_t r = 1; r < runs.size(); r++)
        if (runs[r].second < runs[head].second) head = r;
    n_facets_ = (int64_t)runs.size();
    for (size_t t = 0; t < runs.size(); t++) {
        const Facet &a = f_[w[runs[(head + t) % runs.size()].first]];
        const int va = a.top ? a.v[0] : a.v[1];
        if (t == 0 && first) *first = va;
        if (out && (int64_t)t < cap) out[t] = va;
    }
}

// A device array of walks turned to start at the vertex with the given coordinates (no change where found[s] < 0
// or no vertex has them), and brought to Qhull's vertex set.  The walks come from an exact predicate and keep a
// vertex at the slightest turn; Qhull's result lacks some of them - it merges two edges whose far end lies within its
// roundoff of the other's line (coordinates rounded to a few decimals make such vertices), and it never makes a
// vertex of a point that no facet ever saw outside by its threshold (a point 2.7 roundoff units beyond the chord of
// its neighbours, 7e6 from the origin).  Which vertices Qhull ends with depends on the course of its construction,
// so the replay says it: verts holds vcount[s] coordinate pairs per slice (vstride apart), and the walk keeps the
// vertices found among them, in its own order from the start vertex on.  n_hull[s] becomes their number.  A slice
// whose list names a point that is no vertex of the walk keeps its walk whole (and its turn).
__global__ __launch_bounds__(256) void hull_rotate_kernel(const int64_t *__restrict__ offsets, int64_t *__restrict__ n_hull,
                                                          const double *__restrict__ ux, const double *__restrict__ uy,
                                                          const int64_t *__restrict__ found,
                                                          const double *__restrict__ start_xy,
                                                          int64_t *__restrict__ hull_idx, int64_t *__restrict__ tmp,
                                                          const double2 *__restrict__ verts,
                                                          const int32_t *__restrict__ vcount, int vstride)
{
    __shared__ int found_at, n_named;
    const int64_t s = blockIdx.x, off = offsets[s], h = n_hull[s];
    if (found[s] < 0 || h <= 0)
        return;
    const double sx = start_xy[2 * s], sy = start_xy[2 * s + 1];
    if (threadIdx.x == 0) {
        found_at = 0x7fffffff;
        n_named = 0;
    }
    __syncthreads();
    const int nv = verts && vcount ? vcount[s] : -1;
    const double2 *list = verts ? verts + s * (int64_t)vstride : nullptr;
    // tmp: the walk's vertices, the bit above the index set where Qhull's list names the vertex
    for (int64_t i = threadIdx.x; i < h; i += 256) {
        const int64_t v = hull_idx[off + i];
        const double px = ux[off + v], py = uy[off + v];
        bool named = nv < 3;
        for (int k = 0; k < nv && !named; k++)
            named = list[k].x == px && list[k].y == py;
        tmp[off + i] = v | (named ? (int64_t)1 << 62 : 0);
        if (named && nv >= 3) atomicAdd(&n_named, 1);
        if (px == sx && py == sy)
            atomicMin(&found_at, (int)i);
    }
    __syncthreads();
    const int64_t p0 = found_at == 0x7fffffff ? 0 : found_at;
    const bool reduce = nv >= 3 && n_named == nv && nv < h;           // (every listed point is a vertex of the walk)
    const int64_t mask = ((int64_t)1 << 62) - 1;
    if (!reduce) {
        for (int64_t i = threadIdx.x; i < h; i += 256)
            hull_idx[off + i] = tmp[off + (i + p0) % h] & mask;
        return;
    }
    if (threadIdx.x != 0)
        return;
    // one thread closes the gaps, in order
    int64_t kept = 0;
    for (int64_t i = 0; i < h; i++) {
        const int64_t t = tmp[off + (i + p0) % h];
        if (t >> 62) hull_idx[off + kept++] = t & mask;
    }
    n_hull[s] = kept;
}

}  // namespace

int qhull_start_host(const double *ux, const double *uy, const int64_t *offsets, const int64_t *n_unique,
                     int64_t n_slices, int32_t n_threads, const int64_t *select, int64_t *start, uint8_t *sure,
                     int64_t *walk, int64_t *n_walk)
{
    const int T = (int)std::max<int64_t>(
        1, std::min<int64_t>(n_slices, n_threads > 0 ? n_threads
                                                    : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()))));
    // a worker that runs out of memory must not end the process: it notes the failure and the call reports it
    std::atomic<int> failed{0};
    auto work = [&](int t) {
        try {
        for (int64_t s = t; s < n_slices; s += T) {
            const int64_t off = offsets[s], m = n_unique ? n_unique[s] : offsets[s + 1] - offsets[s];
            start[s] = -1;
            if (sure) sure[s] = 0;
            if (n_walk) n_walk[s] = 0;
            if ((select && select[s] != PCFIT_HULL_START_QHULL) || m < 3 || m >= (1ll << 31)) continue;
            Replay r(ux + off, uy + off, m);
            if (!r.run()) continue;
            int64_t first = -1;
            r.walk(walk ? walk + off : nullptr, m, &first);
            start[s] = first;
            if (sure) sure[s] = r.sure() ? 1 : 0;
            if (n_walk) n_walk[s] = r.count();
        }
        } catch (...) {
            failed.store(1, std::memory_order_relaxed);
        }
    };
    if (T == 1)
        work(0);
    else {
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < T; t++) pool.emplace_back(work, t);
        } catch (...) {                       // no more threads: the strides of the missing ones stay undone
            failed.store(1, std::memory_order_relaxed);
        }
        work(0);
        for (auto &th : pool) th.join();
    }
    if (failed.load()) {
        set_error("qhull_order: out of host memory or threads during the replay");
        return PCFIT_ERR_INVALID;
    }
    return PCFIT_OK;
}

int hull_rotate_dev(const int64_t *offsets, int64_t *n_hull, const double *ux, const double *uy,
                    const int64_t *found, const double *start_xy, int64_t n_slices, int64_t *hull_idx, int64_t *tmp,
                    hipStream_t st, const void *verts, const int32_t *vcount, int vstride)
{
    if (n_slices <= 0) return PCFIT_OK;
    hull_rotate_kernel<<<dim3((unsigned)n_slices), dim3(256), 0, st>>>(offsets, n_hull, ux, uy, found, start_xy, hull_idx,
                                                                       tmp, static_cast<const double2 *>(verts), vcount,
                                                                       vstride);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API int pcfit_qhull_order_batch(const double *ux, const double *uy, const int64_t *offsets,
                                      const int64_t *n_unique, int64_t n_slices, int32_t n_threads, int64_t *start,
                                      uint8_t *sure, int64_t *walk, int64_t *n_walk)
{
    PCFIT_REQUIRE(n_slices >= 0, "qhull_order: n_slices < 0");
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(ux && uy && offsets && start, "qhull_order: null pointer");
    try {
        return qhull_start_host(ux, uy, offsets, n_unique, n_slices, n_threads, nullptr, start, sure, walk, n_walk);
    } catch (const std::exception &e) {
        set_error("qhull_order: %s", e.what());
        return PCFIT_ERR_INVALID;
    }
}

