// The replay of Qhull's 2-D construction (see qhull_order.hip for what is replayed and why) on the device: one
// workgroup per slice of distinct points.  The facet list, with Qhull's list moves, is kept by one thread in
// LDS; the work that grows with the slice - the partition of the points into outside sets and of the outside
// sets of visible facets among the two new facets, and the search of every set's farthest point - runs
// across the workgroup.  The order of the points inside an outside set is not kept (it decides nothing but ties).
// The kernel follows the host replay decision for decision.  Points that lie below both new facets of an
// insertion by less than a roundoff unit are listed during the parallel pass and taken through Qhull's horizon
// search by the one thread afterwards (whatever mode Qhull's search is in by then, it ends in that walk).  Where
// the order inside a set would matter - two points tie for the farthest of a set, a point lies 4 to 8 roundoff
// units above the first new facet, two facets get their first point through the horizon search in one
// insertion - or one of the fixed capacities below runs out, or the replay ends with more vertices than
// the exact hull has, the slice is handed back (start = QHULL_DEFER) and the host replay decides it.
#include <stdlib.h>

#include "pcfit_internal.h"

namespace pcfit {
__device__ unsigned long long g_qhull_reason[16];      // slices handed back, by cause (tools and tests)
// Optional phase timing (tools only; -DPCFIT_PHASE_CLOCK): thread 0 of one workgroup in 16 adds the clocks spent
// between marks to g_qphase[k] (0 extremes, 1 simplex, 2 first partition, 3 placing, 4 the list's thread,
// 5 partition of visible sets, 6 farthest points, 7 the end).
#ifdef PCFIT_PHASE_CLOCK
__device__ unsigned long long g_qphase[8];
struct QPhase {
    long long last;
    __device__ __forceinline__ bool on() const { return threadIdx.x == 0 && (blockIdx.x & 15) == 0; }
    __device__ __forceinline__ void init() { last = on() ? wall_clock64() : 0; }
    __device__ __forceinline__ void mark(int k)
    {
        if (on()) {
            const long long now = wall_clock64();
            atomicAdd(&g_qphase[k], (unsigned long long)(now - last));
            last = now;
        }
    }
};
#else
struct QPhase {
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void mark(int) {}
};
#endif
namespace {

constexpr int QT = 256;          // threads of a workgroup
constexpr int QF_MAX = 480;      // live facets of a slice (slot 0 is the list's tail): the kernel's roomiest form
constexpr int QP = 5;            // pieces an outside set may lie in
constexpr int QV = 6;            // visible facets of one insertion
constexpr int QU = 32;           // points of one insertion that go through the horizon search
constexpr int64_t QHULL_DEFER = QHULL_START_DEFER;
constexpr unsigned IDX_MASK = 0x7fffffffu;      // a piece's start; the top bit names the buffer it lies in

template <int QF>
struct Shared {
    static constexpr int FACETS = QF;
    double n0[QF], n1[QF], off[QF];
    double far_d[QF];
    int v0[QF], v1[QF], total[QF], far_idx[QF];
    int plo[QF][QP], pcnt[QF][QP];
    short nb0[QF], nb1[QF], prev[QF], next[QF], free_slot[QF];
    unsigned char np[QF], top[QF], visible[QF];
    // one insertion
    int c0[QV * QP], c1[QV * QP];
    double wave_d[2][QT / 64];
    int wave_i[2][QT / 64], wave_tie[2][QT / 64], wave_flag[QT / 64];
    int vis[QV], nvis, newf[2], apex, state, reason, n_free, head, nextf;
    // points below both new facets by less than a roundoff unit: Qhull searches the horizon for them
    int unc_p[QU], unc_slot[QU], unc_from[QU], n_unc, gap_used[QV * QP];
    double unc_d[QU];
    unsigned short visit[QF], visit_id;
    short copl[64];
    double red_d[QT / 64];
    int red_i[QT / 64];
    double ext_k1[4][QT / 64], ext_k2[4][QT / 64];
    int ext_at[4][QT / 64];
    double thr[6];               // distround, minvisible, minoutside, distoutside, band, (unused)
    double search[3];            // the simplex search: best determinant so far, the one that ends it, near zero
    int ext[4];
};

template <class SH>
__device__ inline double plane_dist(const SH &S, int f, double x, double y) { return S.off[f] + x * S.n0[f] + y * S.n1[f]; }

template <class SH>
__device__ inline void set_plane(SH &S, int f, const double *x, const double *y)
{
    const int a = S.v0[f], b = S.v1[f];
    const double nx = y[b] - y[a], ny = x[a] - x[b];
    double norm = sqrt(nx * nx + ny * ny);
    if (!S.top[f]) norm = -norm;
    S.n0[f] = nx / norm;
    S.n1[f] = ny / norm;
    S.off[f] = -(x[a] * S.n0[f] + y[a] * S.n1[f]);
}

// qh_removefacet / qh_appendfacet on the list in LDS (one thread)
template <class SH>
__device__ inline void list_remove(SH &S, int k)
{
    if (S.nextf == k) S.nextf = S.next[k];
    if (S.head == k) { S.head = S.next[k]; S.prev[S.next[k]] = -1; }
    else { S.next[S.prev[k]] = S.next[k]; S.prev[S.next[k]] = S.prev[k]; }
}
template <class SH>
__device__ inline void list_append(SH &S, int k)
{
    if (S.nextf == 0) S.nextf = k;
    S.prev[k] = S.prev[0];
    S.next[k] = 0;
    if (S.head == 0) { S.head = k; S.prev[k] = -1; }
    else S.next[S.prev[0]] = (short)k;
    S.prev[0] = (short)k;
}

// Is point a before point b in the lexicographic order of (x, y)?  For sorted distinct input that is a < b; for a
// slice as it comes (any order, duplicates) it names the point the sorted distinct array would list first.
__device__ inline bool lex_before(const double *x, const double *y, int a, int b)
{
    return x[a] < x[b] || (x[a] == x[b] && y[a] < y[b]);
}
__device__ inline bool same_point(const double *x, const double *y, int a, int b) { return x[a] == x[b] && y[a] == y[b]; }

// The four coordinate extremes in one pass over the slice: least x, greatest x, least y, greatest y, each the
// lexicographically first point among equals (what "first in the sorted distinct array" means).  Each is an
// argmin over a pair key: (x, y), (-x, y), (y, x), (-y, x).  Every thread gets the result.
struct Key {
    double k1, k2;
    int at;
};
__device__ inline void key_take(Key &a, double k1, double k2, int at)
{
    if (at >= 0 && (a.at < 0 || k1 < a.k1 || (k1 == a.k1 && k2 < a.k2))) { a.k1 = k1; a.k2 = k2; a.at = at; }
}
template <int M>
__device__ __forceinline__ void key_fold(Key &e, int lane)
{
    if constexpr (M >= 1) {
        key_take(e, lane_xor<M>(e.k1, lane), lane_xor<M>(e.k2, lane), lane_xor_i32<M>(e.at, lane));
        key_fold<M / 2>(e, lane);
    }
}
template <class SH>
__device__ void block_extremes(SH &S, const double *x, const double *y, int m, int out[4])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Key e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = {0.0, 0.0, -1};
    // the loads of the next trip are in flight while this one is looked at (past the end they read the last point)
    const int last = m - 1;
    auto fetch = [&](double (&qx)[4], double (&qy)[4], int base) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * QT + tid;
            qx[u] = x[i < last ? i : last];
            qy[u] = y[i < last ? i : last];
        }
    };
    auto take = [&](const double (&px)[4], const double (&py)[4], int base) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * QT + tid, at = i < m ? i : -1;
            key_take(e[0], px[u], py[u], at);
            key_take(e[1], -px[u], py[u], at);
            key_take(e[2], py[u], px[u], at);
            key_take(e[3], -py[u], px[u], at);
        }
    };
    double ax[4], ay[4], bx[4], by[4];
    fetch(ax, ay, 0);
    for (int base = 0; base < m; base += QT * 8) {
        fetch(bx, by, base + QT * 4);
        take(ax, ay, base);
        fetch(ax, ay, base + QT * 8);
        if (base + QT * 4 < m) take(bx, by, base + QT * 4);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) key_fold<32>(e[k], lane);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 4; k++) { S.ext_k1[k][wave] = e[k].k1; S.ext_k2[k][wave] = e[k].k2; S.ext_at[k][wave] = e[k].at; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        Key r = {0.0, 0.0, -1};
        for (int w = 0; w < QT / 64; w++) key_take(r, S.ext_k1[k][w], S.ext_k2[k][w], S.ext_at[k][w]);
        out[k] = r.at;
    }
}

// The farthest point of each of two classes, over what every thread has seen: (distance, point) per class in
// out_d / out_i, and whether the greatest distance was met by more than one point; `flag` comes back as the "or"
// of every thread's.  One barrier: the partials of the last call were read before the barrier that every caller
// passes between two calls.
template <int M>
__device__ __forceinline__ void far_fold(double &d, int &at, int &tie, const double *x, const double *y, int lane)
{
    if constexpr (M >= 1) {
        const double od = lane_xor<M>(d, lane);
        const int oi = lane_xor_i32<M>(at, lane), ot = lane_xor_i32<M>(tie, lane);
        if (oi >= 0) {
            if (at < 0 || od > d) { d = od; at = oi; tie = ot; }
            else if (od == d && !same_point(x, y, oi, at)) tie = 1;
        }
        far_fold<M / 2>(d, at, tie, x, y, lane);
    }
}
// t0, t1: this thread has met its own greatest distance at two different points already.
template <class SH>
__device__ void block_far(SH &S, const double *x, const double *y, double d0, int i0, int t0, double d1, int i1,
                          int t1, double out_d[2], int out_i[2], int out_tie[2], bool &flag)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double d[2] = {d0, d1};
    int at[2] = {i0, i1}, tie[2] = {t0, t1};
    far_fold<32>(d[0], at[0], tie[0], x, y, lane);
    far_fold<32>(d[1], at[1], tie[1], x, y, lane);
    const bool wave_flag = __builtin_amdgcn_ballot_w64(flag) != 0ull;
    if (lane == 0) {
        for (int c = 0; c < 2; c++) { S.wave_d[c][wave] = d[c]; S.wave_i[c][wave] = at[c]; S.wave_tie[c][wave] = tie[c]; }
        S.wave_flag[wave] = wave_flag ? 1 : 0;
    }
    __syncthreads();
    flag = false;
    for (int w = 0; w < QT / 64; w++) flag = flag || S.wave_flag[w] != 0;
    for (int c = 0; c < 2; c++) {
        out_d[c] = 0.0;
        out_i[c] = -1;
        out_tie[c] = 0;
        for (int w = 0; w < QT / 64; w++) {
            const double od = S.wave_d[c][w];
            const int oi = S.wave_i[c][w], ot = S.wave_tie[c][w];
            if (oi < 0) continue;
            if (out_i[c] < 0 || od > out_d[c]) { out_d[c] = od; out_i[c] = oi; out_tie[c] = ot; }
            else if (od == out_d[c] && !same_point(x, y, oi, out_i[c])) out_tie[c] = 1;
        }
    }
}

// Where the points of one turn (U per thread) go in their class's run: one LDS atomic per wave and class for the
// whole turn, its answer read by every lane from the lane that asked; slot[u] = the place of point u when it is of
// that class.  (The order inside a run is not Qhull's and need not be.)
template <int U>
__device__ __forceinline__ void wave_slots(int *counter, const bool (&mine)[U], int (&slot)[U])
{
    const int lane = threadIdx.x & 63;
    unsigned long long votes[U];
    int total = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        votes[u] = __builtin_amdgcn_ballot_w64(mine[u]);
        total += __popcll(votes[u]);
    }
    int base = 0;
    if (total) {                                        // (uniform over the wave)
        if (lane == 0) base = atomicAdd(counter, total);
        base = __builtin_amdgcn_readfirstlane(base);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        slot[u] = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(votes[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)votes[u], 0u));
        base += __popcll(votes[u]);
    }
}

// QF: facet slots (the LDS a workgroup takes grows with them, and the workgroups a CU holds shrink: every phase of
// this kernel waits - for memory, for the one thread that keeps the list - so it is the number of slices in
// flight that fills the chip); WG: workgroups to a CU the registers are budgeted for.
template <int QF, int WG>
__global__ __launch_bounds__(QT, WG) void qhull_order_kernel(const double *__restrict__ ux, const double *__restrict__ uy,
                                                         const int64_t *__restrict__ offsets,
                                                         const int64_t *__restrict__ n_unique,
                                                         const int64_t *__restrict__ n_hull,
                                                         const int64_t *__restrict__ select, int *__restrict__ buf_a,
                                                         int *__restrict__ buf_b, int64_t *__restrict__ start,
                                                         double *__restrict__ start_xy)
{
    __shared__ Shared<QF> S;
    const int64_t s = blockIdx.x, off = offsets[s];
    const int tid = threadIdx.x;
    if (select && select[s] != PCFIT_HULL_START_QHULL) {
        if (tid == 0) start[s] = -1;
        return;
    }
    const int64_t m64 = n_unique ? n_unique[s] : offsets[s + 1] - off;
    if (m64 < 3 || n_hull[s] < 3) {
        if (tid == 0) start[s] = -1;
        return;
    }
    if (m64 >= (1ll << 30) || n_hull[s] > QF - 2) {                // more facets at the end than there are slots
        if (tid == 0) { start[s] = QHULL_DEFER; atomicAdd(&g_qhull_reason[8], 1ull); }
        return;
    }
    const int m = (int)m64;
    const double *x = ux + off, *y = uy + off;
    int *const bufs0 = buf_a + off, *const bufs1 = buf_b + off;

    // (the free list and the visit marks: idle until the simplex is made, several barriers from here)
    for (int k = tid; k < QF; k += QT) {
        S.free_slot[k] = (short)(QF - 1 - k);       // slots QF-1 .. 1 (0 is the tail)
        S.visit[k] = 0;
    }
    // qh_maxmin, qh_detroundoff, qh_maxsimplex
    QPhase ph;
    ph.init();
    int ext4[4];
    block_extremes(S, x, y, m, ext4);
    ph.mark(0);
    const int xlo = ext4[0], xhi = ext4[1], ylo = ext4[2], yhi = ext4[3];
    if (tid == 0) {
        S.state = 0;
        S.reason = 0;
        const int ext[4] = {xlo, xhi, ylo, yhi};
        const double ax = fmax(fabs(x[xlo]), fabs(x[xhi])), ay = fmax(fabs(y[ylo]), fabs(y[yhi]));
        const double maxabs = fmax(ax, ay), sumabs = ax + ay;
        const double maxwidth = fmax(x[xhi] - x[xlo], y[yhi] - y[ylo]);
        const double eps = 2.220446049250313e-16;
        const double distround = eps * (2 * fmin(sqrt(2.0) * maxabs, sumabs) * 1.01 + maxabs);
        S.thr[0] = distround;
        S.thr[1] = 2 * distround;                  // minvisible
        S.thr[2] = 4 * distround;                  // minoutside
        S.thr[3] = 8 * distround;                  // distoutside
        S.thr[4] = 64 * distround + 8 * distround; // every threshold lies inside this band about zero
        int minx = ext[0], maxx = ext[0];
        for (int k = 0; k < 4; k++) {
            if (x[ext[k]] > x[maxx]) maxx = ext[k];
            if (x[ext[k]] < x[minx]) minx = ext[k];
        }
        int third = -1;
        double bestdet = -1.0;
        if (x[minx] != x[maxx])
            for (int k = 0; k < 4; k++) {
                const int i = ext[k];
                if (same_point(x, y, i, minx) || same_point(x, y, i, maxx)) continue;
                const double det = fabs((x[minx] - x[i]) * (y[maxx] - y[i]) - (x[maxx] - x[i]) * (y[minx] - y[i]));
                if (det > bestdet) { bestdet = det; third = i; }
            }
        const double target = (x[maxx] - x[minx]) * maxwidth, nearzero = 10 * (80 * sumabs * eps);
        S.ext[0] = minx; S.ext[1] = maxx; S.ext[2] = third;
        S.search[0] = bestdet; S.search[1] = 10 * 1e-3 * target; S.search[2] = nearzero;
        S.nvis = 0;
        if (x[minx] == x[maxx])
            { S.state = 2; S.reason = 1; }          // all points on one vertical line: no hull
        else if (third < 0 || bestdet < nearzero || bestdet / target < 1e-3)
            S.nvis = 1;                             // a flat simplex: Qhull looks through all other points
        S.vis[0] = xlo; S.vis[1] = xhi; S.vis[2] = ylo; S.vis[3] = yhi;
    }
    __syncthreads();
    if (S.state == 2) {
        if (tid == 0) { start[s] = QHULL_DEFER; atomicAdd(&g_qhull_reason[S.reason & 15], 1ull); }
        return;
    }
    if (S.nvis == 1) {
        // qh_maxsimplex's search, in the points' order: it stops at the first point that spans a hundredth of what
        // the extents allow (and is not near zero); if none does, the largest determinant wins, the first of equals
        const int minx = S.ext[0], maxx = S.ext[1];
        const double mindet = S.search[1], nearzero = S.search[2];
        const double ax = x[minx], ay = y[minx], bx = x[maxx], by = y[maxx];
        int first = -1, at = -1;
        double best = -1.0;
        auto larger = [&](double ob, int oa, double b, int a) {
            return oa >= 0 && (a < 0 || ob > b || (ob == b && lex_before(x, y, oa, a)));
        };
        for (int i = tid; i < m; i += QT) {
            if (same_point(x, y, i, xlo) || same_point(x, y, i, xhi) || same_point(x, y, i, ylo) || same_point(x, y, i, yhi))
                continue;
            const double det = fabs((ax - x[i]) * (by - y[i]) - (bx - x[i]) * (ay - y[i]));
            if (det > mindet && !(det < nearzero) && (first < 0 || lex_before(x, y, i, first))) first = i;
            if (larger(det, i, best, at)) { best = det; at = i; }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const int of = __shfl_xor(first, d);
            if (of >= 0 && (first < 0 || lex_before(x, y, of, first))) first = of;
            const double ob = __shfl_xor(best, d);
            const int oa = __shfl_xor(at, d);
            if (larger(ob, oa, best, at)) { best = ob; at = oa; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { S.red_d[tid >> 6] = best; S.red_i[tid >> 6] = at; S.c0[tid >> 6] = first; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < QT / 64; w++) {
                const int of = S.c0[w];
                if (of >= 0 && (first < 0 || lex_before(x, y, of, first))) first = of;
                if (larger(S.red_d[w], S.red_i[w], best, at)) { best = S.red_d[w]; at = S.red_i[w]; }
            }
            if (first >= 0) S.ext[2] = first;
            else if (at >= 0 && best > S.search[0]) { S.ext[2] = at; S.search[0] = best; }
            if (first < 0 && (S.ext[2] < 0 || !(S.search[0] > 0.0))) { S.state = 2; S.reason = 1; }
        }
        __syncthreads();
        if (S.state == 2) {
            if (tid == 0) { start[s] = QHULL_DEFER; atomicAdd(&g_qhull_reason[S.reason & 15], 1ull); }
            return;
        }
    }
    if (tid == 0) {
        {
            const int minx = S.ext[0], maxx = S.ext[1], third = S.ext[2];
            // qh_createsimplex, qh_initialhull
            S.n_free = QF - 1;
            S.visit_id = 0;
            S.head = S.nextf = 0;
            S.prev[0] = -1;
            S.next[0] = 0;
            const int vs[3] = {third, maxx, minx};
            int fs[3];
            bool top = true;
            for (int i = 0; i < 3; i++) {
                const int f = S.free_slot[--S.n_free + 0];
                fs[i] = f;
                S.v0[f] = vs[i == 0 ? 1 : 0];
                S.v1[f] = vs[i == 2 ? 1 : 2];
                S.top[f] = top;
                S.visible[f] = 0;
                S.np[f] = 0;
                S.total[f] = 0;
                S.far_idx[f] = -1;
                list_append(S, f);
                top = !top;
            }
            S.nb0[fs[0]] = (short)fs[1]; S.nb1[fs[0]] = (short)fs[2];
            S.nb0[fs[1]] = (short)fs[0]; S.nb1[fs[1]] = (short)fs[2];
            S.nb0[fs[2]] = (short)fs[0]; S.nb1[fs[2]] = (short)fs[1];
            const double cx = (x[minx] + x[maxx] + x[third]) / 3.0, cy = (y[minx] + y[maxx] + y[third]) / 3.0;
            set_plane(S, fs[0], x, y);
            if (plane_dist(S, fs[0], cx, cy) > 0)
                for (int i = 0; i < 3; i++) S.top[fs[i]] = !S.top[fs[i]];
            for (int i = 0; i < 3; i++) {
                set_plane(S, fs[i], x, y);
                if (!(plane_dist(S, fs[i], cx, cy) < -S.thr[4])) { S.state = 2; S.reason = 2; }
            }
            S.vis[0] = fs[0]; S.vis[1] = fs[1]; S.vis[2] = fs[2];
            S.ext[0] = minx; S.ext[1] = maxx; S.ext[2] = third;
            S.c0[0] = S.c0[1] = S.c0[2] = 0;
        }
    }
    __syncthreads();
    if (S.state == 2) {
        if (tid == 0) { start[s] = QHULL_DEFER; atomicAdd(&g_qhull_reason[S.reason & 15], 1ull); }
        return;
    }
    ph.mark(1);
    // qh_partitionall: every point to the first of the three facets it lies outside of (counts, then places)
    {
        const int f0 = S.vis[0], f1 = S.vis[1], f2 = S.vis[2];
        const int e0 = S.ext[0], e1 = S.ext[1], e2 = S.ext[2];
        const double thr = S.thr[3], minoutside = S.thr[2];
        double fd[3] = {0, 0, 0};
        int fi[3] = {-1, -1, -1}, ft[3] = {0, 0, 0};
        bool doubt = false;
        const double pl[3][3] = {{S.off[f0], S.n0[f0], S.n1[f0]}, {S.off[f1], S.n0[f1], S.n1[f1]}, {S.off[f2], S.n0[f2], S.n1[f2]}};
        // every point's class goes to a byte array in the second index buffer (idle until the first insertion),
        // so that the placing pass below reads one byte a point instead of its coordinates
        unsigned char *cls = reinterpret_cast<unsigned char *>(bufs1);
        const double e0x = x[e0], e0y = y[e0], e1x = x[e1], e1y = y[e1], e2x = x[e2], e2y = y[e2];
        // (the loads of the next turn are in flight while this one is classified; a wave counts its classes in
        // scalar registers and adds them to the workgroup's once, at the end)
        int n_of[3] = {0, 0, 0};
        const int last = m - 1;
        auto fetch = [&](double (&qx)[4], double (&qy)[4], int base) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + u * QT + tid;
                qx[u] = x[i < last ? i : last];
                qy[u] = y[i < last ? i : last];
            }
        };
        auto classify = [&](const double (&px)[4], const double (&py)[4], int base) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + u * QT + tid;
                int c = 3;
                double d = 0.0;
                const bool vertex = (px[u] == e0x && py[u] == e0y) || (px[u] == e1x && py[u] == e1y) ||
                                    (px[u] == e2x && py[u] == e2y);
                if (!vertex && i < m) {
                    double most = -1e300;
                    int most_k = 3;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        if (c != 3) continue;
                        d = pl[k][0] + px[u] * pl[k][1] + py[u] * pl[k][2];
                        if (d >= thr) c = k;
                        else if (d > most) { most = d; most_k = k; }
                    }
                    // outside of no facet by the first threshold: Qhull looks at such a point once more and gives
                    // it to the facet it is farthest from if that is outside by the lower threshold
                    if (c == 3 && most >= minoutside) { c = most_k; d = most; }
                }
                if (i < m) cls[i] = (unsigned char)c;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    n_of[k] += __popcll(__builtin_amdgcn_ballot_w64(c == k));
                    if (c == k) {
                        if (fi[k] < 0 || d > fd[k]) { fd[k] = d; fi[k] = i; ft[k] = 0; }
                        else if (d == fd[k] && !same_point(x, y, i, fi[k])) ft[k] = 1;
                    }
                }
            }
        };
        {
            double ax[4], ay[4], bx[4], by[4];
            fetch(ax, ay, 0);
            for (int base = 0; base < m; base += QT * 8) {
                fetch(bx, by, base + QT * 4);
                classify(ax, ay, base);
                fetch(ax, ay, base + QT * 8);
                if (base + QT * 4 < m) classify(bx, by, base + QT * 4);
            }
        }
        if ((tid & 63) == 0)
            for (int k = 0; k < 3; k++)
                if (n_of[k]) atomicAdd(&S.c0[k], n_of[k]);
        if (doubt) { S.state = 2; S.reason = 3; }
        // the farthest point of each of the three sets
        double od[2];
        int oi[2], ot[2];
        double far_d[3];
        int far_i[3], far_t[3];
        bool none = false;
        block_far(S, x, y, fd[0], fi[0], ft[0], fd[1], fi[1], ft[1], od, oi, ot, none);
        far_d[0] = od[0]; far_i[0] = oi[0]; far_t[0] = ot[0] | 0;
        far_d[1] = od[1]; far_i[1] = oi[1]; far_t[1] = ot[1];
        __syncthreads();                           // (the partials are read: they may be written again)
        block_far(S, x, y, fd[2], fi[2], ft[2], 0.0, -1, 0, od, oi, ot, none);
        far_d[2] = od[0]; far_i[2] = oi[0]; far_t[2] = ot[0];
        __syncthreads();
        if (tid == 0) {
            const int cnt[3] = {S.c0[0], S.c0[1], S.c0[2]};
            const int fs[3] = {f0, f1, f2};
            int at = 0;
            for (int c = 0; c < 3; c++) {
                if (far_t[c]) { S.state = 2; S.reason = 4; }
                S.total[fs[c]] = cnt[c];
                S.far_idx[fs[c]] = far_i[c];
                S.far_d[fs[c]] = far_d[c];
                if (cnt[c] > 0) { S.np[fs[c]] = 1; S.plo[fs[c]][0] = at; S.pcnt[fs[c]][0] = cnt[c]; }
                S.c1[c] = at;                      // where the set's run starts
                S.c0[c] = 0;
                at += cnt[c];
            }
            // qh_furthestnext: the facet with the farthest outside point goes to the front of the list
            int bestf = 0;
            double bestd = -1e300;
            for (int c = 0; c < 3; c++)
                if (cnt[c] > 0 && far_d[c] > bestd) { bestd = far_d[c]; bestf = fs[c]; }
            if (bestf && bestf != S.head) {
                list_remove(S, bestf);
                S.prev[bestf] = -1;
                S.next[bestf] = (short)S.head;
                S.prev[S.head] = (short)bestf;
                S.head = bestf;
            }
            S.nextf = S.head;
        }
        __syncthreads();
        if (S.state == 2) {
            if (tid == 0) { start[s] = QHULL_DEFER; atomicAdd(&g_qhull_reason[S.reason & 15], 1ull); }
            return;
        }
        ph.mark(2);
        int *A = bufs0;
        const int run0 = S.c1[0], run1 = S.c1[1], run2 = S.c1[2];
        for (int base = 0; base < m; base += QT * 4) {
            int c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + u * QT + tid;
                c[u] = i < m ? (int)cls[i] : 3;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bool mine[4];
                int slot[4];
#pragma unroll
                for (int u = 0; u < 4; u++) mine[u] = c[u] == k;
                wave_slots<4>(&S.c0[k], mine, slot);
                const int run = k == 0 ? run0 : k == 1 ? run1 : run2;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (mine[u]) A[run + slot[u]] = base + u * QT + tid;
            }
        }
        __syncthreads();
    }
    // qh_buildhull.  One thread keeps the facet list; between two of its turns the workgroup partitions the outside
    // sets of the visible facets.  Two barriers per insertion: behind the thread's turn, and inside the search of the
    // farthest points.
    unsigned long long visits = 0;
    int steps = 0;
    // the next insertion: qh_nextfurthest, qh_findhorizon, qh_makenew_simplicial
    auto next_insertion = [&]() {
            const double minvisible = S.thr[1];
            while (S.nextf != 0 && S.total[S.nextf] == 0) S.nextf = S.next[S.nextf];           // qh_nextfurthest
            if (S.nextf == 0)
                S.state = 1;
            else {
                const int facet = S.nextf, apex = S.far_idx[facet];
                const double ax = x[apex], ay = y[apex];
                S.apex = apex;
                // qh_findhorizon: visible facets move to the end of the list as they are found
                list_remove(S, facet);
                list_append(S, facet);
                S.visible[facet] = 1;
                int nvis = 1;
                S.vis[0] = facet;
                for (int k = 0; k < nvis && S.state == 0; k++) {
                    const int v = S.vis[k];
                    for (int j = 0; j < 2; j++) {
                        const int g = j ? S.nb1[v] : S.nb0[v];
                        if (S.visible[g]) continue;
                        const double d = plane_dist(S, g, ax, ay);
                        if (d > minvisible) {
                            if (nvis == QV) { S.state = 2; S.reason = 6; break; }
                            list_remove(S, g);
                            list_append(S, g);
                            S.visible[g] = 1;
                            S.vis[nvis++] = g;
                        }
                    }
                }
                S.nvis = nvis;
                steps++;
                for (int k = 0; k < nvis; k++) visits += (unsigned long long)S.total[S.vis[k]];
                // qh_makenew_simplicial: one facet per horizon ridge, the apex first
                int n_new = 0;
                for (int k = 0; k < nvis && S.state == 0; k++) {
                    const int v = S.vis[k];
                    for (int j = 0; j < 2; j++) {
                        const int g = j ? S.nb1[v] : S.nb0[v];
                        if (S.visible[g]) continue;
                        const int hskip = S.nb0[g] == v ? 0 : 1;
                        if (n_new == 2 || S.n_free == 0) { S.state = 2; S.reason = n_new == 2 ? 7 : 8; break; }
                        const int nf = S.free_slot[--S.n_free];
                        S.v0[nf] = apex;
                        S.v1[nf] = hskip ? S.v0[g] : S.v1[g];
                        S.top[nf] = S.top[g] ? (hskip & 1) != 0 : (hskip & 1) == 0;
                        S.visible[nf] = 0;
                        S.np[nf] = 0;
                        S.total[nf] = 0;
                        S.far_idx[nf] = -1;
                        S.nb0[nf] = (short)g;
                        if (hskip) S.nb1[g] = (short)nf; else S.nb0[g] = (short)nf;
                        set_plane(S, nf, x, y);
                        list_append(S, nf);
                        S.newf[n_new++] = nf;
                    }
                }
                if (S.state == 0 && n_new != 2) { S.state = 2; S.reason = 7; }
                if (S.state == 0) {
                    S.nb1[S.newf[0]] = (short)S.newf[1];
                    S.nb1[S.newf[1]] = (short)S.newf[0];
                    for (int k = 0; k < QV * QP; k++) S.c0[k] = S.c1[k] = S.gap_used[k] = 0;
                    S.n_unc = 0;
                }
            }
            };
    ph.mark(3);
    if (tid == 0) next_insertion();
    __syncthreads();
    ph.mark(4);
    for (int step = 0; step < m + 8 && S.state == 0; step++) {
        // qh_partitionvisible: the outside points of the visible facets to the first new facet they lie outside of
        const int nvis = S.nvis, a0 = S.newf[0], a1 = S.newf[1], apex = S.apex;
        const double apex_x = x[apex], apex_y = y[apex];
        const double minoutside = S.thr[2], reach = S.thr[3], distround = S.thr[0];
        const double a0_n0 = S.n0[a0], a0_n1 = S.n1[a0], a0_off = S.off[a0];
        const double a1_n0 = S.n0[a1], a1_n1 = S.n1[a1], a1_off = S.off[a1];
        double fd0 = 0.0, fd1 = 0.0;
        int fi0 = -1, fi1 = -1, ft0 = 0, ft1 = 0;
        bool doubt = false;
        for (int k = 0; k < nvis; k++) {
            const int v = S.vis[k], np = S.np[v];
            for (int q = 0; q < np; q++) {
                const unsigned word = (unsigned)S.plo[v][q];
                const int lo = (int)(word & IDX_MASK), from = (int)(word >> 31), cnt = S.pcnt[v][q], slot = k * QP + q;
                const int *src = from ? bufs1 : bufs0;
                int *dst = from ? bufs0 : bufs1;
                // four points a thread and turn: the gathers of their coordinates are in flight together
                for (int base = 0; base < cnt; base += QT * 4) {
                    int p[4];
                    double px[4], py[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = base + u * QT + tid;
                        p[u] = i < cnt ? src[lo + i] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        px[u] = p[u] >= 0 ? x[p[u]] : apex_x;
                        py[u] = p[u] >= 0 ? y[p[u]] : apex_y;
                    }
                    int cls[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        int c = 2;
                        double d = 0.0;
                        if (px[u] != apex_x || py[u] != apex_y) {
                            d = a0_off + px[u] * a0_n0 + py[u] * a0_n1;
                            if (d >= minoutside) {
                                c = 0;
                                if (d < reach) doubt = true;        // (Qhull's other search mode may prefer the second facet)
                            } else {
                                const double d0 = d;
                                d = a1_off + px[u] * a1_n0 + py[u] * a1_n1;
                                if (d >= minoutside) c = 1;
                                else if (fmax(d0, d) >= -distround) {
                                    // below both by less than a roundoff unit: the horizon search, after the pass
                                    const int at = atomicAdd(&S.n_unc, 1);
                                    if (at < QU) {
                                        S.unc_p[at] = p[u];
                                        S.unc_slot[at] = slot;
                                        S.unc_from[at] = d > d0 ? a1 : a0;
                                        S.unc_d[at] = fmax(d0, d);
                                    }
                                }
                            }
                        }
                        cls[u] = c;
                        if (c == 0) {
                            if (fi0 < 0 || d > fd0) { fd0 = d; fi0 = p[u]; ft0 = 0; }
                            else if (d == fd0 && !same_point(x, y, p[u], fi0)) ft0 = 1;
                        } else if (c == 1) {
                            if (fi1 < 0 || d > fd1) { fd1 = d; fi1 = p[u]; ft1 = 0; }
                            else if (d == fd1 && !same_point(x, y, p[u], fi1)) ft1 = 1;
                        }
                    }
                    // the points of the first new facet upwards from the piece's start, those of the second
                    // downwards from its end
                    bool m0[4], m1[4];
                    int s0[4], s1[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { m0[u] = cls[u] == 0; m1[u] = cls[u] == 1; }
                    wave_slots<4>(&S.c0[slot], m0, s0);
                    wave_slots<4>(&S.c1[slot], m1, s1);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (m0[u]) dst[lo + s0[u]] = p[u];
                        if (m1[u]) dst[lo + cnt - 1 - s1[u]] = p[u];
                    }
                }
            }
        }
        ph.mark(5);
        double od[2];
        int oi[2], ot[2];
        block_far(S, x, y, fd0, fi0, ft0, fd1, fi1, ft1, od, oi, ot, doubt);
        ph.mark(6);
        if (tid == 0) {
            if (doubt) { S.state = 2; S.reason = 10; }
            if (S.state == 0) {
            if (ot[0] || ot[1]) { S.state = 2; S.reason = 4; }
            const int child[2] = {a0, a1};
            for (int k = 0; k < nvis && S.state == 0; k++) {
                const int v = S.vis[k], np = S.np[v];
                for (int q = 0; q < np; q++) {
                    const unsigned word = (unsigned)S.plo[v][q];
                    const int lo = (int)(word & IDX_MASK), to = (int)(word >> 31) ^ 1, cnt = S.pcnt[v][q], slot = k * QP + q;
                    const int c0 = S.c0[slot], c1 = S.c1[slot];
                    const int at[2] = {lo, lo + cnt - c1}, n[2] = {c0, c1};
                    for (int c = 0; c < 2; c++) {
                        if (!n[c]) continue;
                        const int f = child[c];
                        if (S.np[f] == QP) { S.state = 2; S.reason = 11; break; }
                        S.plo[f][S.np[f]] = (int)((unsigned)at[c] | ((unsigned)to << 31));
                        S.pcnt[f][S.np[f]] = n[c];
                        S.np[f]++;
                        S.total[f] += n[c];
                    }
                }
            }
            S.far_idx[a0] = oi[0];
            S.far_idx[a1] = oi[1];
            S.far_d[a0] = od[0];
            S.far_d[a1] = od[1];
            // qh_findbesthorizon for the points that lie below both new facets by less than a roundoff unit
            // (whatever mode Qhull's search is in by then, it ends in this walk): over neighbours whose distance
            // is no more than the search distance below the best one found; the point joins the facet found if it
            // lies outside of it, and a facet that had no outside points goes to the end of the list
            if (S.n_unc > QU) { S.state = 2; S.reason = 10; }
            int moved = 0;
            const double searchdist = S.thr[0] * 2 + S.thr[1];
            for (int k = 0; k < S.n_unc && S.state == 0; k++) {
                const int p = S.unc_p[k];
                const double px = x[p], py = y[p];
                double bestdist = S.unc_d[k], minsearch = bestdist - searchdist;
                int bestf = S.unc_from[k], facet = bestf, nextfacet = 0, size = 0;
                const unsigned short id = ++S.visit_id;
                S.visit[bestf] = id;
                while (true) {
                    for (int j = 0; j < 2; j++) {
                        const int nb = j ? S.nb1[facet] : S.nb0[facet];
                        if (S.visit[nb] == id || S.visible[nb]) continue;
                        S.visit[nb] = id;
                        const double d = plane_dist(S, nb, px, py);
                        if (d > bestdist) {
                            minsearch = d - searchdist;
                            if (d > bestdist + searchdist) size = 0;
                            bestf = nb;
                            bestdist = d;
                        } else if (d < minsearch)
                            continue;
                        if (nextfacet) {
                            if (size == 0) { S.copl[0] = (short)nextfacet; size = 1; }
                            else if (size < 64) S.copl[size++] = (short)nextfacet;
                        }
                        nextfacet = nb;
                    }
                    facet = nextfacet;
                    if (facet) nextfacet = 0;
                    else if (size == 0) break;
                    else facet = S.copl[--size];
                }
                if (bestdist < minoutside) continue;                // not outside of anything: dropped
                if (bestf == a0 || bestf == a1) { S.state = 2; S.reason = 10; break; }
                if (S.total[bestf] == 0) {
                    if (moved) { S.state = 2; S.reason = 10; break; }       // (two such moves: their order is the sets')
                    moved = 1;
                    list_remove(S, bestf);
                    list_append(S, bestf);
                    S.far_idx[bestf] = p;
                    S.far_d[bestf] = bestdist;
                } else if (S.far_d[bestf] < bestdist) {
                    S.far_idx[bestf] = p;
                    S.far_d[bestf] = bestdist;
                }
                if (S.np[bestf] == QP) { S.state = 2; S.reason = 11; break; }
                // a one-point piece in the gap the dropped points left in the piece this point came from
                const int slot = S.unc_slot[k], v = S.vis[slot / QP], q = slot % QP;
                const unsigned word = (unsigned)S.plo[v][q];
                const int lo = (int)(word & IDX_MASK), to = (int)(word >> 31) ^ 1;
                const int at = lo + S.c0[slot] + S.gap_used[slot]++;
                (to ? bufs1 : bufs0)[at] = p;
                S.plo[bestf][S.np[bestf]] = (int)((unsigned)at | ((unsigned)to << 31));
                S.pcnt[bestf][S.np[bestf]] = 1;
                S.np[bestf]++;
                S.total[bestf]++;
            }
            // qh_deletevisible
            for (int k = 0; k < nvis; k++) {
                const int v = S.vis[k];
                list_remove(S, v);
                S.visible[v] = 0;
                S.free_slot[S.n_free++] = (short)v;
            }
                    }
            if (S.state == 0) next_insertion();
        }
        __syncthreads();
        ph.mark(4);
    }
    if (tid == 0) {
        int64_t out = QHULL_DEFER;
        if (S.state == 1) {
            // scipy's walk from the head of the list; facets in one line are one facet in Qhull's result (it merged
            // them), standing in the list where the newest part stands.  (The free list, the visit marks and the
            // visibility flags are idle by now and hold the walk, the places in the list and the joints.)
            short *walk = S.free_slot;
            unsigned short *place = S.visit;
            unsigned char *joined = S.visible;
            int m = 0, k = 0;
            for (int f = S.head; m < QF;) {
                walk[m++] = (short)f;
                f = S.top[f] ? S.nb0[f] : S.nb1[f];
                if (f == S.head) break;
            }
            for (int g = S.head; g != 0 && k < QF; g = S.next[g]) place[g] = (unsigned short)k++;
            const double radius = 2 * S.thr[0];
            int n_joined = 0;
            auto centrum_dist = [&](int of, int to) {          // a facet's centrum, brought onto its own line, from another's line
                double cx = (x[S.v0[of]] + x[S.v1[of]]) / 2, cy = (y[S.v0[of]] + y[S.v1[of]]) / 2;
                const double own = plane_dist(S, of, cx, cy);
                cx -= own * S.n0[of];
                cy -= own * S.n1[of];
                return plane_dist(S, to, cx, cy);
            };
            for (int i = 0; i < m; i++) {
                const int b = walk[i], a = walk[(i + m - 1) % m];
                joined[i] = fabs(centrum_dist(b, a)) <= radius || fabs(centrum_dist(a, b)) <= radius;
                n_joined += joined[i];
            }
            if (m - n_joined < 3) {
                for (int i = 0; i < m; i++) joined[i] = 0;
                n_joined = 0;
            }
            int first = 0;
            while (joined[first]) first++;
            int count = 0, run_first = first, run_place = 0, best_first = first, best_place = 0x7fffffff;
            for (int t = 0, i = first; t <= m; t++, i = (i + 1) % m) {
                if (t == m || !joined[i]) {
                    if (t > 0 && run_place < best_place) { best_place = run_place; best_first = run_first; }
                    if (t == m) break;
                    count++;
                    run_first = i;
                    run_place = place[walk[i]];
                } else if ((int)place[walk[i]] > run_place)
                    run_place = place[walk[i]];
            }
            // the hull the other kernel found by an exact predicate has these vertices and, where coordinates were
            // rounded, some at the slightest of turns that Qhull does not keep (the turn of the walk takes them out);
            // fewer there than here would mean that something was missed
            if (m == k && count <= n_hull[s]) {
                const int h = walk[best_first];
                out = S.top[h] ? S.v0[h] : S.v1[h];
            } else
                S.reason = 12;
        } else if (S.state == 0)
            S.reason = 13;
        if (out == QHULL_DEFER) atomicAdd(&g_qhull_reason[S.reason & 15], 1ull);
        atomicAdd(&g_qhull_reason[14], (unsigned long long)steps);
        atomicAdd(&g_qhull_reason[15], visits);
        start[s] = out;
        if (start_xy && out >= 0) { start_xy[2 * s] = x[out]; start_xy[2 * s + 1] = y[out]; }
    }
    ph.mark(7);
}

}  // namespace

// start[s] (device): the first vertex of Qhull's order, -1 where there is no hull or the slice was not selected,
// QHULL_DEFER where the host has to decide; start_xy (nullable): its coordinates.  The points of a slice may
// come sorted and distinct (n_unique of them) or as they are (n_unique null: the whole slice, any order,
// duplicates allowed - the answer is the same point, its index then names one of its copies).
// scratch: 2 * total int32 entries.
int qhull_start_dev(const double *ux, const double *uy, const int64_t *offsets, const int64_t *n_unique,
                    const int64_t *n_hull, const int64_t *select, int64_t n_slices, int64_t total, int *scratch,
                    int64_t *start, double *start_xy, hipStream_t st, int64_t max_hull)
{
    if (n_slices <= 0) return PCFIT_OK;
    // The form with the fewest facet slots that holds every hull of the batch twice over (a slice that runs out of
    // slots all the same is handed back): fewer slots, more slices in flight on a CU.
    static const int forced = [] { const char *e = getenv("PCFIT_QHULL_FORM"); return e ? atoi(e) : 0; }();
    const int64_t need = max_hull < 0 ? QF_MAX : 2 * max_hull + 16;
    const int form = forced ? forced : need <= 112 ? 112 : need <= 224 ? 224 : QF_MAX;
    const dim3 grid((unsigned)n_slices), block(QT);
#define PCFIT_QHULL_LAUNCH(QF, WG)                                                                                  \
    qhull_order_kernel<QF, WG><<<grid, block, 0, st>>>(ux, uy, offsets, n_unique, n_hull, select, scratch,          \
                                                       scratch + total, start, start_xy)
    if (form == 112) PCFIT_QHULL_LAUNCH(112, 6);
    else if (form == 113) PCFIT_QHULL_LAUNCH(112, 8);
    else if (form == 224) PCFIT_QHULL_LAUNCH(224, 5);
    else if (form == 225) PCFIT_QHULL_LAUNCH(224, 4);
    else PCFIT_QHULL_LAUNCH(QF_MAX, 3);
#undef PCFIT_QHULL_LAUNCH
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API int pcfit_qhull_start_batch_dev(const double *ux, const double *uy, const int64_t *offsets,
                                          const int64_t *n_unique, const int64_t *n_hull, const int64_t *select,
                                          int64_t n_slices, int64_t total_points, void *scratch, size_t scratch_bytes,
                                          int64_t *start, void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31) && total_points >= 0, "qhull_start: sizes out of range");
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(ux && uy && offsets && n_hull && start, "qhull_start: null pointer");
    if (!scratch || scratch_bytes < (size_t)total_points * 8) {
        set_error("qhull_start: the scratch buffer holds %zu bytes, %zu needed", scratch_bytes, (size_t)total_points * 8);
        return PCFIT_ERR_WORKSPACE;
    }
    return qhull_start_dev(ux, uy, offsets, n_unique, n_hull, select, n_slices, total_points, static_cast<int *>(scratch),
                           start, nullptr, as_stream(stream), -1);
}

// tools and tests: how many slices the kernel handed back since the last reset, by cause (1 flat simplex, 2 simplex
// orientation, 3 first partition near a threshold, 4 tie for the farthest point, 5 horizon near the threshold,
// 6 visible facets, 7 new facets, 8 facet slots, 9 nearly collinear facets, 10 partition near a threshold,
// 11 pieces of a set, 12 vertex count, 13 steps)
PCFIT_API int pcfit_debug_qhull_reasons(int64_t *out16, int reset)
{
    PCFIT_ENTER();
    PCFIT_HIP(hipDeviceSynchronize());
    PCFIT_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(pcfit::g_qhull_reason), 16 * sizeof(int64_t)));
    if (reset) {
        int64_t zero[16] = {0};
        PCFIT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pcfit::g_qhull_reason), zero, sizeof(zero)));
    }
    return PCFIT_OK;
}

#ifdef PCFIT_PHASE_CLOCK
// tools only: clocks accumulated per phase by thread 0 of one workgroup in 16 (8 counters)
PCFIT_API int pcfit_debug_qhull_phase_clocks(int64_t *out8, int reset)
{
    PCFIT_ENTER();
    PCFIT_HIP(hipDeviceSynchronize());
    PCFIT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(pcfit::g_qphase), 8 * sizeof(int64_t)));
    if (reset) {
        int64_t zero[8] = {0};
        PCFIT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pcfit::g_qphase), zero, sizeof(zero)));
    }
    return PCFIT_OK;
}
#endif
