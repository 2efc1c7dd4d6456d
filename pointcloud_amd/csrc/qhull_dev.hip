// The replay of Qhull's 2-D construction (see qhull_order.hip for what is replayed and why) on the device: one
// workgroup per slice of distinct points.  The facet list, with Qhull's list moves, is kept by one thread in
// LDS; the work that grows with the slice - the partition of the points into outside sets and of the outside
// sets of visible facets among the two new facets, and the search of every set's farthest point - runs
// across the workgroup.  The order of the points inside an outside set is not kept (it decides nothing but ties).
// The kernel follows the host replay decision for decision.  Points that lie below both new facets of an
// insertion by less than a roundoff unit go through Qhull's horizon search (whatever mode Qhull's search is in by
// then, it ends in that walk): their wave walks it during the parallel pass, and the few that end outside of a
// facet are listed and placed by the one thread afterwards.  In the first partition the points arrive in the order
// of the sorted distinct array, so of two points that tie for the farthest of a set the lexicographically first is
// Qhull's.  Where the order inside a later set would matter - two points tie for the farthest of a set, a point lies
// 4 to 8 roundoff units above the first new facet, two facets get their first point through the horizon search in
// one insertion - or one of the fixed capacities below runs out, or the replay ends with more vertices than
// the exact hull has, the slice is handed back (start = QHULL_DEFER) and the host replay decides it.
#include <stdlib.h>

#include <atomic>

#include "pcfit_internal.h"

namespace pcfit {
// Slices handed back, by cause, for tools and tests.  The kernel counts only when it is given the array's address,
// and the launcher gives it only once pcfit_debug_qhull_reasons has been called in the process: the library's own
// calls leave no trace in device globals.
__device__ unsigned long long g_qhull_reason[16];
static std::atomic<bool> g_qhull_counting{false};
__device__ __forceinline__ void count_reason(unsigned long long *why, int k, unsigned long long by = 1ull)
{
    if (why) atomicAdd(&why[k], by);
}
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

constexpr int QF_MAX = 480;      // live facets of a slice (slot 0 is the list's tail): the kernel's roomiest form
constexpr int QP = 4;            // pieces an outside set may lie in
constexpr int QV = 6;            // visible facets of one insertion
constexpr int QU = 16;           // points of one insertion that go through the horizon search
constexpr int64_t QHULL_DEFER = QHULL_START_DEFER;
constexpr unsigned IDX_MASK = 0x7fffffffu;      // a piece's start; the top bit names the buffer it lies in

// The farthest point of an outside set as a thread, a wave or the workgroup has seen it: distance and coordinates
// (has = 0: no point yet), and whether that distance was met at two different points.
struct Far {
    double d, x, y;
    int has, tie;
    __device__ __forceinline__ void take(double od, double ox, double oy)
    {
        if (!has || od > d) { d = od; x = ox; y = oy; has = 1; tie = 0; }
        else if (od == d && (ox != x || oy != y)) tie = 1;
    }
    __device__ __forceinline__ void join(const Far &o)
    {
        if (!o.has) return;
        if (!has || o.d > d) *this = o;
        else if (o.d == d && (o.tie || o.x != x || o.y != y)) tie = 1;      // (o may hold a copy of this point and know of a tie)
    }
    // The same where the points arrive in the order of the sorted distinct array, whatever order they are read in
    // (Qhull's first partition): of two points at the greatest distance the lexicographically first is the one
    // Qhull met first, and it keeps the first.
    __device__ __forceinline__ void take_first(double od, double ox, double oy)
    {
        if (!has || od > d || (od == d && (ox < x || (ox == x && oy < y)))) { d = od; x = ox; y = oy; has = 1; tie = 0; }
    }
    __device__ __forceinline__ void join_first(const Far &o)
    {
        if (o.has) take_first(o.d, o.x, o.y);
    }
};
template <int QF, int QT>
struct Shared {
    static constexpr int FACETS = QF, THREADS = QT;
    double n0[QF], n1[QF], off[QF];
    double far_d[QF], far_x[QF], far_y[QF];      // the farthest point of the outside set (total > 0)
    double vx0[QF], vy0[QF], vx1[QF], vy1[QF];   // the facet's two vertices
    int total[QF];
    int plo[QF][QP], pcnt[QF][QP];
    short nb0[QF], nb1[QF], prev[QF], next[QF], free_slot[QF];
    unsigned char np[QF], top[QF], visible[QF];
    // one insertion
    int c0[QV * QP], c1[QV * QP];
    Far wave_far[2][QT / 64];
    int wave_flag[QT / 64];
    int vis[QV], nvis, newf[2], state, reason, n_free, head, nextf, gap_lo;
    int pt_first[QV * QP + 1], pt_word[QV * QP], pt_slot[QV * QP], n_pt;     // the pieces of one insertion, end to end
    double apex_x, apex_y;
    // points below both new facets by less than a roundoff unit: Qhull searches the horizon for them
    int unc_slot[QU], unc_from[QU], n_unc, gap_used[QV * QP];
    double unc_d[QU], unc_x[QU], unc_y[QU];
    short copl[64];
    unsigned wave_seen[QT / 64][(QF + 31) / 32];     // the visit marks and the stack of a horizon walk, one set per wave
    short wave_copl[QT / 64][16];
    double red_d[QT / 64];
    int red_i[QT / 64];
    double ext_k1[4][QT / 64], ext_k2[4][QT / 64];
    int ext_at[4][QT / 64];
    double thr[6];               // distround, minvisible, minoutside, distoutside, band, (unused)
    double search[3];            // the simplex search: best determinant so far, the one that ends it, near zero
    double sx[3], sy[3];         // the simplex's vertices: least x, greatest x, the third; at the end: the answer
    int found;                   // at the end: the answer's place in the slice
};

// A value that every lane holds alike (read from LDS, which the compiler cannot know), moved to scalar registers.
__device__ __forceinline__ double uni(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <class SH>
__device__ inline double plane_dist(const SH &S, int f, double x, double y) { return S.off[f] + x * S.n0[f] + y * S.n1[f]; }

template <class SH>
__device__ inline void set_plane(SH &S, int f)
{
    const double ax = S.vx0[f], ay = S.vy0[f], bx = S.vx1[f], by = S.vy1[f];
    const double nx = by - ay, ny = ax - bx;
    double norm = sqrt(nx * nx + ny * ny);
    if (!S.top[f]) norm = -norm;
    S.n0[f] = nx / norm;
    S.n1[f] = ny / norm;
    S.off[f] = -(ax * S.n0[f] + ay * S.n1[f]);
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
    constexpr int QT = SH::THREADS;
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

// The farthest point of a wave's 64: the greatest distance by a fold of the distances alone, then the lanes that
// hold it are asked for their point (one of them, and whether another one differs from it).  Every lane gets it.
__device__ __forceinline__ double lane_value(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <bool FIRST = false>
__device__ __forceinline__ void far_fold(Far &f)
{
    const double top = wave_max(f.has ? f.d : -1.0);            // (distances of outside points are positive)
    const unsigned long long best = __builtin_amdgcn_ballot_w64(f.has && f.d == top);
    if (best == 0ull) {
        f.has = 0;
        return;
    }
    if constexpr (FIRST) {
        // the lexicographically first of the points at that distance
        const bool holds = f.has && f.d == top;
        const double big = 1.7976931348623157e308;
        const double bx = -wave_max(holds ? -f.x : -big);
        const double by = -wave_max(holds && f.x == bx ? -f.y : -big);
        f.d = top;
        f.x = bx;
        f.y = by;
        f.has = 1;
        f.tie = 0;
        return;
    }
    const int lane = (int)__builtin_ctzll(best);
    const double bx = lane_value(f.x, lane), by = lane_value(f.y, lane);
    const bool differs = f.has && f.d == top && (f.tie || f.x != bx || f.y != by);
    f.d = top;
    f.x = bx;
    f.y = by;
    f.has = 1;
    f.tie = __builtin_amdgcn_ballot_w64(differs) != 0ull;
}
// The farthest points of two sets over what every thread has seen, to every thread; `flag` comes back as the "or"
// of every thread's.  One barrier: the partials of the last call were read before the barrier that every caller
// passes between two calls.
template <bool FIRST = false, class SH>
__device__ void block_far(SH &S, Far &f0, Far &f1, bool &flag)
{
    constexpr int QT = SH::THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    far_fold<FIRST>(f0);
    far_fold<FIRST>(f1);
    const bool wave_flag = __builtin_amdgcn_ballot_w64(flag) != 0ull;
    if (lane == 0) {
        S.wave_far[0][wave] = f0;
        S.wave_far[1][wave] = f1;
        S.wave_flag[wave] = wave_flag ? 1 : 0;
    }
    __syncthreads();
    flag = false;
    f0.has = f1.has = 0;
    for (int w = 0; w < QT / 64; w++) {
        flag = flag || S.wave_flag[w] != 0;
        if constexpr (FIRST) {
            f0.join_first(S.wave_far[0][w]);
            f1.join_first(S.wave_far[1][w]);
        } else {
            f0.join(S.wave_far[0][w]);
            f1.join(S.wave_far[1][w]);
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

// qh_findbesthorizon for one point, by a whole wave at once (every lane holds the same point and does the same
// steps; the marks and the stack are the wave's own): the greatest distance the walk meets, from facet `from` at
// distance `bestdist` on.  The facet structure is only read.  False when the walk's stack ran out.
template <class SH>
__device__ __attribute__((noinline)) bool wave_horizon(SH &S, int wave, int from, double px, double py, double searchdist, double &bestdist)
{
    constexpr int WORDS = (SH::FACETS + 31) / 32;
    unsigned *seen = S.wave_seen[wave];
    short *copl = S.wave_copl[wave];
    for (int k = threadIdx.x & 63; k < WORDS; k += 64) seen[k] = 0u;
    double minsearch = bestdist - searchdist;
    int facet = from, nextfacet = 0, size = 0;
    seen[from >> 5] |= 1u << (from & 31);
    while (true) {
        for (int j = 0; j < 2; j++) {
            const int nb = j ? S.nb1[facet] : S.nb0[facet];
            if (((seen[nb >> 5] >> (nb & 31)) & 1u) || S.visible[nb]) continue;
            seen[nb >> 5] |= 1u << (nb & 31);
            const double d = plane_dist(S, nb, px, py);
            if (d > bestdist) {
                minsearch = d - searchdist;
                if (d > bestdist + searchdist) size = 0;
                bestdist = d;
            } else if (d < minsearch)
                continue;
            if (nextfacet) {
                if (size == 16) return false;
                copl[size++] = (short)nextfacet;
            }
            nextfacet = nb;
        }
        facet = nextfacet;
        if (facet) nextfacet = 0;
        else if (size == 0) break;
        else facet = copl[--size];
    }
    return true;
}

// QF: facet slots (the LDS a workgroup takes grows with them, and the workgroups a CU holds shrink: every phase of
// this kernel waits - for memory, for the one thread that keeps the list - so it is the number of slices in
// flight that fills the chip); WG: workgroups to a CU the registers are budgeted for.
// QT: threads of the workgroup - 256 for few long slices, one wave when the slices are many: a slice's construction
// is a chain of short phases, so what fills the chip is the number of slices in flight.
template <int QF, int WG, int QT>
__global__ __launch_bounds__(QT, WG) void qhull_order_kernel(const double *__restrict__ ux, const double *__restrict__ uy,
                                                         const int64_t *__restrict__ offsets,
                                                         const int64_t *__restrict__ n_unique,
                                                         const int64_t *__restrict__ n_hull,
                                                         const int64_t *__restrict__ select, double2 *__restrict__ buf_a,
                                                         double2 *__restrict__ buf_b, int64_t *__restrict__ start,
                                                         double *__restrict__ start_xy, int want_index,
                                                         const double *__restrict__ hull_x,
                                                         const double *__restrict__ hull_y,
                                                         const int64_t *__restrict__ hull_idx,
                                                         unsigned long long *__restrict__ why,
                                                         double2 *__restrict__ verts, int32_t *__restrict__ vcount,
                                                         int vstride)
{
    __shared__ Shared<QF, QT> S;
    const int64_t s = blockIdx.x, off = offsets[s];
    const int tid = threadIdx.x;
    // The thread that keeps the facet list: the first of one of the workgroup's waves, another one from slice to
    // slice, so that the list threads of the workgroups that share a compute unit do not all sit on one SIMD.
    const int lt = 64 * (int)(blockIdx.x % (QT / 64));
    if (select && select[s] != PCFIT_HULL_START_QHULL) {
        if (tid == lt) start[s] = -1;
        return;
    }
    const int64_t m64 = n_unique ? n_unique[s] : offsets[s + 1] - off;
    if (m64 < 3 || n_hull[s] < 3) {
        if (tid == lt) start[s] = -1;
        return;
    }
    if (m64 >= (1ll << 30) || n_hull[s] > QF - 2) {                // more facets at the end than there are slots
        if (tid == lt) { start[s] = QHULL_DEFER; count_reason(why, 8); }
        return;
    }
    const int m = (int)m64;
    const double *x = ux + off, *y = uy + off;
    // The outside sets are kept as the points themselves, (x, y) pairs in runs ("pieces") of two buffers: an
    // insertion reads the pieces of the visible facets front to back and writes the points that stay outside to
    // the same entries of the other buffer - no index lists, no gathers.  The first buffer has a slice's m
    // entries, the second 2 m (what the upper half is for: see the first partition).
    double2 *const bufs0 = buf_a + off, *const bufs1 = buf_b + 2 * off;

    // (the free list: idle until the simplex is made, several barriers from here)
    for (int k = tid; k < QF; k += QT) {
        S.free_slot[k] = (short)(QF - 1 - k);       // slots QF-1 .. 1 (0 is the tail)
    }
    // qh_maxmin, qh_detroundoff, qh_maxsimplex.  The four coordinate extremes - least x, greatest x, least y, greatest
    // y, each the lexicographically first point among equals - by their coordinates: they are vertices of the hull,
    // so when the caller has the hull they are found among its few vertices instead of by a pass over the slice.
    QPhase ph;
    ph.init();
    double ex[4], ey[4];
    if (hull_x) {
        const int h = (int)n_hull[s];
        const double *hx = hull_x + off, *hy = hull_y + off;
        const int64_t *hi = hull_idx + off;
        Key e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = {0.0, 0.0, -1};
        for (int i = tid; i < h; i += QT) {
            const int v = (int)hi[i];
            const double px = hx[v], py = hy[v];
            key_take(e[0], px, py, v);
            key_take(e[1], -px, py, v);
            key_take(e[2], py, px, v);
            key_take(e[3], -py, px, v);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) key_fold<32>(e[k], tid & 63);
        if ((tid & 63) == 0)
#pragma unroll
            for (int k = 0; k < 4; k++) { S.ext_k1[k][tid >> 6] = e[k].k1; S.ext_k2[k][tid >> 6] = e[k].k2; S.ext_at[k][tid >> 6] = e[k].at; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            Key r = {0.0, 0.0, -1};
            for (int w = 0; w < QT / 64; w++) key_take(r, S.ext_k1[k][w], S.ext_k2[k][w], S.ext_at[k][w]);
            ex[k] = hx[r.at];
            ey[k] = hy[r.at];
        }
    } else {
        int ext4[4];
        block_extremes(S, x, y, m, ext4);
#pragma unroll
        for (int k = 0; k < 4; k++) { ex[k] = x[ext4[k]]; ey[k] = y[ext4[k]]; }
    }
    ph.mark(0);
    auto same_xy = [](double ax, double ay, double bx, double by) { return ax == bx && ay == by; };
    if (tid == lt) {
        S.state = 0;
        S.reason = 0;
        const double ax = fmax(fabs(ex[0]), fabs(ex[1])), ay = fmax(fabs(ey[2]), fabs(ey[3]));
        const double maxabs = fmax(ax, ay), sumabs = ax + ay;
        const double maxwidth = fmax(ex[1] - ex[0], ey[3] - ey[2]);
        const double eps = 2.220446049250313e-16;
        const double distround = eps * (2 * fmin(sqrt(2.0) * maxabs, sumabs) * 1.01 + maxabs);
        S.thr[0] = distround;
        S.thr[1] = 2 * distround;                  // minvisible
        S.thr[2] = 4 * distround;                  // minoutside
        S.thr[3] = 8 * distround;                  // distoutside
        S.thr[4] = 64 * distround + 8 * distround; // every threshold lies inside this band about zero
        int minx = 0, maxx = 0;
        for (int k = 0; k < 4; k++) {
            if (ex[k] > ex[maxx]) maxx = k;
            if (ex[k] < ex[minx]) minx = k;
        }
        int third = -1;
        double bestdet = -1.0;
        if (ex[minx] != ex[maxx])
            for (int k = 0; k < 4; k++) {
                if (same_xy(ex[k], ey[k], ex[minx], ey[minx]) || same_xy(ex[k], ey[k], ex[maxx], ey[maxx])) continue;
                const double det = fabs((ex[minx] - ex[k]) * (ey[maxx] - ey[k]) - (ex[maxx] - ex[k]) * (ey[minx] - ey[k]));
                if (det > bestdet) { bestdet = det; third = k; }
            }
        const double target = (ex[maxx] - ex[minx]) * maxwidth, nearzero = 10 * (80 * sumabs * eps);
        // the simplex by the coordinates of its vertices: 0 least x, 1 greatest x, 2 the third
        S.sx[0] = ex[minx]; S.sy[0] = ey[minx];
        S.sx[1] = ex[maxx]; S.sy[1] = ey[maxx];
        S.sx[2] = third >= 0 ? ex[third] : 0.0; S.sy[2] = third >= 0 ? ey[third] : 0.0;
        S.search[0] = bestdet; S.search[1] = 10 * 1e-3 * target; S.search[2] = nearzero;
        S.nvis = 0;
        if (ex[minx] == ex[maxx])
            { S.state = 2; S.reason = 1; }          // all points on one vertical line: no hull
        else if (third < 0 || bestdet < nearzero || bestdet / target < 1e-3)
            S.nvis = 1;                             // a flat simplex: Qhull looks through all other points
    }
    __syncthreads();
    if (S.state == 2) {
        if (tid == lt) { start[s] = QHULL_DEFER; count_reason(why, S.reason & 15); }
        return;
    }
    if (S.nvis == 1) {
        // qh_maxsimplex's search, in the points' order: it stops at the first point that spans a hundredth of what
        // the extents allow (and is not near zero); if none does, the largest determinant wins, the first of equals
        const double mindet = S.search[1], nearzero = S.search[2];
        const double ax = S.sx[0], ay = S.sy[0], bx = S.sx[1], by = S.sy[1];
        int first = -1, at = -1;
        double best = -1.0;
        auto larger = [&](double ob, int oa, double b, int a) {
            return oa >= 0 && (a < 0 || ob > b || (ob == b && lex_before(x, y, oa, a)));
        };
        for (int i = tid; i < m; i += QT) {
            const double px = x[i], py = y[i];
            if (same_xy(px, py, ex[0], ey[0]) || same_xy(px, py, ex[1], ey[1]) || same_xy(px, py, ex[2], ey[2]) ||
                same_xy(px, py, ex[3], ey[3]))
                continue;
            const double det = fabs((ax - px) * (by - py) - (bx - px) * (ay - py));
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
        if (tid == lt) {
            for (int w = 0; w < QT / 64; w++) {         // (its own wave's among them: taken twice, no harm)
                const int of = S.c0[w];
                if (of >= 0 && (first < 0 || lex_before(x, y, of, first))) first = of;
                if (larger(S.red_d[w], S.red_i[w], best, at)) { best = S.red_d[w]; at = S.red_i[w]; }
            }
            bool have = S.search[0] >= 0.0;         // (a third vertex among the extremes)
            if (first >= 0) { S.sx[2] = x[first]; S.sy[2] = y[first]; have = true; }
            else if (at >= 0 && best > S.search[0]) { S.sx[2] = x[at]; S.sy[2] = y[at]; S.search[0] = best; have = true; }
            if (first < 0 && (!have || !(S.search[0] > 0.0))) { S.state = 2; S.reason = 1; }
        }
        __syncthreads();
        if (S.state == 2) {
            if (tid == lt) { start[s] = QHULL_DEFER; count_reason(why, S.reason & 15); }
            return;
        }
    }
    if (tid == lt) {
        // qh_createsimplex, qh_initialhull
        S.n_free = QF - 1;
        S.head = S.nextf = 0;
        S.prev[0] = -1;
        S.next[0] = 0;
        const int vs[3] = {2, 1, 0};                // third, greatest x, least x
        int fs[3];
        bool top = true;
        for (int i = 0; i < 3; i++) {
            const int f = S.free_slot[--S.n_free + 0];
            fs[i] = f;
            S.vx0[f] = S.sx[vs[i == 0 ? 1 : 0]]; S.vy0[f] = S.sy[vs[i == 0 ? 1 : 0]];
            S.vx1[f] = S.sx[vs[i == 2 ? 1 : 2]]; S.vy1[f] = S.sy[vs[i == 2 ? 1 : 2]];
            S.top[f] = top;
            S.visible[f] = 0;
            S.np[f] = 0;
            S.total[f] = 0;
            list_append(S, f);
            top = !top;
        }
        S.nb0[fs[0]] = (short)fs[1]; S.nb1[fs[0]] = (short)fs[2];
        S.nb0[fs[1]] = (short)fs[0]; S.nb1[fs[1]] = (short)fs[2];
        S.nb0[fs[2]] = (short)fs[0]; S.nb1[fs[2]] = (short)fs[1];
        const double cx = (S.sx[0] + S.sx[1] + S.sx[2]) / 3.0, cy = (S.sy[0] + S.sy[1] + S.sy[2]) / 3.0;
        set_plane(S, fs[0]);
        if (plane_dist(S, fs[0], cx, cy) > 0)
            for (int i = 0; i < 3; i++) S.top[fs[i]] = !S.top[fs[i]];
        for (int i = 0; i < 3; i++) {
            set_plane(S, fs[i]);
            if (!(plane_dist(S, fs[i], cx, cy) < -S.thr[4])) { S.state = 2; S.reason = 2; }
        }
        S.vis[0] = fs[0]; S.vis[1] = fs[1]; S.vis[2] = fs[2];
        S.c0[0] = S.c0[1] = S.c0[2] = 0;
    }
    __syncthreads();
    if (S.state == 2) {
        if (tid == lt) { start[s] = QHULL_DEFER; count_reason(why, S.reason & 15); }
        return;
    }
    ph.mark(1);
    // qh_partitionall in one pass: every point to the first of the three facets it lies outside of, and into that
    // facet's run at once - the first facet's from the start of the first buffer upwards, the second's from entry
    // m - 1 downwards, the third's from entry m of the second buffer upwards: no two sets may share entries of
    // [0, m), whichever buffer they are in, and where the first two will end is not known before the end of the
    // pass.  Once it is, the third set's first move is directed into the gap between them (gap_lo).  With the
    // farthest point of each set.
    {
        const int f0 = S.vis[0], f1 = S.vis[1], f2 = S.vis[2];
        const double thr = uni(S.thr[3]), minoutside = uni(S.thr[2]);
        Far far[3];
#pragma unroll
        for (int k = 0; k < 3; k++) far[k] = {0.0, 0.0, 0.0, 0, 0};
        const double pl[3][3] = {{uni(S.off[f0]), uni(S.n0[f0]), uni(S.n1[f0])}, {uni(S.off[f1]), uni(S.n0[f1]), uni(S.n1[f1])},
                                 {uni(S.off[f2]), uni(S.n0[f2]), uni(S.n1[f2])}};
        // (the loads of the next turn are in flight while this one is classified)
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
            int cls[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + u * QT + tid;
                int c = 3;
                double d = 0.0;
                // (the simplex's own vertices need no test: they lie on two of the facets to within the roundoff
                // of a distance, a quarter of the lower threshold, and below the third)
                if (i < m) {
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
                cls[u] = c;
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (c == k) far[k].take_first(d, px[u], py[u]);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bool mine[4];
                int slot[4];
#pragma unroll
                for (int u = 0; u < 4; u++) mine[u] = cls[u] == k;
                wave_slots<4>(&S.c0[k], mine, slot);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (mine[u]) {
                        const double2 xy = make_double2(px[u], py[u]);
                        if (k == 0) bufs0[slot[u]] = xy;
                        else if (k == 1) bufs0[last - slot[u]] = xy;
                        else bufs1[m + slot[u]] = xy;
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
        // the farthest point of each of the three sets
        Far unused = {0.0, 0.0, 0.0, 0, 0};
        bool none = false;
        block_far<true>(S, far[0], far[1], none);
        __syncthreads();                           // (the partials are read: they may be written again)
        block_far<true>(S, far[2], unused, none);
        if (tid == lt) {
            const int cnt[3] = {S.c0[0], S.c0[1], S.c0[2]};
            const int fs[3] = {f0, f1, f2};
            const unsigned at[3] = {0u, (unsigned)(m - cnt[1]), 0x80000000u | (unsigned)m};
            S.gap_lo = cnt[0];
            for (int c = 0; c < 3; c++) {
                S.total[fs[c]] = cnt[c];
                S.far_x[fs[c]] = far[c].x;
                S.far_y[fs[c]] = far[c].y;
                S.far_d[fs[c]] = far[c].d;
                if (cnt[c] > 0) { S.np[fs[c]] = 1; S.plo[fs[c]][0] = (int)at[c]; S.pcnt[fs[c]][0] = cnt[c]; }
            }
            // qh_furthestnext: the facet with the farthest outside point goes to the front of the list
            int bestf = 0;
            double bestd = -1e300;
            for (int c = 0; c < 3; c++)
                if (cnt[c] > 0 && far[c].d > bestd) { bestd = far[c].d; bestf = fs[c]; }
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
            if (tid == lt) { start[s] = QHULL_DEFER; count_reason(why, S.reason & 15); }
            return;
        }
    }
    ph.mark(2);
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
                const int facet = S.nextf;
                const double ax = S.far_x[facet], ay = S.far_y[facet];
                S.apex_x = ax;
                S.apex_y = ay;
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
                        S.vx0[nf] = ax;
                        S.vy0[nf] = ay;
                        S.vx1[nf] = hskip ? S.vx0[g] : S.vx1[g];
                        S.vy1[nf] = hskip ? S.vy0[g] : S.vy1[g];
                        S.top[nf] = S.top[g] ? (hskip & 1) != 0 : (hskip & 1) == 0;
                        S.visible[nf] = 0;
                        S.np[nf] = 0;
                        S.total[nf] = 0;
                        S.nb0[nf] = (short)g;
                        if (hskip) S.nb1[g] = (short)nf; else S.nb0[g] = (short)nf;
                        set_plane(S, nf);
                        list_append(S, nf);
                        S.newf[n_new++] = nf;
                    }
                }
                if (S.state == 0 && n_new != 2) { S.state = 2; S.reason = 7; }
                if (S.state == 0) {
                    S.nb1[S.newf[0]] = (short)S.newf[1];
                    S.nb1[S.newf[1]] = (short)S.newf[0];
                    // the pieces of the visible facets, end to end: the workgroup goes through them as one run
                    int n_pt = 0, run = 0;
                    for (int k = 0; k < nvis; k++)
                        for (int q = 0; q < S.np[S.vis[k]]; q++) {
                            const int slot = k * QP + q;
                            S.c0[slot] = S.c1[slot] = S.gap_used[slot] = 0;
                            S.pt_first[n_pt] = run;
                            S.pt_word[n_pt] = S.plo[S.vis[k]][q];
                            S.pt_slot[n_pt++] = slot;
                            run += S.pcnt[S.vis[k]][q];
                        }
                    S.pt_first[n_pt] = run;
                    S.n_pt = n_pt;
                    S.n_unc = 0;
                }
            }
            };
    if (tid == lt) next_insertion();
    __syncthreads();
    ph.mark(4);
    for (int step = 0; step < m + 8 && S.state == 0; step++) {
        // qh_partitionvisible: the outside points of the visible facets to the first new facet they lie outside of
        const int nvis = S.nvis, a0 = S.newf[0], a1 = S.newf[1], gap_lo = S.gap_lo;
        const double apex_x = uni(S.apex_x), apex_y = uni(S.apex_y);
        const double minoutside = uni(S.thr[2]), reach = uni(S.thr[3]), distround = uni(S.thr[0]);
        const double searchdist = distround * 2 + uni(S.thr[1]);
        const double a0_n0 = uni(S.n0[a0]), a0_n1 = uni(S.n1[a0]), a0_off = uni(S.off[a0]);
        const double a1_n0 = uni(S.n0[a1]), a1_n1 = uni(S.n1[a1]), a1_off = uni(S.off[a1]);
        Far far0 = {0.0, 0.0, 0.0, 0, 0}, far1 = far0;
        bool doubt = false;
        // The pieces of the visible facets as one run of n_run points, four points a thread and turn (past the end:
        // the apex, which goes nowhere), so that the loads of a whole insertion - some hundred points in a handful
        // of pieces, as a rule - are in flight together.
        const int n_pt = S.n_pt, n_run = S.pt_first[n_pt];
        for (int base = 0; base < n_run; base += QT * 4) {
            double px[4], py[4];
            int piece[4], to_at[4], to_end[4];          // the point's piece, where that piece goes: first and last entry
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = base + u * QT + tid, jj = j < n_run ? j : n_run - 1;
                int pi = 0;
                while (jj >= S.pt_first[pi + 1]) pi++;
                const unsigned word = (unsigned)S.pt_word[pi];
                const int first = S.pt_first[pi], cnt = S.pt_first[pi + 1] - first;
                const int lo = (int)(word & IDX_MASK), from = (int)(word >> 31);
                // (the one piece in the second buffer's upper half - the third set of the first partition - moves
                // into the gap the other two left; the top bit of to_at: the buffer it goes to is the first one)
                const int to_lo = lo >= m ? gap_lo : lo;
                const double2 xy = (from ? bufs1 : bufs0)[lo + (jj - first)];
                px[u] = j < n_run ? xy.x : apex_x;
                py[u] = j < n_run ? xy.y : apex_y;
                piece[u] = pi;
                to_at[u] = to_lo | (from << 31);
                to_end[u] = to_lo + cnt - 1;
            }
            int cls[4], near_bits = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int c = 2;
                double d = 0.0;
                bool near = false;
                if (px[u] != apex_x || py[u] != apex_y) {
                    d = a0_off + px[u] * a0_n0 + py[u] * a0_n1;
                    if (d >= minoutside) {
                        c = 0;
                        if (d < reach) doubt = true;        // (Qhull's other search mode may prefer the second facet)
                    } else {
                        const double d0 = d;
                        d = a1_off + px[u] * a1_n0 + py[u] * a1_n1;
                        if (d >= minoutside) c = 1;
                        else if (fmax(d0, d) >= -distround)
                            near = true;            // below both by less than a roundoff unit: the horizon search
                    }
                }
                cls[u] = c;
                if (c == 0) far0.take(d, px[u], py[u]);
                else if (c == 1) far1.take(d, px[u], py[u]);
                if (near) near_bits |= 1 << u;
            }
            // the points of the first new facet upwards from their piece's start, those of the second downwards from
            // its end; a wave's points of one turn lie in consecutive pieces (one, as a rule)
            const int p_first = __builtin_amdgcn_readfirstlane(piece[0]), p_last = __builtin_amdgcn_readlane(piece[3], 63);
            for (int p = p_first; p <= p_last; p++) {
                const int slot = S.pt_slot[p];
                bool m0[4], m1[4];
                int s0[4], s1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { m0[u] = cls[u] == 0 && piece[u] == p; m1[u] = cls[u] == 1 && piece[u] == p; }
                wave_slots<4>(&S.c0[slot], m0, s0);
                wave_slots<4>(&S.c1[slot], m1, s1);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    double2 *dst = to_at[u] < 0 ? bufs0 : bufs1;
                    if (m0[u]) dst[(to_at[u] & (int)IDX_MASK) + s0[u]] = make_double2(px[u], py[u]);
                    if (m1[u]) dst[to_end[u] - s1[u]] = make_double2(px[u], py[u]);
                }
            }
            // Points below both new facets by less than a roundoff unit are many where coordinates were rounded (points
            // in line with the new facets) and nearly all of them end outside of nothing.  The wave walks the horizon
            // for each, one after the other, and only those that did get outside of a facet are listed for the
            // thread that keeps the list, which walks again and places them after the pass.
            if (__builtin_amdgcn_ballot_w64(near_bits != 0) != 0ull) {
                for (int u = 0; u < 4; u++) {
                    unsigned long long todo = __builtin_amdgcn_ballot_w64((near_bits >> u) & 1);
                    while (todo) {
                        const int src = (int)__builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        const double wx = lane_value(px[u], src), wy = lane_value(py[u], src);
                        const double d0 = a0_off + wx * a0_n0 + wy * a0_n1, d1 = a1_off + wx * a1_n0 + wy * a1_n1;
                        const int from = d1 > d0 ? a1 : a0;
                        double best = fmax(d0, d1);
                        const bool walked = wave_horizon(S, tid >> 6, from, wx, wy, searchdist, best);
                        if ((tid & 63) == src && (!walked || best >= minoutside)) {
                            const int at = atomicAdd(&S.n_unc, 1);
                            if (at < QU) {
                                S.unc_x[at] = wx;
                                S.unc_y[at] = wy;
                                S.unc_slot[at] = S.pt_slot[piece[u]];
                                S.unc_from[at] = from;
                                S.unc_d[at] = fmax(d0, d1);
                            }
                        }
                    }
                }
            }
        }
        ph.mark(5);
        block_far(S, far0, far1, doubt);
        ph.mark(6);
        if (tid == lt) {
            if (doubt) { S.state = 2; S.reason = 10; }
            if (S.state == 0) {
            if ((far0.has && far0.tie) || (far1.has && far1.tie)) { S.state = 2; S.reason = 4; }
            const int child[2] = {a0, a1};
            for (int k = 0; k < nvis && S.state == 0; k++) {
                const int v = S.vis[k], np = S.np[v];
                for (int q = 0; q < np; q++) {
                    const unsigned word = (unsigned)S.plo[v][q];
                    const int from_lo = (int)(word & IDX_MASK), lo = from_lo >= m ? gap_lo : from_lo;
                    const int to = (int)(word >> 31) ^ 1, cnt = S.pcnt[v][q], slot = k * QP + q;
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
            S.far_x[a0] = far0.x; S.far_y[a0] = far0.y; S.far_d[a0] = far0.d;
            S.far_x[a1] = far1.x; S.far_y[a1] = far1.y; S.far_d[a1] = far1.d;
            // qh_findbesthorizon for the points that lie below both new facets by less than a roundoff unit
            // (whatever mode Qhull's search is in by then, it ends in this walk): over neighbours whose distance
            // is no more than the search distance below the best one found; the point joins the facet found if it
            // lies outside of it, and a facet that had no outside points goes to the end of the list
            if (S.n_unc > QU) { S.state = 2; S.reason = 10; }
            int moved = 0;
            const double searchdist = S.thr[0] * 2 + S.thr[1];
            for (int k = 0; k < S.n_unc && S.state == 0; k++) {
                const double px = S.unc_x[k], py = S.unc_y[k];
                double bestdist = S.unc_d[k], minsearch = bestdist - searchdist;
                int bestf = S.unc_from[k], facet = bestf, nextfacet = 0, size = 0;
                unsigned *seen = S.wave_seen[lt >> 6];       // (this wave's own marks: no wave walks by now)
                for (int w = 0; w < (QF + 31) / 32; w++) seen[w] = 0u;
                seen[bestf >> 5] |= 1u << (bestf & 31);
                while (true) {
                    for (int j = 0; j < 2; j++) {
                        const int nb = j ? S.nb1[facet] : S.nb0[facet];
                        if (((seen[nb >> 5] >> (nb & 31)) & 1u) || S.visible[nb]) continue;
                        seen[nb >> 5] |= 1u << (nb & 31);
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
                    S.far_x[bestf] = px; S.far_y[bestf] = py;
                    S.far_d[bestf] = bestdist;
                } else if (S.far_d[bestf] < bestdist) {
                    S.far_x[bestf] = px; S.far_y[bestf] = py;
                    S.far_d[bestf] = bestdist;
                }
                if (S.np[bestf] == QP) { S.state = 2; S.reason = 11; break; }
                // a one-point piece in the gap the dropped points left in the piece this point came from
                const int slot = S.unc_slot[k], v = S.vis[slot / QP], q = slot % QP;
                const unsigned word = (unsigned)S.plo[v][q];
                const int from_lo = (int)(word & IDX_MASK), lo = from_lo >= m ? gap_lo : from_lo, to = (int)(word >> 31) ^ 1;
                const int at = lo + S.c0[slot] + S.gap_used[slot]++;
                (to ? bufs1 : bufs0)[at] = make_double2(px, py);
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
    if (tid == lt) {
        bool answered = false;
        if (S.state == 1) {
            // scipy's walk from the head of the list; facets in one line are one facet in Qhull's result (it merged
            // them), standing in the list where the newest part stands.  (The free list, the sets' sizes and the
            // visibility flags are idle by now and hold the walk, the places in the list and the joints.)
            short *walk = S.free_slot;
            int *place = S.total;
            unsigned char *joined = S.visible;
            int nw = 0, k = 0;
            for (int f = S.head; nw < QF;) {
                walk[nw++] = (short)f;
                f = S.top[f] ? S.nb0[f] : S.nb1[f];
                if (f == S.head) break;
            }
            for (int g = S.head; g != 0 && k < QF; g = S.next[g]) place[g] = k++;
            const double radius = 2 * S.thr[0];
            int n_joined = 0;
            auto centrum_dist = [&](int of, int to) {          // a facet's centrum, brought onto its own line, from another's line
                double cx = (S.vx0[of] + S.vx1[of]) / 2, cy = (S.vy0[of] + S.vy1[of]) / 2;
                const double own = plane_dist(S, of, cx, cy);
                cx -= own * S.n0[of];
                cy -= own * S.n1[of];
                return plane_dist(S, to, cx, cy);
            };
            for (int i = 0; i < nw; i++) {
                const int b = walk[i], a = walk[(i + nw - 1) % nw];
                joined[i] = fabs(centrum_dist(b, a)) <= radius || fabs(centrum_dist(a, b)) <= radius;
                n_joined += joined[i];
            }
            if (nw - n_joined < 3) {
                for (int i = 0; i < nw; i++) joined[i] = 0;
                n_joined = 0;
            }
            int first = 0;
            while (joined[first]) first++;
            int count = 0, run_first = first, run_place = 0, best_first = first, best_place = 0x7fffffff;
            for (int t = 0, i = first; t <= nw; t++, i = (i + 1) % nw) {
                if (t == nw || !joined[i]) {
                    if (t > 0 && run_place < best_place) { best_place = run_place; best_first = run_first; }
                    if (t == nw) break;
                    count++;
                    run_first = i;
                    run_place = place[walk[i]];
                } else if ((int)place[walk[i]] > run_place)
                    run_place = place[walk[i]];
            }
            // the hull the other kernel found by an exact predicate has these vertices and, where coordinates were
            // rounded, some at the slightest of turns that Qhull does not keep (the turn of the walk takes them out);
            // fewer there than here would mean that something was missed
            if (nw == k && count <= n_hull[s] && (!verts || count <= vstride)) {
                const int h = walk[best_first];
                S.sx[0] = S.top[h] ? S.vx0[h] : S.vx1[h];
                S.sy[0] = S.top[h] ? S.vy0[h] : S.vy1[h];
                answered = true;
                // the vertices Qhull ends with, in its order (one per facet or run of facets in line): which of the
                // exact hull's vertices they are is for the caller to look up - the exact hull keeps a vertex at
                // the slightest turn, Qhull never makes one of a point that no facet ever saw outside
                if (verts) {
                    int n_out = 0;
                    for (int t = 0, i = best_first; t < nw; t++, i = (i + 1) % nw)
                        if (!joined[i]) {
                            const int f = walk[i];
                            verts[(int64_t)s * vstride + n_out++] = make_double2(S.top[f] ? S.vx0[f] : S.vx1[f], S.top[f] ? S.vy0[f] : S.vy1[f]);
                        }
                    vcount[s] = n_out;
                }
            } else
                S.reason = 12;
        } else if (S.state == 0)
            S.reason = 13;
        if (!answered) count_reason(why, S.reason & 15);
        count_reason(why, 14, (unsigned long long)steps);
        count_reason(why, 15, visits);
        S.found = answered ? 0x7fffffff : -1;
    }
    __syncthreads();
    const bool answered = S.found >= 0;
    if (answered && want_index) {
        // which point of the slice that is (the first of its copies): one more pass, for callers that ask
        const double vx = S.sx[0], vy = S.sy[0];
        int at = 0x7fffffff;
        for (int i = tid; i < m; i += QT)
            if (x[i] == vx && y[i] == vy && i < at) at = i;
        if (at != 0x7fffffff) atomicMin(&S.found, at);
        __syncthreads();
    }
    if (tid == lt) {
        // (without the pass the answer's coordinates are the result, and start[s] = 0 says that there is one)
        start[s] = !answered ? QHULL_DEFER : want_index ? (S.found == 0x7fffffff ? QHULL_DEFER : S.found) : 0;
        if (start_xy && answered) { start_xy[2 * s] = S.sx[0]; start_xy[2 * s + 1] = S.sy[0]; }
    }
    ph.mark(7);
}

}  // namespace

// start[s] (device): the first vertex of Qhull's order, -1 where there is no hull or the slice was not selected,
// QHULL_DEFER where the host has to decide; start_xy (nullable): its coordinates.  The points of a slice may
// come sorted and distinct (n_unique of them) or as they are (n_unique null: the whole slice, any order,
// duplicates allowed - the answer is the same point, its index then names one of its copies).
// scratch: 3 * total (x, y) pairs = 48 bytes a point.  want_index = false: start[s] = 0 wherever there is an answer
// (its coordinates are in start_xy), which saves the pass that finds the point's place in its slice.
int qhull_start_dev(const double *ux, const double *uy, const int64_t *offsets, const int64_t *n_unique,
                    const int64_t *n_hull, const int64_t *select, int64_t n_slices, int64_t total, void *scratch,
                    int64_t *start, double *start_xy, hipStream_t st, int64_t max_hull, const double *hull_x,
                    const double *hull_y, const int64_t *hull_idx, bool want_index, void *verts, int32_t *vcount,
                    int vstride)
{
    if (n_slices <= 0) return PCFIT_OK;
    // The form with the fewest facet slots that holds the largest hull of the batch and two dozen facets more: the
    // facets alive at a time are the edges of the polygon built so far, whose vertices are, but for a few that a later
    // point takes out again, vertices of the hull (with 16 more than the hull no slice of the randomised sweeps ran out; one
    // that runs out of slots all the same is handed back).  Fewer slots, more slices in flight on a compute unit.
    // PCFIT_QHULL_FORM (tools and tests; read at every call) names a form: slots + 0 / 1 / 2 / 3 for 256 / 64 / 512 /
    // 128 threads.
    const char *env_form = getenv("PCFIT_QHULL_FORM");
    const int forced = env_form ? atoi(env_form) : 0;
    static const int margin = [] { const char *e = getenv("PCFIT_QHULL_MARGIN"); return e ? atoi(e) : 24; }();
    const int64_t need = max_hull < 0 ? QF_MAX : max_hull + std::max(margin, 2);
    // A slice's construction is a chain of short phases (0.7 ms of dependent latency for a slice of 50 000 points
    // however many threads work on it), so what fills the chip is the number of slices in flight, and the LDS of the
    // facet slots (131 bytes each) is what limits it: 11 / 10 / 8 / 8 / 7 / 5 / 2 slices to a compute unit with 96 /
    // 112 / 128 / 144 / 160 / 224 / 480 slots, and no more than 5 in workgroups of 256 threads or 8 of 128 threads
    // (registers).  Measured on 96 slots, slices of 5000 to 100 000 points: up to 5 slices a compute unit 256
    // threads win (1000 x 100 000: 1.94 ms against 2.09 with 128), up to a dozen 128 threads (2000 x 50 000: 1.77
    // against 2.45 with 256, 1.85 with 64), beyond that one wave a slice (10 000 x 8192: 2.03 against 2.39 and
    // 3.70; 20 000 x 5000: 2.67, 3.49, 5.57).  With more than 160 slots the narrow forms hold no more slices
    // than the wide ones.
    // Few long slices (a scan cut into some dozen windows): a slice's passes are chains of turns of one point a
    // thread, each a trip to memory long, and with no more slices than compute units there is nothing to fill the
    // chip with but threads - workgroups of 512 then.  Measured on 57 to 256 slices of 50 000 to 1 000 000 points:
    // 5 to 40 % faster than with 256 threads (1024 threads gain less: sixteen waves wait for the slowest at every
    // barrier); from 400 slices on the narrow form wins.
    int dev = 0, cus = 256;
    PCFIT_HIP(hipGetDevice(&dev));
    PCFIT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int slots = need <= 96 ? 96 : need <= 112 ? 112 : need <= 128 ? 128 : need <= 144 ? 144 : need <= 160 ? 160 : need <= 224 ? 224 : QF_MAX;
    const bool wide = n_slices <= 256 && total / n_slices >= 32768;
    const int64_t per_cu = (n_slices + cus - 1) / std::max(cus, 1);
    const int threads = wide ? 2 : (slots > 160 || per_cu <= 5) ? 0 : per_cu <= 12 ? 3 : 1;
    if ((threads == 0 || threads == 2) && (slots == 112 || slots == 144)) slots += 16;     // (narrow forms only)
    const int form = forced ? forced : slots + threads;
    const dim3 grid((unsigned)n_slices);
    double2 *const pairs = reinterpret_cast<double2 *>(scratch);
    unsigned long long *why = nullptr;
    if (g_qhull_counting.load()) PCFIT_HIP(hipGetSymbolAddress(reinterpret_cast<void **>(&why), HIP_SYMBOL(g_qhull_reason)));
#define PCFIT_QHULL_LAUNCH(QF, WG, QT)                                                                              \
    qhull_order_kernel<QF, WG, QT><<<grid, dim3(QT), 0, st>>>(ux, uy, offsets, n_unique, n_hull, select, pairs,     \
                                                              pairs + total, start, start_xy, want_index ? 1 : 0, \
                                                              hull_x, hull_y, hull_idx, why,                    \
                                                              reinterpret_cast<double2 *>(verts), vcount, vstride)
    switch (form) {
    case 96: PCFIT_QHULL_LAUNCH(96, 5, 256); break;
    case 97: PCFIT_QHULL_LAUNCH(96, 10, 64); break;
    case 98: PCFIT_QHULL_LAUNCH(96, 2, 512); break;
    case 99: PCFIT_QHULL_LAUNCH(96, 8, 128); break;
    case 113: PCFIT_QHULL_LAUNCH(112, 9, 64); break;
    case 115: PCFIT_QHULL_LAUNCH(112, 8, 128); break;
    case 128: PCFIT_QHULL_LAUNCH(128, 5, 256); break;
    case 129: PCFIT_QHULL_LAUNCH(128, 8, 64); break;
    case 130: PCFIT_QHULL_LAUNCH(128, 2, 512); break;
    case 131: PCFIT_QHULL_LAUNCH(128, 8, 128); break;
    case 145: PCFIT_QHULL_LAUNCH(144, 8, 64); break;
    case 147: PCFIT_QHULL_LAUNCH(144, 8, 128); break;
    case 160: PCFIT_QHULL_LAUNCH(160, 5, 256); break;
    case 161: PCFIT_QHULL_LAUNCH(160, 7, 64); break;
    case 162: PCFIT_QHULL_LAUNCH(160, 2, 512); break;
    case 163: PCFIT_QHULL_LAUNCH(160, 7, 128); break;
    case 224: PCFIT_QHULL_LAUNCH(224, 5, 256); break;
    case 225: PCFIT_QHULL_LAUNCH(224, 5, 64); break;
    case 226: PCFIT_QHULL_LAUNCH(224, 2, 512); break;
    case 227: PCFIT_QHULL_LAUNCH(224, 4, 128); break;
    case QF_MAX + 1: PCFIT_QHULL_LAUNCH(QF_MAX, 3, 64); break;
    case QF_MAX + 2: PCFIT_QHULL_LAUNCH(QF_MAX, 2, 512); break;
    default: PCFIT_QHULL_LAUNCH(QF_MAX, 3, 256); break;
    }
#undef PCFIT_QHULL_LAUNCH
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API int pcfit_qhull_start_batch_dev(const double *ux, const double *uy, const int64_t *offsets,
                                          const int64_t *n_unique, const int64_t *n_hull, const int64_t *select,
                                          int64_t n_slices, int64_t total_points, void *scratch, size_t scratch_bytes,
                                          int64_t *start, const double *hull_x, const double *hull_y,
                                          const int64_t *hull_idx, int64_t max_hull, void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31) && total_points >= 0, "qhull_start: sizes out of range");
    PCFIT_REQUIRE((hull_x != nullptr) == (hull_y != nullptr) && (hull_x != nullptr) == (hull_idx != nullptr),
                  "qhull_start: hull_x, hull_y and hull_idx come together or not at all");
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(ux && uy && offsets && n_hull && start, "qhull_start: null pointer");
    if (!scratch || scratch_bytes < (size_t)total_points * 48 || ((uintptr_t)scratch & 15)) {
        set_error("qhull_start: the scratch buffer holds %zu bytes, %zu needed (aligned to 16)", scratch_bytes,
                  (size_t)total_points * 48);
        return PCFIT_ERR_WORKSPACE;
    }
    return qhull_start_dev(ux, uy, offsets, n_unique, n_hull, select, n_slices, total_points, scratch, start,
                           nullptr, as_stream(stream), max_hull < 0 ? -1 : max_hull, hull_x, hull_y, hull_idx, true);
}

// tools and tests: how many slices the kernel handed back since the last reset, by cause (1 flat simplex, 2 simplex
// orientation, 3 first partition near a threshold, 4 tie for the farthest point, 5 horizon near the threshold,
// 6 visible facets, 7 new facets, 8 facet slots, 9 nearly collinear facets, 10 partition near a threshold,
// 11 pieces of a set, 12 vertex count, 13 steps)
PCFIT_API int pcfit_debug_qhull_reasons(int64_t *out16, int reset)
{
    PCFIT_ENTER();
    pcfit::g_qhull_counting.store(true);
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
