// Distinct points, rank check and convex hull of every slice
// (circle_fitting.py:14-19 and :182-198), one 1024-thread workgroup per slice,
// gfx950.
//
//   sort     the slice's (x, y) pairs lexicographically (what np.unique(axis=0)
//            does).  In LDS, from 96 points on: a counting sort on a bucket
//            number that never decreases with x, every bucket then put in
//            order by one thread (bucket_sort_lds).  Otherwise, and for inputs
//            that do not suit the buckets: an ascending-only bitonic network;
//            slots past the end count as +inf and are never stored;
//   unique   neighbours that compare equal are dropped by a prefix-sum
//            compaction;
//   rank     singular values of the centred cloud as root sums of squares of
//            its projections on the principal axes of the 2x2 scatter matrix;
//   hull     the sorted points below the line from the first to the last one,
//            ascending, then those above it, descending, form a closed walk;
//            rounds in which every candidate at which the walk does not turn
//            left is dropped, all at once, leave the counter-clockwise hull
//            from the lexicographic minimum (hull_by_pruning).  Where those
//            rounds do not end in time, and in global memory: Andrew's monotone
//            chain, cut into chunks: every thread runs the chain on its own run
//            of 16+ consecutive candidates, survivors are compacted, and the
//            rounds repeat (chunks double when a round removes less than half)
//            until one thread finishes the rest.  The orientation test is exact
//            (floating-point filter, then an expansion sum), so the vertex set
//            does not depend on how the rounds or chunks fall and equals the
//            serial chain's.
//
// A slice of up to 8192 points is sorted and hulled inside the 160 KB LDS
// (16 B of coordinates + 2 B of index per point); longer slices run the same
// code on a global-memory workspace - after a filter that drops the points
// which cannot be hull vertices: around a centre c inside the hull (the mean
// of a part of the slice's points) the plane is cut
// into K sectors (128..512 by slice length) and the farthest point of every
// sector is found.  A point of sector j lies between the farthest points A, B
// of the two flanking sectors as seen from c, at most 24/K rad apart, and the
// segment AB stays farther from c than min(|cA|, |cB|) cos(12/K): a point closer
// than that is inside the triangle cAB, hence no hull vertex (details at
// hull_filter_slice_kernel).  The threshold follows the outline sector by
// sector, so it does not matter where inside the outline the centre lies.
// For a noisy ring that leaves a few percent of the points.  The filter runs
// one workgroup per 8192 points; what it keeps is then sorted, made unique and
// hulled as before, so hull vertices and their order are unchanged.  A caller
// that wants the hulls only (hull_only) stops there: n_unique / ux / uy of such
// a slice then describe the kept points.  Otherwise the whole slice is sorted
// and made unique as well (longsort.hip: tile sort in LDS, rank merges in global
// memory), the hull vertices are looked up in that full sorted set by their
// coordinates, and a requested start vertex is honoured.
#include <algorithm>

#include "pcfit_internal.h"
#include "segments.h"

namespace pcfit {

constexpr int HULL_BLOCK = 1024;          // threads of the wide launch (and the most a launch may have)
constexpr int HULL_NARROW = 256;          // threads of the narrow launch for short filtered remainders
#define HULL_NT ((int)blockDim.x)          // threads of this launch, in device code

// Optional phase timing (tools only; -DPCFIT_PHASE_CLOCK): thread 0 of every workgroup adds the clocks
// spent between marks to g_hull_phase[k].
#ifdef PCFIT_PHASE_CLOCK
__device__ unsigned long long g_hull_phase[8];
#define HULL_MARK(k)                                                                   \
    do {                                                                               \
        if (threadIdx.x == 0) {                                                        \
            const long long now_ = clock64();                                          \
            atomicAdd(&g_hull_phase[k], (unsigned long long)(now_ - hull_last_));      \
            hull_last_ = now_;                                                         \
        }                                                                              \
    } while (0)
#define HULL_MARK_INIT() long long hull_last_ = clock64()
#else
#define HULL_MARK(k) do { } while (0)
#define HULL_MARK_INIT() do { } while (0)
#endif
constexpr int HULL_LDS_CAP = 8192;
constexpr int HULL_BRUTE_MAX = 96;            // candidates of a chain from which on every one tests itself (chain_finish)
constexpr int HULL_SMALL_CAP = 2048;          // LDS capacity (points) of the narrow launch
constexpr int64_t HULL_NARROW_MIN_SLICES = 512;   // fewer slices than this do not fill the chip either way

__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e)
{
    s = a + b;
    const double bv = s - a;
    const double av = s - bv;
    e = (a - av) + (b - bv);
}

// exact sign of (bx-ax)(cy-ay) - (by-ay)(cx-ax)
__device__ int orient_sign(double ax, double ay, double bx, double by, double cx, double cy)
{
    const double l = (bx - ax) * (cy - ay), r = (by - ay) * (cx - ax);
    const double det = l - r;
    const double bound = 3.3306690738754716e-16 * (fabs(l) + fabs(r));
    if (det > bound) return 1;
    if (det < -bound) return -1;
    // bx*cy - bx*ay - ax*cy - by*cx + by*ax + ay*cx, every product as hi + lo
    const double pa[6] = {bx, -bx, -ax, -by, by, ay};
    const double pb[6] = {cy, ay, cy, cx, ax, cx};
    double ex[12];
    int len = 0;
    for (int k = 0; k < 6; k++) {
        const double hi = pa[k] * pb[k];
        const double lo = fma(pa[k], pb[k], -hi);
        for (int t = 0; t < 2; t++) {
            double q = t == 0 ? lo : hi;
            for (int i = 0; i < len; i++) {
                double s, e;
                two_sum(q, ex[i], s, e);
                ex[i] = e;
                q = s;
            }
            ex[len++] = q;
        }
    }
    for (int i = len - 1; i >= 0; i--) {
        if (ex[i] > 0) return 1;
        if (ex[i] < 0) return -1;
    }
    return 0;
}

__device__ __forceinline__ bool lex_less(double ax, double ay, double bx, double by)
{
    return ax < bx || (ax == bx && ay < by);
}

// exclusive prefix sum of one int per thread over the workgroup; total in *total
__device__ __forceinline__ int block_exscan(int v, int *scratch /* 17 ints */, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, WAVE);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < nw; w++) {
            const int t = scratch[w];
            scratch[w] = run;
            run += t;
        }
        scratch[16] = run;
    }
    __syncthreads();
    *total = scratch[16];
    return scratch[wave] + inc - v;
}

// One chain round over list[0..L): chunks of c entries, one per thread; the
// survivors of a chunk stay at the front of its own segment.  Returns the
// number this thread kept (0 for threads without a chunk).
template <class IdxT>
__device__ __forceinline__ int chain_chunk(const double *px, const double *py, IdxT *list, int64_t L, int64_t c)
{
    const int64_t begin = (int64_t)threadIdx.x * c;
    if (begin >= L)
        return 0;
    const int64_t end = min(L, begin + c);
    IdxT *st = list + begin;
    int k = 0;
    // the two points on top of the stack stay in registers: a push costs no load, a pop one
    double ax = 0.0, ay = 0.0, bx = 0.0, by = 0.0;          // st[k-2], st[k-1]
    for (int64_t i = begin; i < end; i++) {
        const IdxT v = list[i];
        const double vx = px[v], vy = py[v];
        while (k >= 2 && orient_sign(ax, ay, bx, by, vx, vy) <= 0) {
            k--;
            bx = ax;
            by = ay;
            if (k >= 2) {
                const IdxT u = st[k - 2];
                ax = px[u];
                ay = py[u];
            }
        }
        st[k++] = v;
        ax = bx;
        ay = by;
        bx = vx;
        by = vy;
    }
    return k;
}

// The end of a chain: once a round has left at most HULL_BRUTE_MAX candidates every one of them decides for
// itself, all at once.  Candidate q stays exactly when it lies strictly to the right of the chord from A to B
// (walking the chain: a left turn at q), A being the candidate before q from which the line to q turns least
// to the left of all and B the one after q likewise - any other chord over q passes on the far side of that
// one.  One thread per candidate walks all the others (the candidate looked at is the same for every thread:
// LDS broadcasts).  The test is the chain's own exact orientation test, so the survivors are the chain's.  Rounds of chunks down to one thread's chunk, which this replaces, were a string of barriers and a
// serial tail: about 40 us for the 200 points that the interior filter leaves of a long slice, against 10.
template <class IdxT>
__device__ int chain_finish(const double *px, const double *py, IdxT *list, int L, int *scratch)
{
    __shared__ double fx[HULL_BRUTE_MAX], fy[HULL_BRUTE_MAX];
    const int q = threadIdx.x;
    const bool have = q < L;
    double qx = 0.0, qy = 0.0;
    IdxT me = 0;
    __syncthreads();                                    // (fx, fy may still be read by the previous chain's finish)
    if (have) {
        me = list[q];
        qx = fx[q] = px[me];
        qy = fy[q] = py[me];
    }
    __syncthreads();
    // orient_sign with its floating-point filter inlined (the call is left for what the filter cannot tell)
    auto turn = [&](double ax, double ay, double bx, double by, double cx, double cy) {
        const double l = (bx - ax) * (cy - ay), r = (by - ay) * (cx - ax);
        const double det = l - r;
        const double bound = 3.3306690738754716e-16 * (fabs(l) + fabs(r));
        if (det > bound) return 1;
        if (det < -bound) return -1;
        return orient_sign(ax, ay, bx, by, cx, cy);
    };
    int keep = 0;
    if (have) {
        keep = 1;
        if (q > 0 && q < L - 1) {
            // A: among the candidates before q; B: among those after q
            double ax = fx[0], ay = fy[0], bx = fx[L - 1], by = fy[L - 1];
            for (int c = 1; c < q; c++) {
                const double cx = fx[c], cy = fy[c];
                if (turn(ax, ay, qx, qy, cx, cy) < 0) {         // c right of A -> q
                    ax = cx;
                    ay = cy;
                }
            }
            for (int c = q + 1; c < L - 1; c++) {
                const double cx = fx[c], cy = fy[c];
                if (turn(qx, qy, bx, by, cx, cy) < 0) {         // c right of q -> B
                    bx = cx;
                    by = cy;
                }
            }
            keep = turn(ax, ay, qx, qy, bx, by) > 0 ? 1 : 0;
        }
    }
    int total;
    const int at = block_exscan(keep, scratch, &total);         // (its barriers end all reads of the list)
    if (keep)
        list[at] = me;
    __syncthreads();
    return total;
}

// One pruning round over list[0..L), L <= HULL_PRUNE_PER * threads: every candidate but the two ends stays exactly
// when the walk from the candidate before it to the one after it turns left at it.  A candidate that goes lies
// on or above the chord of two points of the set, so it is no vertex of the chain whatever else goes in the same
// round; a vertex of the chain lies strictly below every such chord and never goes.  Every lane works (a thread
// holds a run of at most HULL_PRUNE_PER consecutive candidates), the survivors move up in order.  Returns how
// many stayed.
constexpr int HULL_PRUNE_PER = 8;
constexpr int HULL_PRUNE_ROUNDS = 48;
// CYCLIC: the list is a closed walk (the candidate after the last is the first); the candidates numbered 0 and
// `top` always stay.
template <class IdxT, bool CYCLIC = false>
__device__ __forceinline__ int chain_prune_round(const double *px, const double *py, IdxT *list, int L, int *scratch,
                                                 int top = 0)
{
    const int per = (L + HULL_NT - 1) / HULL_NT;
    const int b0 = min(L, (int)threadIdx.x * per), b1 = min(L, b0 + per);
    IdxT mine[HULL_PRUNE_PER];
    uint32_t keep = 0u;
    if (b0 < b1) {
        // the run's candidates and the two around it: one pass, every point loaded once
        IdxT prev = list[b0 > 0 ? b0 - 1 : (CYCLIC ? L - 1 : 0)];
        double ax = px[prev], ay = py[prev];
        IdxT cur = list[b0];
        double qx = px[cur], qy = py[cur];
#pragma unroll
        for (int u = 0; u < HULL_PRUNE_PER; u++) {
            const int i = b0 + u;
            if (i < b1) {
                const IdxT nxt = list[i + 1 < L ? i + 1 : (CYCLIC ? 0 : L - 1)];
                const double bx = px[nxt], by = py[nxt];
                bool stay = CYCLIC ? ((int)cur == 0 || (int)cur == top) : (i == 0 || i == L - 1);
                if (!stay) {
                    const double l = (qx - ax) * (by - ay), r = (qy - ay) * (bx - ax);
                    const double det = l - r;
                    const double bound = 3.3306690738754716e-16 * (fabs(l) + fabs(r));
                    stay = det > bound ? true : det < -bound ? false : orient_sign(ax, ay, qx, qy, bx, by) > 0;
                }
                mine[u] = cur;
                keep |= stay ? 1u << u : 0u;
                ax = qx;
                ay = qy;
                qx = bx;
                qy = by;
                cur = nxt;
            }
        }
    }
    int total;
    int at = block_exscan(__popc(keep), scratch, &total);           // (its barriers end all reads of the list)
#pragma unroll
    for (int u = 0; u < HULL_PRUNE_PER; u++)
        if (keep >> u & 1u)
            list[at++] = mine[u];
    __syncthreads();
    return total;
}

// Both chains at once for m sorted points in reach of the pruning rounds (m <= HULL_PRUNE_PER * threads): the
// points below the line from the first to the last point in ascending order, then those above it in descending
// order, form one closed walk with the two ends; pruning rounds on it leave the hull, counter-clockwise from the
// first point - the lower chain followed by the upper one.  Returns the number of vertices at list[0..), or -1
// when the rounds did not come to an end (the caller then builds the chains one by one).
template <class IdxT>
__device__ int hull_by_pruning(const double *px, const double *py, IdxT *list, int m, int *scratch)
{
    const double ax = px[0], ay = py[0], bx = px[m - 1], by = py[m - 1];
    const int per = (m + HULL_NT - 1) / HULL_NT;
    const int b0 = min(m, (int)threadIdx.x * per), b1 = min(m, b0 + per);
    uint32_t below = 0u, above = 0u;
#pragma unroll
    for (int u = 0; u < HULL_PRUNE_PER; u++) {
        const int i = b0 + u;
        if (i < b1 && i > 0 && i < m - 1) {
            const double cx = px[i], cy = py[i];
            const double l = (bx - ax) * (cy - ay), r = (by - ay) * (cx - ax);
            const double det = l - r;
            const double bound = 3.3306690738754716e-16 * (fabs(l) + fabs(r));
            const int sgn = det > bound ? 1 : det < -bound ? -1 : orient_sign(ax, ay, bx, by, cx, cy);
            below |= sgn < 0 ? 1u << u : 0u;
            above |= sgn > 0 ? 1u << u : 0u;
        }
    }
    int total;
    const int at = block_exscan(__popc(below) | (__popc(above) << 16), scratch, &total);   // (m < 65536)
    const int n_below = total & 0xffff, n_above = total >> 16;
    int at_b = 1 + (at & 0xffff), at_a = 1 + n_below + n_above - (at >> 16);
#pragma unroll
    for (int u = 0; u < HULL_PRUNE_PER; u++) {
        if (below >> u & 1u)
            list[at_b++] = (IdxT)(b0 + u);
        if (above >> u & 1u)
            list[at_a--] = (IdxT)(b0 + u);
    }
    if (threadIdx.x == 0) {
        list[0] = (IdxT)0;
        list[1 + n_below] = (IdxT)(m - 1);
    }
    __syncthreads();
    int L = n_below + n_above + 2;
    for (int round = 0; round < HULL_PRUNE_ROUNDS; round++) {
        const int left = chain_prune_round<IdxT, true>(px, py, list, L, scratch, m - 1);
        if (left == L)
            return L;
        L = left;
    }
    return -1;
}

// Monotone chain over the candidates list[0..L) -> hull chain at list[0..return)
template <class IdxT>
__device__ int64_t chain_reduce(const double *px, const double *py, IdxT *list, int64_t L, int64_t *pong,
                                int *scratch)
{
    // Pruning rounds first: each costs a handful of instructions per candidate with every lane at work, where a
    // round of chunks walks 16 candidates per thread one after the other on a few lanes and the finish below
    // compares every candidate with every other.  A round that takes nothing away has left the chain itself.
    // Inputs on which the rounds go on and on (each takes at least one candidate) are cut short and go through
    // the chunks below, which always end.
    if (L <= (int64_t)HULL_PRUNE_PER * HULL_NT) {
        for (int round = 0; round < HULL_PRUNE_ROUNDS; round++) {
            const int left = chain_prune_round<IdxT>(px, py, list, (int)L, scratch);
            if (left == L)
                return L;
            L = left;
        }
    }
    constexpr int C0 = 16;
    int64_t c = C0;
    while (true) {
        if (L <= HULL_BRUTE_MAX)
            return chain_finish<IdxT>(px, py, list, (int)L, scratch);
        if (L <= c) {                                   // (all of them vertices, round after round:) one thread finishes
            __shared__ int last_k;
            if (threadIdx.x == 0)
                last_k = chain_chunk(px, py, list, L, L);
            __syncthreads();
            const int k = last_k;
            __syncthreads();
            return k;
        }
        while ((L + c - 1) / c > HULL_NT)
            c <<= 1;
        const int kept = chain_chunk(px, py, list, L, c);
        const int64_t begin = (int64_t)threadIdx.x * c;
        int total;
        if (c == C0) {
            // chunks of the first size: a thread's survivors wait in its registers while the others read theirs
            IdxT mine[C0];
#pragma unroll
            for (int i = 0; i < C0; i++)
                mine[i] = i < kept ? list[begin + i] : (IdxT)0;
            const int at = block_exscan(kept, scratch, &total);     // (its barriers end all reads of the list)
#pragma unroll
            for (int i = 0; i < C0; i++)
                if (i < kept)
                    list[at + i] = mine[i];
            __syncthreads();
        } else {
            const int at = block_exscan(kept, scratch, &total);
            // compact through the slice's own hull output buffer (global memory)
            for (int i = 0; i < kept; i++)
                pong[at + i] = (int64_t)list[begin + i];
            __syncthreads();
            for (int64_t i = threadIdx.x; i < total; i += HULL_NT)
                list[i] = (IdxT)pong[i];
            __syncthreads();
        }
        if (2 * (int64_t)total > L)
            c <<= 1;
        L = total;
    }
}

// Sort of n points held in LDS (n <= HULL_PRUNE_PER * threads) by (x, y) in a few barriers: a counting sort on a
// bucket number that never decreases with x, then every bucket put in order by one thread.  The bucket number
// is uniform in the angle of a ring's points seen from its centre line (their x crowd towards both ends, as
// 1 / sqrt(t (1 - t)) in t = (x - min) / (max - min)), so buckets of a ring hold a point or two.  Returns false
// and leaves the points alone when the input does not suit (a coordinate that is not finite, all x equal, a
// bucket of more than HULL_BUCKET_MAX points as quantised coordinates make them): the caller's sorting network,
// which takes anything, runs then.  cnt: n_buckets counters.
constexpr int HULL_BUCKET_MAX = 24;
__device__ bool bucket_sort_lds(double *px, double *py, int n, uint32_t *cnt, int n_buckets, double *red,
                                int *scan_scratch)
{
    const int tid = threadIdx.x, NT = HULL_NT, lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double qx[HULL_PRUNE_PER], qy[HULL_PRUNE_PER];
    double lo = inf, hi = -inf;
    bool odd = false;
#pragma unroll
    for (int u = 0; u < HULL_PRUNE_PER; u++) {
        const int i = tid + u * NT;
        qx[u] = qy[u] = 0.0;
        if (i < n) {
            qx[u] = px[i];
            qy[u] = py[i];
            odd |= !(fabs(qx[u]) < inf) || !(fabs(qy[u]) < inf);
            lo = fmin(lo, qx[u]);
            hi = fmax(hi, qx[u]);
        }
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0) {
        red[wave] = lo;
        red[16 + wave] = hi;
    }
    for (int b = tid; b < n_buckets; b += NT)
        cnt[b] = 0u;
    if (__syncthreads_or(odd))
        return false;
    for (int w = 0; w < nw; w++) {
        lo = fmin(lo, red[w]);
        hi = fmax(hi, red[16 + w]);
    }
    const double range = hi - lo;
    if (!(range > 0.0) || !(range < inf))
        return false;                                   // (uniform over the workgroup)
    const double scale = 1.0 / range, nb = (double)n_buckets;
    uint32_t bk[HULL_PRUNE_PER], rk[HULL_PRUNE_PER];
#pragma unroll
    for (int u = 0; u < HULL_PRUNE_PER; u++) {
        const int i = tid + u * NT;
        bk[u] = rk[u] = 0u;
        if (i < n) {
            const double t = fmin((qx[u] - lo) * scale, 1.0);
            const double g = t < 0.5 ? sqrt(0.5 * t) : 1.0 - sqrt(0.5 * (1.0 - t));
            const int b = (int)(g * nb);
            bk[u] = (uint32_t)(b < n_buckets - 1 ? b : n_buckets - 1);
            rk[u] = atomicAdd(&cnt[bk[u]], 1u);
        }
    }
    __syncthreads();
    // first slot of every bucket (a run of buckets per thread), and the fullest bucket
    const int seg = (n_buckets + NT - 1) / NT;
    const int b0 = min(n_buckets, tid * seg), b1 = min(n_buckets, b0 + seg);
    int sum = 0;
    uint32_t most = 0u;
    for (int b = b0; b < b1; b++) {
        const uint32_t c = cnt[b];
        sum += (int)c;
        most = c > most ? c : most;
    }
    int total;
    int at = block_exscan(sum, scan_scratch, &total);
    if (__syncthreads_or(most > (uint32_t)HULL_BUCKET_MAX))
        return false;
    for (int b = b0; b < b1; b++) {
        const uint32_t c = cnt[b];
        cnt[b] = (uint32_t)at;
        at += (int)c;
    }
    __syncthreads();
    // every point sits in a register: into its bucket, in place
#pragma unroll
    for (int u = 0; u < HULL_PRUNE_PER; u++) {
        const int i = tid + u * NT;
        if (i < n) {
            const uint32_t d = cnt[bk[u]] + rk[u];
            px[d] = qx[u];
            py[d] = qy[u];
        }
    }
    __syncthreads();
    for (int b = tid; b < n_buckets; b += NT) {
        const int s0 = (int)cnt[b], s1 = b + 1 < n_buckets ? (int)cnt[b + 1] : n;
        for (int i = s0 + 1; i < s1; i++) {
            const double kx = px[i], ky = py[i];
            int j = i - 1;
            while (j >= s0 && lex_less(kx, ky, px[j], py[j])) {
                px[j + 1] = px[j];
                py[j + 1] = py[j];
                j--;
            }
            px[j + 1] = kx;
            py[j + 1] = ky;
        }
    }
    __syncthreads();
    return true;
}

// ---- the filter for long slices
constexpr int HULL_SECTORS = 512;
constexpr int64_t HULL_FILTER_MAX_SLICES = 4096;       // (the per-slice sector table is 4 KB)

struct HullSliceAux {
    double cx, cy, spare;               // centroid
    long long filtered;                 // 1: the slice went through the filter, kept points are in the workspace
    unsigned long long kept;            // how many
    long long sectors;                  // sectors in use: fewer for shorter slices, so that each still holds a crowd
    unsigned long long sect[HULL_SECTORS];      // bit pattern + 1 of the largest squared distance per sector; after
                                                // the second per-slice step: bit pattern of the sector's threshold
};

// sector of a direction: uniform in the "diamond angle" dy / (|dx| + |dy|), whose slope against the
// true angle lies in [1/2, 1], so one of K sectors is at most 8 / K radians wide
__device__ __forceinline__ int sector_of(double dx, double dy, int K)
{
    const double a = fabs(dx) + fabs(dy);
    if (!(a > 0.0))
        return 0;
    const double p = dy / a;
    const double d = dx >= 0.0 ? (dy >= 0.0 ? p : 4.0 + p) : 2.0 - p;      // [0, 4)
    const int j = (int)(d * (double)(K / 4));
    return j < 0 ? 0 : (j > K - 1 ? K - 1 : j);
}

// PASS 0: partial coordinate sums.  PASS 1: per-sector largest squared distance.  PASS 2: copy the points
// that are not provably inside to the workspace.  Slices that fit the LDS path are left alone.
template <int PASS>
__global__ __launch_bounds__(SPLIT_BLOCK) void hull_filter_kernel(const double *__restrict__ x,
                                                                 const double *__restrict__ y,
                                                                 const int64_t *__restrict__ offsets, int64_t S,
                                                                 const int64_t *__restrict__ segstart, int lds_cap,
                                                                 HullSliceAux *__restrict__ aux,
                                                                 double *__restrict__ part, double *__restrict__ wsx,
                                                                 double *__restrict__ wsy)
{
    __shared__ double red[2 * 16];
    __shared__ unsigned long long lsect[HULL_SECTORS];
    SegRef r;
    if (!find_segment(segstart, offsets, S, x, y, r))
        return;
    const int64_t off = offsets[r.slice];
    if (offsets[r.slice + 1] - off <= lds_cap)
        return;
    if constexpr (PASS == 0) {
        // (the centre the sectors are laid about has to lie inside the hull and no more: the mean of the first
        // eighth of every segment, at least 1024 points of it, serves as well as the mean of all)
        SegRef q = r;
        q.n = min(r.n, max(1024, r.n / 8));
        double s1[2] = {0.0, 0.0};
        segment_points(q, [&](double px, double py, int32_t, bool valid) {
            s1[0] += valid ? px : 0.0;
            s1[1] += valid ? py : 0.0;
        });
        block_sum<2>(s1, red);
        if (threadIdx.x == 0) {
            part[r.seg * SPLIT_PART] = s1[0];
            part[r.seg * SPLIT_PART + 1] = s1[1];
            part[r.seg * SPLIT_PART + 2] = (double)q.n;
        }
    } else if constexpr (PASS == 1) {
        for (int j = threadIdx.x; j < HULL_SECTORS; j += SPLIT_BLOCK)
            lsect[j] = 0ull;
        __syncthreads();
        const double cx = aux[r.slice].cx, cy = aux[r.slice].cy;
        const int K = (int)aux[r.slice].sectors;
        segment_points(r, [&](double px, double py, int32_t, bool valid) {
            const double dx = px - cx, dy = py - cy;
            const double r2 = dx * dx + dy * dy;
            if (valid && r2 < __longlong_as_double(0x7ff0000000000000ll))
                atomicMax(&lsect[sector_of(dx, dy, K)], (unsigned long long)__double_as_longlong(r2) + 1ull);
        });
        __syncthreads();
        for (int j = threadIdx.x; j < HULL_SECTORS; j += SPLIT_BLOCK)
            if (lsect[j])
                atomicMax(&aux[r.slice].sect[j], lsect[j]);
    } else {
        const double cx = aux[r.slice].cx, cy = aux[r.slice].cy;
        const int K = (int)aux[r.slice].sectors;
        for (int j = threadIdx.x; j < K; j += SPLIT_BLOCK)
            lsect[j] = aux[r.slice].sect[j];                    // by now: the sector's threshold (bits of a double)
        __syncthreads();
        const int lane = threadIdx.x & 63;
        segment_points(r, [&](double px, double py, int32_t, bool valid) {
            const double dx = px - cx, dy = py - cy;
            const double r2 = dx * dx + dy * dy;
            const double thr = __longlong_as_double((long long)lsect[sector_of(dx, dy, K)]);
            const bool keep = valid && !(r2 < thr);            // (a NaN stays)
            const unsigned long long m = __ballot(keep);
            if (m) {
                unsigned long long base = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader)
                    base = atomicAdd(&aux[r.slice].kept, (unsigned long long)__popcll(m));
                base = __shfl(base, leader, WAVE);
                if (keep) {
                    const unsigned long long at = base + __popcll(m & ((1ull << lane) - 1ull));
                    wsx[off + at] = px;
                    wsy[off + at] = py;
                }
            }
        });
    }
}

// per slice, between the passes: STEP 0 centroid from the partial sums and a clean table,
// STEP 1 the threshold from the table
template <int STEP>
__global__ __launch_bounds__(WAVE) void hull_filter_slice_kernel(const int64_t *__restrict__ offsets,
                                                                const int64_t *__restrict__ segstart, int lds_cap,
                                                                const double *__restrict__ part,
                                                                HullSliceAux *__restrict__ aux)
{
    const int64_t s = blockIdx.x;
    const int64_t n = offsets[s + 1] - offsets[s];
    HullSliceAux &a = aux[s];
    if (n <= lds_cap) {
        if (threadIdx.x == 0 && STEP == 0)
            a.filtered = 0;
        return;
    }
    if constexpr (STEP == 0) {
        double v[3] = {0.0, 0.0, 0.0};
        const int64_t seg0 = segstart[s], nseg = segstart[s + 1] - seg0;
        for (int64_t j = threadIdx.x; j < nseg; j += WAVE) {
            v[0] += part[(seg0 + j) * SPLIT_PART];
            v[1] += part[(seg0 + j) * SPLIT_PART + 1];
            v[2] += part[(seg0 + j) * SPLIT_PART + 2];
        }
        v[0] = wave_sum(v[0]);
        v[1] = wave_sum(v[1]);
        v[2] = wave_sum(v[2]);
        for (int j = threadIdx.x; j < HULL_SECTORS; j += WAVE)
            a.sect[j] = 0ull;
        if (threadIdx.x == 0) {
            a.cx = v[0] / v[2];
            a.cy = v[1] / v[2];
            a.filtered = 1;
            a.kept = 0ull;
            a.sectors = n >= 32768 ? HULL_SECTORS : (n >= 16384 ? HULL_SECTORS / 2 : HULL_SECTORS / 4);
        }
    } else {
        const int K = (int)a.sectors;
        unsigned long long m = ~0ull;
        for (int j = threadIdx.x; j < K; j += WAVE)
            m = a.sect[j] < m ? a.sect[j] : m;
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) {
            const unsigned long long o = __shfl_xor(m, k, WAVE);
            m = o < m ? o : m;
        }
        // Every sector holds a point (else, or with a centroid that is not finite, nothing is dropped): the
        // hull then contains a disc about the centroid O.  For a point P of sector j take the farthest points
        // A, B of the two flanking sectors: P's direction lies between theirs, the angle AOB is at most three
        // sector widths (24 / K), and the segment AB stays farther from O than min(|OA|, |OB|) cos(12 / K).
        // A point of sector j closer than that lies inside the triangle OAB, so it is no hull vertex.  The
        // thresholds follow the outline sector by sector, so a centroid that outliers have pulled off the
        // centre costs nothing.  They replace the table in place (reads of the whole wave first, then writes).
        const bool live = m != 0ull && a.cx == a.cx && a.cy == a.cy;
        const double c = cos(12.0 / (double)K);
        const double shrink = (c * c) * (1.0 - 1e-9);
        constexpr int PER = HULL_SECTORS / WAVE;
        double thr[PER];
#pragma unroll
        for (int t = 0; t < PER; t++) {
            const int j = threadIdx.x + t * WAVE;
            thr[t] = 0.0;
            if (j < K && live) {
                const unsigned long long p = a.sect[(j + K - 1) % K], q = a.sect[(j + 1) % K];
                const unsigned long long lo = p < q ? p : q;
                thr[t] = __longlong_as_double((long long)(lo - 1ull)) * shrink;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < PER; t++) {
            const int j = threadIdx.x + t * WAVE;
            if (j < K)
                a.sect[j] = (unsigned long long)__double_as_longlong(thr[t]);
        }
    }
}

template <bool IN_LDS, class IdxT>
__global__ __launch_bounds__(HULL_BLOCK) void unique_hull_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, int lds_cap,
    double *__restrict__ ux, double *__restrict__ uy, int64_t *__restrict__ n_unique, int64_t *__restrict__ hull_idx,
    int64_t *__restrict__ n_hull, int32_t *__restrict__ status, const int64_t *__restrict__ hull_start,
    double *__restrict__ wsx, double *__restrict__ wsy, uint32_t *__restrict__ wslist,
    uint32_t *__restrict__ wspark, const HullSliceAux *__restrict__ aux, int hull_only, int pass)
{
    extern __shared__ double lds_dyn[];
    __shared__ double red[9 * 16];
    __shared__ int scan_scratch[17];
    __shared__ int found_at;
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s];
    int64_t n = offsets[s + 1] - off;
    const double *src_x = x + off, *src_y = y + off;
    const bool filtered = aux && aux[s].filtered == 1;
    if (filtered) {
        n = (int64_t)aux[s].kept;                        // what the filter kept waits in the workspace
        src_x = wsx + off;
        src_y = wsy + off;
    }
    if (IN_LDS ? (n > lds_cap) : (n <= lds_cap))
        return;                                          // the other launch takes this slice
    // pass 1: the narrow launch, for what the filter left of long slices when that is short; pass 2: the wide
    // launch that follows it and leaves those slices alone; pass 0: one launch for everything that fits the LDS
    if (IN_LDS && ((pass == 1 && !filtered) || (pass == 2 && filtered && n <= HULL_SMALL_CAP)))
        return;
    double *px, *py;
    IdxT *list;
    if constexpr (IN_LDS) {
        px = lds_dyn;
        py = lds_dyn + lds_cap;
        list = reinterpret_cast<IdxT *>(lds_dyn + 2 * lds_cap);
    } else {
        px = wsx + off;
        py = wsy + off;
        list = reinterpret_cast<IdxT *>(wslist + off);
    }
    const int tid = threadIdx.x;
    HULL_MARK_INIT();
    if (tid == 0) {
        n_unique[s] = 0;
        n_hull[s] = 0;
        status[s] = PCFIT_ST_OK;
    }
    if (IN_LDS || !filtered) {                           // (filtered and in global memory: already in place)
        for (int64_t i = tid; i < n; i += HULL_NT) {
            px[i] = src_x[i];
            py[i] = src_y[i];
        }
    }
    __syncthreads();
    // ---- a caller that wants hulls only: the interior filter of the long slices, here inside the LDS and
    // before the sort (same thresholds, see hull_filter_slice_kernel); what is left is a few hundred points
    if constexpr (IN_LDS) {
        constexpr int K = HULL_SECTORS / 4;
        constexpr int PER = HULL_LDS_CAP / HULL_BLOCK;
        __shared__ unsigned long long fsect[K];
        __shared__ double fthr[K];
        __shared__ int fcount, fdead;
        if (hull_only && !filtered && n > 1024) {
            double s1[2] = {0.0, 0.0};
            for (int64_t i = tid; i < n; i += HULL_NT) {
                s1[0] += px[i];
                s1[1] += py[i];
            }
            block_sum<2>(s1, red);
            const double cx = s1[0] / (double)n, cy = s1[1] / (double)n;
            if (tid < K)
                fsect[tid] = 0ull;
            if (tid == 0) {
                fcount = 0;
                fdead = !(cx == cx && cy == cy);
            }
            __syncthreads();
            double qx[PER], qy[PER], q2[PER];
            int qs[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int64_t i = tid + (int64_t)u * HULL_BLOCK;
                qs[u] = -1;
                if (i < n) {
                    qx[u] = px[i];
                    qy[u] = py[i];
                    const double dx = qx[u] - cx, dy = qy[u] - cy;
                    q2[u] = dx * dx + dy * dy;
                    qs[u] = sector_of(dx, dy, K);
                    if (q2[u] < __longlong_as_double(0x7ff0000000000000ll))
                        atomicMax(&fsect[qs[u]], (unsigned long long)__double_as_longlong(q2[u]) + 1ull);
                }
            }
            __syncthreads();
            if (tid < K && fsect[tid] == 0ull)
                fdead = 1;                                 // an empty sector: nothing may be dropped
            __syncthreads();
            if (tid < K) {
                const double c = cos(12.0 / (double)K);
                const unsigned long long a = fsect[(tid + K - 1) % K], b = fsect[(tid + 1) % K];
                const unsigned long long lo = a < b ? a : b;
                fthr[tid] = fdead ? 0.0 : __longlong_as_double((long long)(lo - 1ull)) * ((c * c) * (1.0 - 1e-9));
            }
            __syncthreads();
            // every thread holds its points in registers: the kept ones go back to the front of the arrays
            // in any order (the sort follows)
#pragma unroll
            for (int u = 0; u < PER; u++) {
                if (qs[u] >= 0 && !(q2[u] < fthr[qs[u]])) {
                    const int at = atomicAdd(&fcount, 1);
                    px[at] = qx[u];
                    py[at] = qy[u];
                }
            }
            __syncthreads();
            n = fcount;
        }
    }
    HULL_MARK(0);

    // ---- sort (ascending-only bitonic network, virtual +inf padding).  Strides are powers of two: the
    // partner indices come from shifts and masks on 32-bit values (n < 2^31).
    const uint32_t n32 = (uint32_t)n;
    int lp = 0;
    while ((1u << lp) < n32 && lp < 31) lp++;
    const uint32_t half = lp > 0 ? (1u << (lp - 1)) : 0u;
    // (a few hundred points and more in LDS: the counting sort first, where the input suits it; the list's
    // room, not in use yet, holds its counters)
    bool sorted = false;
    if constexpr (IN_LDS) {
        if (n >= 96 && n <= (int64_t)HULL_PRUNE_PER * HULL_NT && lds_cap >= 256)
            sorted = bucket_sort_lds(px, py, (int)n, reinterpret_cast<uint32_t *>(list), lds_cap / 2, red, scan_scratch);
    }
    for (int lk = 1; lk <= (sorted ? 0 : lp); lk++) {
        for (int ld = lk - 1; ld >= 0; ld--) {
            const uint32_t d = 1u << ld;
            const bool flip = ld == lk - 1;
            for (uint32_t q = tid; q < half; q += HULL_NT) {
                const uint32_t blk = q >> ld, o = q & (d - 1u);
                uint32_t i, j;
                if (flip) {
                    i = (blk << lk) + o;
                    j = (blk << lk) + ((1u << lk) - 1u - o);
                } else {
                    i = (blk << (ld + 1)) + o;
                    j = i + d;
                }
                if (j < n32) {
                    const double ax = px[i], ay = py[i], bx = px[j], by = py[j];
                    if (lex_less(bx, by, ax, ay)) {
                        px[i] = bx; py[i] = by;
                        px[j] = ax; py[j] = ay;
                    }
                }
            }
            __syncthreads();
        }
    }

    HULL_MARK(1);
    // ---- unique: keep the first of every run of equal rows
    const int64_t per = (n + HULL_NT - 1) / HULL_NT;
    const int64_t b0 = min(n, (int64_t)tid * per), b1 = min(n, b0 + per);
    int keep = 0;
    for (int64_t i = b0; i < b1; i++)
        keep += (i == 0) || px[i] != px[i - 1] || py[i] != py[i - 1];
    int total_unique;
    const int at0 = block_exscan(keep, scan_scratch, &total_unique);
    {
        // stage through the output arrays: all reads of px/py above are done (barriers in the scan)
        int at = at0;
        for (int64_t i = b0; i < b1; i++) {
            if ((i == 0) || px[i] != px[i - 1] || py[i] != py[i - 1]) {
                ux[off + at] = px[i];
                uy[off + at] = py[i];
                at++;
            }
        }
    }
    __syncthreads();
    const int64_t m = total_unique;
    for (int64_t i = tid; i < m; i += HULL_NT) {
        px[i] = ux[off + i];
        py[i] = uy[off + i];
    }
    __syncthreads();
    if (tid == 0)
        n_unique[s] = m;
    if (m < 3) {
        if (tid == 0)
            status[s] = PCFIT_ST_SPARSE;
        return;
    }

    HULL_MARK(2);
    // ---- rank check (circle_fitting.py:14-19)
    {
        double s1[2] = {0.0, 0.0};
        for (int64_t i = tid; i < m; i += HULL_NT) {
            s1[0] += px[i];
            s1[1] += py[i];
        }
        block_sum<2>(s1, red);
        const double mx = s1[0] / (double)m, my = s1[1] / (double)m;
        double sc[3] = {0.0, 0.0, 0.0};
        for (int64_t i = tid; i < m; i += HULL_NT) {
            const double u = px[i] - mx, v = py[i] - my;
            sc[0] += u * u;
            sc[1] += u * v;
            sc[2] += v * v;
        }
        block_sum<3>(sc, red);
        const double phi = 0.5 * atan2(2 * sc[1], sc[0] - sc[2]);
        const double c = cos(phi), sn = sin(phi);
        double pr[2] = {0.0, 0.0};
        for (int64_t i = tid; i < m; i += HULL_NT) {
            const double u = px[i] - mx, v = py[i] - my;
            const double a = c * u + sn * v, b = c * v - sn * u;
            pr[0] += a * a;
            pr[1] += b * b;
        }
        block_sum<2>(pr, red);
        double s0 = sqrt(pr[0]), s1v = sqrt(pr[1]);
        if (s1v > s0) {
            const double t = s0;
            s0 = s1v;
            s1v = t;
        }
        if (!(s1v > 1e-12 * (s0 + 1e-12))) {
            if (tid == 0)
                status[s] = PCFIT_ST_RANK;
            return;
        }
    }

    HULL_MARK(3);
    int64_t *pong = hull_idx + off;
    uint32_t *park = wspark + off;
    int64_t h = -1;
    if (m <= (int64_t)HULL_PRUNE_PER * HULL_NT && m < 65536) {
        // ---- hull: both chains at once
        h = hull_by_pruning<IdxT>(px, py, list, (int)m, scan_scratch);
        for (int64_t i = tid; i < h; i += HULL_NT)
            pong[i] = (int64_t)list[i];
    }
    if (h < 0) {
        // ---- hull: lower chain (left to right), then upper chain (right to left)
        for (int64_t i = tid; i < m; i += HULL_NT)
            list[i] = (IdxT)i;
        __syncthreads();
        const int64_t hl = chain_reduce<IdxT>(px, py, list, m, pong, scan_scratch);
        for (int64_t i = tid; i < hl - 1; i += HULL_NT)     // the lower chain waits in park
            park[i] = (uint32_t)list[i];
        __syncthreads();
        for (int64_t i = tid; i < m; i += HULL_NT)
            list[i] = (IdxT)(m - 1 - i);
        __syncthreads();
        const int64_t hu = chain_reduce<IdxT>(px, py, list, m, pong, scan_scratch);
        h = (hl - 1) + (hu - 1);
        // both chains end on the other's first vertex: drop each chain's last entry
        for (int64_t i = tid; i < hl - 1; i += HULL_NT)
            pong[i] = (int64_t)park[i];
        for (int64_t i = tid; i < hu - 1; i += HULL_NT)
            pong[hl - 1 + i] = (int64_t)list[i];
    }
    HULL_MARK(4);
    if (tid == 0) {
        n_hull[s] = h;
        found_at = 0x7fffffff;
    }
    __syncthreads();
    // optional rotation of the walk to a requested first vertex
    const int64_t want = hull_start && !filtered ? hull_start[s] : -1;
    if (want > 0) {
        for (int64_t i = tid; i < h; i += HULL_NT) {
            const int64_t v = pong[i];
            park[i] = (uint32_t)v;
            if (v == want)
                atomicMin(&found_at, (int)i);
        }
        __syncthreads();
        const int p0 = found_at;
        if (p0 != 0x7fffffff && p0 > 0) {
            for (int64_t i = tid; i < h; i += HULL_NT)
                pong[i] = (int64_t)park[(i + p0) % h];
        }
    }
}

// Workspace: wsx, wsy (8 B per point each), wslist, wspark (4 B each), then - for batches of at most
// HULL_FILTER_MAX_SLICES slices - the filter's segment table, partial sums and per-slice records.
struct HullLayout {
    size_t total, seg_off, part_off, aux_off, b1_off, hc_off, ls_off, bytes;
    int64_t max_segs;
    bool filter, full;
};

// full: room for the sort of whole long slices as well (a second buffer pair, the hull vertices' coordinates
// and the tile tables of longsort.hip)
static HullLayout hull_layout(size_t total, int64_t S, bool full)
{
    HullLayout l;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    l.total = total;
    l.filter = S <= HULL_FILTER_MAX_SLICES;
    l.full = full && l.filter;
    l.max_segs = (int64_t)(total / SPLIT_SEG) + S + 1;
    l.seg_off = up(total * 24 + 1024);
    l.part_off = l.seg_off + (l.filter ? up((size_t)(S + 2) * 8) : 0);
    l.aux_off = l.part_off + (l.filter ? up((size_t)l.max_segs * SPLIT_PART * 8) : 0);
    l.b1_off = l.aux_off + (l.filter ? up((size_t)std::max<int64_t>(S, 1) * sizeof(HullSliceAux)) : 0);
    l.hc_off = l.b1_off + (l.full ? up(total * 16) : 0);
    l.ls_off = l.hc_off + (l.full ? up(total * 16) : 0);
    l.bytes = l.ls_off + (l.full ? long_sort_workspace_bytes((int64_t)total, S) : 0);
    return l;
}

// Filtered long slices, when the caller wants every distinct point: the coordinates of the hull vertices are
// set aside before the full distinct set replaces the kept one in ux / uy ...
__global__ __launch_bounds__(256) void hull_coords_kernel(const int64_t *__restrict__ offsets,
                                                         const HullSliceAux *__restrict__ aux,
                                                         const double *__restrict__ ux, const double *__restrict__ uy,
                                                         const int64_t *__restrict__ hull_idx,
                                                         const int64_t *__restrict__ n_hull, double *__restrict__ hcx,
                                                         double *__restrict__ hcy)
{
    const int64_t s = blockIdx.x;
    if (aux[s].filtered != 1)
        return;
    const int64_t off = offsets[s], h = n_hull[s];
    for (int64_t i = threadIdx.x; i < h; i += 256) {
        const int64_t v = hull_idx[off + i];
        hcx[off + i] = ux[off + v];
        hcy[off + i] = uy[off + v];
    }
}

// ... and looked up in it afterwards (binary search by coordinates: every vertex is one of the distinct
// points), followed by the optional rotation of the walk to a requested first vertex.
__global__ __launch_bounds__(256) void hull_remap_kernel(const int64_t *__restrict__ offsets,
                                                        const HullSliceAux *__restrict__ aux,
                                                        const double *__restrict__ ux, const double *__restrict__ uy,
                                                        const int64_t *__restrict__ n_unique,
                                                        const int64_t *__restrict__ n_hull, double *__restrict__ hcx,
                                                        const double *__restrict__ hcy, int64_t *__restrict__ hull_idx,
                                                        const int64_t *__restrict__ hull_start)
{
    __shared__ int found_at;
    const int64_t s = blockIdx.x;
    if (aux[s].filtered != 1)
        return;
    const int64_t off = offsets[s], h = n_hull[s], m = n_unique[s];
    const int64_t want = hull_start ? hull_start[s] : -1;
    if (threadIdx.x == 0)
        found_at = 0x7fffffff;
    __syncthreads();
    long long *found = reinterpret_cast<long long *>(hcx + off);       // the x coordinate's slot takes the index
    for (int64_t i = threadIdx.x; i < h; i += 256) {
        const double vx = hcx[off + i], vy = hcy[off + i];
        int64_t lo = 0, hi = m;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (lex_less(ux[off + mid], uy[off + mid], vx, vy)) lo = mid + 1; else hi = mid;
        }
        found[i] = lo;
        if (lo == want)
            atomicMin(&found_at, (int)i);
    }
    __syncthreads();
    const int p0 = (want > 0 && found_at != 0x7fffffff) ? found_at : 0;
    for (int64_t i = threadIdx.x; i < h; i += 256)
        hull_idx[off + i] = found[(i + p0) % h];
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API size_t pcfit_unique_hull_workspace_bytes(int64_t total_points, int64_t n_slices)
{
    return hull_layout((size_t)(total_points > 0 ? total_points : 1), std::max<int64_t>(n_slices, 0), true).bytes;
}

size_t pcfit::unique_hull_workspace_bytes(int64_t total_points, int64_t n_slices, int hull_only)
{
    return hull_layout((size_t)(total_points > 0 ? total_points : 1), std::max<int64_t>(n_slices, 0), !hull_only).bytes;
}

PCFIT_API int pcfit_unique_hull_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                          int64_t max_slice_len, double *ux, double *uy, int64_t *n_unique,
                                          int64_t *hull_idx, int64_t *n_hull, int32_t *status,
                                          const int64_t *hull_start, void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    PCFIT_ENTER();
    return unique_hull_dev(x, y, offsets, n_slices, max_slice_len, ux, uy, n_unique, hull_idx, n_hull, status,
                           hull_start, workspace, workspace_bytes, stream, 0);
}

// hull_only: the caller wants the hulls and no list of distinct points; slices of more than 1024 points then drop
// their provably interior points before the sort, and ux / uy / n_unique describe what was kept; hull_start must
// be null.  Otherwise ux / uy / n_unique are the distinct points of the whole slice, whatever its length.
// The workspace is sized by unique_hull_workspace_bytes(total, n_slices, hull_only).
int pcfit::unique_hull_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                           int64_t max_slice_len, double *ux, double *uy, int64_t *n_unique, int64_t *hull_idx,
                           int64_t *n_hull, int32_t *status, const int64_t *hull_start, void *workspace,
                           size_t workspace_bytes, void *stream, int hull_only)
{
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "unique_hull: n_slices out of range");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && ux && uy && n_unique && hull_idx && n_hull && status,
                  "unique_hull: null pointer");
    PCFIT_REQUIRE(max_slice_len >= 0 && max_slice_len < (1ll << 31),
                  "unique_hull: max_slice_len must be given (0..2^31-1)");
    // total points are not known here (offsets live on the device): the caller sized the workspace with
    // pcfit_unique_hull_workspace_bytes(total, n_slices); the largest total whose layout fits is the one
    const bool full = !hull_only;
    if (!workspace || workspace_bytes < hull_layout(1, n_slices, full).bytes) {
        set_error("unique_hull: workspace missing");
        return PCFIT_ERR_WORKSPACE;
    }
    size_t lo = 1, hi = workspace_bytes / 24;
    while (lo < hi) {
        const size_t mid = lo + (hi - lo + 1) / 2;
        if (hull_layout(mid, n_slices, full).bytes <= workspace_bytes)
            lo = mid;
        else
            hi = mid - 1;
    }
    const HullLayout l = hull_layout(lo, n_slices, full);
    const size_t total = l.total;
    char *ws = static_cast<char *>(workspace);
    double *wsx = reinterpret_cast<double *>(ws);
    double *wsy = wsx + total;
    uint32_t *wslist = reinterpret_cast<uint32_t *>(wsy + total);
    uint32_t *wspark = wslist + total;
    hipStream_t st = as_stream(stream);
    int cap = (int)std::min<int64_t>(HULL_LDS_CAP, std::max<int64_t>(64, (max_slice_len + 63) / 64 * 64));
    const size_t lds = (size_t)cap * 18;
    HullSliceAux *aux = nullptr;
    if (l.filter && max_slice_len > cap) {
        int64_t *segstart = reinterpret_cast<int64_t *>(ws + l.seg_off);
        double *part = reinterpret_cast<double *>(ws + l.part_off);
        aux = reinterpret_cast<HullSliceAux *>(ws + l.aux_off);
        const dim3 grid((unsigned)l.max_segs), per_slice((unsigned)n_slices);
        split_map_kernel<<<dim3(1), dim3(1024), 0, st>>>(offsets, n_slices, segstart);
        hull_filter_kernel<0><<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, n_slices, segstart, cap, aux, part, wsx,
                                                                  wsy);
        hull_filter_slice_kernel<0><<<per_slice, dim3(WAVE), 0, st>>>(offsets, segstart, cap, part, aux);
        hull_filter_kernel<1><<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, n_slices, segstart, cap, aux, part, wsx,
                                                                  wsy);
        hull_filter_slice_kernel<1><<<per_slice, dim3(WAVE), 0, st>>>(offsets, segstart, cap, part, aux);
        hull_filter_kernel<2><<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, n_slices, segstart, cap, aux, part, wsx,
                                                                  wsy);
        PCFIT_LAUNCH_CHECK();
    }
    // more than 64 KB of dynamic LDS has to be granted explicitly (per device; idempotent)
    PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&unique_hull_kernel<true, uint16_t>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, HULL_LDS_CAP * 18));
    int pass = 0;
    if (aux && n_slices >= HULL_NARROW_MIN_SLICES) {
        // What the filter left of a long slice is typically a few hundred points: a workgroup of 1024 threads
        // spends its time in barriers on that, and only one of them fits a CU. With many slices a launch of
        // 256-thread workgroups with a quarter of the LDS (four to a CU) takes the remainders that fit it first.
        unique_hull_kernel<true, uint16_t><<<dim3((unsigned)n_slices), dim3(HULL_NARROW), (size_t)HULL_SMALL_CAP * 18,
                                             st>>>(x, y, offsets, HULL_SMALL_CAP, ux, uy, n_unique, hull_idx, n_hull,
                                                   status, hull_start, wsx, wsy, wslist, wspark, aux, hull_only, 1);
        PCFIT_LAUNCH_CHECK();
        pass = 2;
    }
    unique_hull_kernel<true, uint16_t><<<dim3((unsigned)n_slices), dim3(HULL_BLOCK), lds, st>>>(
        x, y, offsets, cap, ux, uy, n_unique, hull_idx, n_hull, status, hull_start, wsx, wsy, wslist, wspark, aux,
        hull_only, pass);
    PCFIT_LAUNCH_CHECK();
    if (max_slice_len > cap) {
        unique_hull_kernel<false, uint32_t><<<dim3((unsigned)n_slices), dim3(HULL_BLOCK), 0, st>>>(
            x, y, offsets, cap, ux, uy, n_unique, hull_idx, n_hull, status, hull_start, wsx, wsy, wslist, wspark, aux, 0,
            0);
        PCFIT_LAUNCH_CHECK();
    }
    if (aux && l.full) {
        // The filtered slices so far hold the distinct points of what the filter kept.  The caller wants all
        // of them: set the hull vertices' coordinates aside, sort the whole slices (the kept points' buffers
        // are free now and serve as one of the two buffer pairs), write the distinct points and look the
        // vertices up again.
        double *b1x = reinterpret_cast<double *>(ws + l.b1_off), *b1y = b1x + total;
        double *hcx = reinterpret_cast<double *>(ws + l.hc_off), *hcy = hcx + total;
        void *small = ws + l.ls_off;
        const dim3 per_slice((unsigned)n_slices);
        hull_coords_kernel<<<per_slice, dim3(256), 0, st>>>(offsets, aux, ux, uy, hull_idx, n_hull, hcx, hcy);
        PCFIT_LAUNCH_CHECK();
        int second = 0;
        PCFIT_TRY(long_sort_dev(x, y, wsx, wsy, b1x, b1y, offsets, n_slices, (int64_t)total, max_slice_len, cap, small,
                                &second, st));
        PCFIT_TRY(long_unique_dev(second ? b1x : wsx, second ? b1y : wsy, offsets, n_slices, (int64_t)total,
                                  max_slice_len, cap, small, ux, uy, n_unique, st));
        hull_remap_kernel<<<per_slice, dim3(256), 0, st>>>(offsets, aux, ux, uy, n_unique, n_hull, hcx, hcy, hull_idx,
                                                           hull_start);
        PCFIT_LAUNCH_CHECK();
    }
    return PCFIT_OK;
}

#ifdef PCFIT_PHASE_CLOCK
// tools only: clocks per phase of unique_hull_kernel (load, sort, unique, rank check, chains)
PCFIT_API int pcfit_debug_hull_phases(int64_t *out8, int reset)
{
    PCFIT_ENTER();
    PCFIT_HIP(hipDeviceSynchronize());
    PCFIT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(pcfit::g_hull_phase), 8 * sizeof(int64_t)));
    if (reset) {
        int64_t zero[8] = {0};
        PCFIT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pcfit::g_hull_phase), zero, sizeof(zero)));
    }
    return PCFIT_OK;
}
#endif
