// Distinct points, rank check and convex hull of every slice
// (circle_fitting.py:14-19 and :182-198), one 1024-thread workgroup per slice,
// gfx950.
//
//   sort     the slice's (x, y) pairs lexicographically (what np.unique(axis=0)
//            does) with an ascending-only bitonic network; slots past the end
//            count as +inf and are never stored;
//   unique   neighbours that compare equal are dropped by a prefix-sum
//            compaction;
//   rank     singular values of the centred cloud as root sums of squares of
//            its projections on the principal axes of the 2x2 scatter matrix;
//   hull     Andrew's monotone chain, cut into chunks: every thread runs the
//            chain on its own run of 16+ consecutive candidates, survivors are
//            compacted, and the rounds repeat (chunks double when a round
//            removes less than half) until one thread finishes the rest.  The
//            orientation test is exact (floating-point filter, then an
//            expansion sum), so the vertex set does not depend on how the
//            chunks fall and equals the serial chain's.
//
// A slice of up to 8192 points is sorted and hulled inside the 160 KB LDS
// (16 B of coordinates + 2 B of index per point); longer slices run the same
// code on a global-memory workspace.
#include <algorithm>

#include "pcfit_internal.h"

namespace pcfit {

constexpr int HULL_BLOCK = 1024;
constexpr int HULL_LDS_CAP = 8192;

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
    for (int64_t i = begin; i < end; i++) {
        const IdxT v = list[i];
        const double vx = px[v], vy = py[v];
        while (k >= 2 && orient_sign(px[st[k - 2]], py[st[k - 2]], px[st[k - 1]], py[st[k - 1]], vx, vy) <= 0)
            k--;
        st[k++] = v;
    }
    return k;
}

// Monotone chain over the candidates list[0..L) -> hull chain at list[0..return)
template <class IdxT>
__device__ int64_t chain_reduce(const double *px, const double *py, IdxT *list, int64_t L, int64_t *pong,
                                int *scratch)
{
    int64_t c = 16;
    while (true) {
        if (L <= c) {                                   // one thread finishes
            __shared__ int last_k;
            if (threadIdx.x == 0)
                last_k = chain_chunk(px, py, list, L, L > 0 ? L : 1);
            __syncthreads();
            const int k = last_k;
            __syncthreads();
            return k;
        }
        while ((L + c - 1) / c > HULL_BLOCK)
            c <<= 1;
        const int kept = chain_chunk(px, py, list, L, c);
        int total;
        const int at = block_exscan(kept, scratch, &total);
        // compact through the slice's own hull output buffer (global memory)
        const int64_t begin = (int64_t)threadIdx.x * c;
        for (int i = 0; i < kept; i++)
            pong[at + i] = (int64_t)list[begin + i];
        __syncthreads();
        for (int64_t i = threadIdx.x; i < total; i += HULL_BLOCK)
            list[i] = (IdxT)pong[i];
        __syncthreads();
        if (2 * (int64_t)total > L)
            c <<= 1;
        L = total;
    }
}

template <bool IN_LDS, class IdxT>
__global__ __launch_bounds__(HULL_BLOCK) void unique_hull_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, int lds_cap,
    double *__restrict__ ux, double *__restrict__ uy, int64_t *__restrict__ n_unique, int64_t *__restrict__ hull_idx,
    int64_t *__restrict__ n_hull, int32_t *__restrict__ status, const int64_t *__restrict__ hull_start,
    double *__restrict__ wsx, double *__restrict__ wsy, uint32_t *__restrict__ wslist,
    uint32_t *__restrict__ wspark)
{
    extern __shared__ double lds_dyn[];
    __shared__ double red[9 * 16];
    __shared__ int scan_scratch[17];
    __shared__ int found_at;
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    if (IN_LDS ? (n > lds_cap) : (n <= lds_cap))
        return;                                          // the other launch takes this slice
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
    int64_t *pong = hull_idx + off;
    const int tid = threadIdx.x;
    if (tid == 0) {
        n_unique[s] = 0;
        n_hull[s] = 0;
        status[s] = PCFIT_ST_OK;
    }
    for (int64_t i = tid; i < n; i += HULL_BLOCK) {
        px[i] = x[off + i];
        py[i] = y[off + i];
    }
    __syncthreads();

    // ---- sort (ascending-only bitonic network, virtual +inf padding)
    int64_t P = 1;
    while (P < n) P <<= 1;
    for (int64_t k = 2; k <= P; k <<= 1) {
        for (int64_t d = k >> 1; d >= 1; d >>= 1) {
            const bool flip = d == (k >> 1);
            for (int64_t q = tid; q < (P >> 1); q += HULL_BLOCK) {
                int64_t i, j;
                if (flip) {
                    const int64_t blk = q / d, o = q % d;
                    i = blk * k + o;
                    j = blk * k + (k - 1 - o);
                } else {
                    i = 2 * d * (q / d) + (q % d);
                    j = i + d;
                }
                if (j < n) {
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

    // ---- unique: keep the first of every run of equal rows
    const int64_t per = (n + HULL_BLOCK - 1) / HULL_BLOCK;
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
    for (int64_t i = tid; i < m; i += HULL_BLOCK) {
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

    // ---- rank check (circle_fitting.py:14-19)
    {
        double s1[2] = {0.0, 0.0};
        for (int64_t i = tid; i < m; i += HULL_BLOCK) {
            s1[0] += px[i];
            s1[1] += py[i];
        }
        block_sum<2>(s1, red);
        const double mx = s1[0] / (double)m, my = s1[1] / (double)m;
        double sc[3] = {0.0, 0.0, 0.0};
        for (int64_t i = tid; i < m; i += HULL_BLOCK) {
            const double u = px[i] - mx, v = py[i] - my;
            sc[0] += u * u;
            sc[1] += u * v;
            sc[2] += v * v;
        }
        block_sum<3>(sc, red);
        const double phi = 0.5 * atan2(2 * sc[1], sc[0] - sc[2]);
        const double c = cos(phi), sn = sin(phi);
        double pr[2] = {0.0, 0.0};
        for (int64_t i = tid; i < m; i += HULL_BLOCK) {
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

    // ---- hull: lower chain (left to right), then upper chain (right to left)
    for (int64_t i = tid; i < m; i += HULL_BLOCK)
        list[i] = (IdxT)i;
    __syncthreads();
    const int64_t hl = chain_reduce<IdxT>(px, py, list, m, pong, scan_scratch);
    uint32_t *park = wspark + off;                      // the lower chain waits here
    for (int64_t i = tid; i < hl - 1; i += HULL_BLOCK)
        park[i] = (uint32_t)list[i];
    __syncthreads();
    for (int64_t i = tid; i < m; i += HULL_BLOCK)
        list[i] = (IdxT)(m - 1 - i);
    __syncthreads();
    const int64_t hu = chain_reduce<IdxT>(px, py, list, m, pong, scan_scratch);
    const int64_t h = (hl - 1) + (hu - 1);
    // both chains end on the other's first vertex: drop each chain's last entry
    for (int64_t i = tid; i < hl - 1; i += HULL_BLOCK)
        pong[i] = (int64_t)park[i];
    for (int64_t i = tid; i < hu - 1; i += HULL_BLOCK)
        pong[hl - 1 + i] = (int64_t)list[i];
    if (tid == 0) {
        n_hull[s] = h;
        found_at = 0x7fffffff;
    }
    __syncthreads();
    // optional rotation of the walk to a requested first vertex
    const int64_t want = hull_start ? hull_start[s] : -1;
    if (want > 0) {
        for (int64_t i = tid; i < h; i += HULL_BLOCK) {
            const int64_t v = pong[i];
            park[i] = (uint32_t)v;
            if (v == want)
                atomicMin(&found_at, (int)i);
        }
        __syncthreads();
        const int p0 = found_at;
        if (p0 != 0x7fffffff && p0 > 0) {
            for (int64_t i = tid; i < h; i += HULL_BLOCK)
                pong[i] = (int64_t)park[(i + p0) % h];
        }
    }
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API size_t pcfit_unique_hull_workspace_bytes(int64_t total_points, int64_t n_slices)
{
    (void)n_slices;
    const size_t t = (size_t)(total_points > 0 ? total_points : 1);
    return t * 8 * 2 + t * 4 * 2 + 1024;               // wsx, wsy, wslist, wspark
}

PCFIT_API int pcfit_unique_hull_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                          int64_t max_slice_len, double *ux, double *uy, int64_t *n_unique,
                                          int64_t *hull_idx, int64_t *n_hull, int32_t *status,
                                          const int64_t *hull_start, void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "unique_hull: n_slices out of range");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && ux && uy && n_unique && hull_idx && n_hull && status,
                  "unique_hull: null pointer");
    PCFIT_REQUIRE(max_slice_len >= 0 && max_slice_len < (1ll << 31),
                  "unique_hull: max_slice_len must be given (0..2^31-1)");
    // total points are not known here (offsets live on the device): the caller sized the workspace
    // with pcfit_unique_hull_workspace_bytes(total); the layout below needs total = (bytes-1024)/24
    if (!workspace || workspace_bytes < 1024 + 24) {
        set_error("unique_hull: workspace missing");
        return PCFIT_ERR_WORKSPACE;
    }
    const size_t total = (workspace_bytes - 1024) / 24;
    char *ws = static_cast<char *>(workspace);
    double *wsx = reinterpret_cast<double *>(ws);
    double *wsy = wsx + total;
    uint32_t *wslist = reinterpret_cast<uint32_t *>(wsy + total);
    uint32_t *wspark = wslist + total;
    hipStream_t st = as_stream(stream);
    int cap = (int)std::min<int64_t>(HULL_LDS_CAP, std::max<int64_t>(64, (max_slice_len + 63) / 64 * 64));
    const size_t lds = (size_t)cap * 18;
    // more than 64 KB of dynamic LDS has to be granted explicitly (per device; idempotent)
    PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&unique_hull_kernel<true, uint16_t>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, HULL_LDS_CAP * 18));
    unique_hull_kernel<true, uint16_t><<<dim3((unsigned)n_slices), dim3(HULL_BLOCK), lds, st>>>(
        x, y, offsets, cap, ux, uy, n_unique, hull_idx, n_hull, status, hull_start, wsx, wsy, wslist, wspark);
    PCFIT_LAUNCH_CHECK();
    if (max_slice_len > cap) {
        unique_hull_kernel<false, uint32_t><<<dim3((unsigned)n_slices), dim3(HULL_BLOCK), 0, st>>>(
            x, y, offsets, cap, ux, uy, n_unique, hull_idx, n_hull, status, hull_start, wsx, wsy, wslist, wspark);
        PCFIT_LAUNCH_CHECK();
    }
    return PCFIT_OK;
}
