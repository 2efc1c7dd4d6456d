// Lexicographic sort of (a, b) pairs of doubles inside every LONG slice of a CSR
// batch, and the distinct pairs of a sorted slice (np.unique(xy, axis=0),
// circle_fitting.py:182-183 / 1.py:318, for slices that do not fit the LDS of
// one workgroup; argsort(theta) of boundary.py:67 for long candidate lists).
//
//   tiles    every slice is cut into tiles of 8192 pairs (the segments of
//            segments.h), each sorted by one 1024-thread workgroup inside the
//            LDS (ascending-only bitonic network);
//   merges   log2(tiles) passes; in a pass every pair of neighbouring sorted
//            runs of a slice is merged by RANK: an element of the left run goes
//            to its own index plus the number of right-run elements strictly
//            below it, an element of the right run to its index plus the number
//            of left-run elements not above it - one binary search per element,
//            no communication between threads, ties keep left before right;
//   unique   first element of every run of equal pairs, counted per tile,
//            prefix-summed per slice, then written in order.
//
// Slices of at most min_len pairs are left alone by every kernel here: they are
// somebody else's (the LDS kernels of hull.hip / pipeline.hip).
#include "pcfit_internal.h"
#include "segments.h"

namespace pcfit {

constexpr int LS_TILE = SPLIT_SEG;
constexpr int LS_SORT_BLOCK = 1024;
constexpr int LS_BLOCK = 256;

__device__ __forceinline__ bool pair_less(double ax, double ay, double bx, double by)
{
    return ax < bx || (ax == bx && ay < by);
}

__global__ __launch_bounds__(LS_SORT_BLOCK) void ls_tile_sort_kernel(
    const double *a_in, const double *b_in, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, int64_t min_len, double *__restrict__ a_out, double *__restrict__ b_out)
{
    extern __shared__ double lds_dyn[];
    SegRef r;
    if (!find_segment(segstart, offsets, S, a_in, b_in, r))
        return;
    const int64_t off = offsets[r.slice];
    if (offsets[r.slice + 1] - off <= min_len)
        return;
    double *pa = lds_dyn, *pb = lds_dyn + LS_TILE;
    const uint32_t n32 = (uint32_t)r.n, tid = threadIdx.x;
    for (uint32_t i = tid; i < n32; i += LS_SORT_BLOCK) {
        pa[i] = r.xs[i];
        pb[i] = r.ys[i];
    }
    __syncthreads();
    int lp = 0;
    while ((1u << lp) < n32)
        lp++;
    const uint32_t half = lp > 0 ? (1u << (lp - 1)) : 0u;
    for (int lk = 1; lk <= lp; lk++) {
        for (int ld = lk - 1; ld >= 0; ld--) {
            const uint32_t d = 1u << ld;
            const bool flip = ld == lk - 1;
            for (uint32_t q = tid; q < half; q += LS_SORT_BLOCK) {
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
                    const double ax = pa[i], ay = pb[i], bx = pa[j], by = pb[j];
                    if (pair_less(bx, by, ax, ay)) {
                        pa[i] = bx; pb[i] = by;
                        pa[j] = ax; pb[j] = ay;
                    }
                }
            }
            __syncthreads();
        }
    }
    double *oa = a_out + off + r.first, *ob = b_out + off + r.first;
    for (uint32_t i = tid; i < n32; i += LS_SORT_BLOCK) {
        oa[i] = pa[i];
        ob[i] = pb[i];
    }
}

// One merge pass: sorted runs of w pairs (w a multiple of the tile) -> runs of 2 w.
__global__ __launch_bounds__(LS_BLOCK) void ls_merge_kernel(
    const double *__restrict__ a_in, const double *__restrict__ b_in, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, int64_t min_len, int64_t w, double *__restrict__ a_out,
    double *__restrict__ b_out)
{
    SegRef r;
    if (!find_segment(segstart, offsets, S, a_in, b_in, r))
        return;
    const int64_t off = offsets[r.slice], len = offsets[r.slice + 1] - off;
    if (len <= min_len)
        return;
    const double *A = a_in + off, *B = b_in + off;
    double *OA = a_out + off, *OB = b_out + off;
    // the whole tile lies in one run (w is a multiple of the tile length)
    const int64_t P = (r.first / (2 * w)) * (2 * w);
    const int64_t mid = P + w < len ? P + w : len, end = P + 2 * w < len ? P + 2 * w : len;
    const bool left = r.first < mid;
    for (int32_t i = threadIdx.x; i < r.n; i += LS_BLOCK) {
        const int64_t g = r.first + i;
        const double ea = A[g], eb = B[g];
        int64_t pos;
        if (left) {
            int64_t lo = mid, hi = end;                  // right-run elements strictly below e
            while (lo < hi) {
                const int64_t m = (lo + hi) >> 1;
                if (pair_less(A[m], B[m], ea, eb)) lo = m + 1; else hi = m;
            }
            pos = g + (lo - mid);
        } else {
            int64_t lo = P, hi = mid;                    // left-run elements not above e
            while (lo < hi) {
                const int64_t m = (lo + hi) >> 1;
                if (!pair_less(ea, eb, A[m], B[m])) lo = m + 1; else hi = m;
            }
            pos = lo + (g - mid);
        }
        OA[pos] = ea;
        OB[pos] = eb;
    }
}

__device__ __forceinline__ bool ls_is_first(const double *A, const double *B, int64_t g)
{
    return g == 0 || A[g] != A[g - 1] || B[g] != B[g - 1];
}

// distinct pairs of a sorted slice: count per tile
__global__ __launch_bounds__(LS_BLOCK) void ls_unique_count_kernel(
    const double *__restrict__ a, const double *__restrict__ b, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, int64_t min_len, int64_t *__restrict__ segcount)
{
    __shared__ int wsum[LS_BLOCK / WAVE];
    SegRef r;
    if (!find_segment(segstart, offsets, S, a, b, r))
        return;
    const int64_t off = offsets[r.slice];
    if (offsets[r.slice + 1] - off <= min_len)
        return;
    const double *A = a + off, *B = b + off;
    int c = 0;
    for (int32_t i = threadIdx.x; i < r.n; i += LS_BLOCK)
        c += ls_is_first(A, B, r.first + i) ? 1 : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        c += __shfl_xor(c, m, WAVE);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < LS_BLOCK / WAVE; k++)
            t += wsum[k];
        segcount[r.seg] = t;
    }
}

// per slice: exclusive prefix of its tiles' counts (in place), total to n_unique
__global__ __launch_bounds__(WAVE) void ls_unique_scan_kernel(const int64_t *__restrict__ offsets,
                                                             const int64_t *__restrict__ segstart, int64_t min_len,
                                                             int64_t *__restrict__ segcount,
                                                             int64_t *__restrict__ n_unique)
{
    const int64_t s = blockIdx.x;
    if (offsets[s + 1] - offsets[s] <= min_len)
        return;
    const int64_t seg0 = segstart[s], nseg = segstart[s + 1] - seg0;
    const int lane = threadIdx.x;
    int64_t carry = 0;
    for (int64_t base = 0; base < nseg; base += WAVE) {
        const int64_t j = base + lane;
        const int64_t v = j < nseg ? segcount[seg0 + j] : 0;
        int64_t inc = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const int64_t o = __shfl_up(inc, d, WAVE);
            if (lane >= d)
                inc += o;
        }
        if (j < nseg)
            segcount[seg0 + j] = carry + inc - v;
        carry += __shfl(inc, WAVE - 1, WAVE);
    }
    if (lane == 0)
        n_unique[s] = carry;
}

// ... and the distinct pairs, in order, to ua / ub at the slice's own offset
__global__ __launch_bounds__(LS_BLOCK) void ls_unique_write_kernel(
    const double *__restrict__ a, const double *__restrict__ b, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, int64_t min_len, const int64_t *__restrict__ segbase,
    double *__restrict__ ua, double *__restrict__ ub)
{
    __shared__ int wsum[LS_BLOCK / WAVE];
    SegRef r;
    if (!find_segment(segstart, offsets, S, a, b, r))
        return;
    const int64_t off = offsets[r.slice];
    if (offsets[r.slice + 1] - off <= min_len)
        return;
    const double *A = a + off, *B = b + off;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t at = off + segbase[r.seg];
    for (int32_t base = 0; base < r.n; base += LS_BLOCK) {
        const int32_t i = base + (int32_t)threadIdx.x;
        const int64_t g = r.first + i;
        const bool first = i < r.n && ls_is_first(A, B, g);
        const unsigned long long bal = __ballot(first);
        if (lane == 0)
            wsum[wave] = __popcll(bal);
        __syncthreads();
        int pre = 0, all = 0;
#pragma unroll
        for (int k = 0; k < LS_BLOCK / WAVE; k++) {
            pre += k < wave ? wsum[k] : 0;
            all += wsum[k];
        }
        if (first) {
            const int64_t dst = at + pre + __popcll(bal & ((1ull << lane) - 1ull));
            ua[dst] = A[g];
            ub[dst] = B[g];
        }
        at += all;
        __syncthreads();
    }
}

size_t long_sort_workspace_bytes(int64_t total, int64_t S)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int64_t max_segs = total / SPLIT_SEG + S + 1;
    return up((size_t)(S + 2) * 8) + up((size_t)max_segs * 8);
}

// Sorts the pairs (in_a, in_b) of every slice longer than min_len.  (a0, b0) and (a1, b1): two buffer pairs
// of `total` doubles each, overwritten at the long slices' positions; in_a / in_b may be a1 / b1 (a tile is
// read whole before its sorted form is written, and that goes to a0 / b0).  *in_second: 1 when the sorted
// pairs ended up in a1 / b1, 0 when in a0 / b0.  small_ws: long_sort_workspace_bytes(total, S) bytes;
// afterwards it holds the tile table that long_unique_dev wants.
int long_sort_dev(const double *in_a, const double *in_b, double *a0, double *b0, double *a1, double *b1,
                  const int64_t *offsets, int64_t S, int64_t total, int64_t max_len, int64_t min_len, void *small_ws,
                  int *in_second, hipStream_t st)
{
    *in_second = 0;
    if (S <= 0 || max_len <= min_len)
        return PCFIT_OK;
    int64_t *segstart = static_cast<int64_t *>(small_ws);
    const int64_t max_segs = total / SPLIT_SEG + S + 1;
    PCFIT_REQUIRE(max_segs < (1ll << 31), "long sort: too many tiles");
    const dim3 grid((unsigned)max_segs);
    split_map_kernel<<<dim3(1), dim3(1024), 0, st>>>(offsets, S, segstart);
    PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ls_tile_sort_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LS_TILE * 16));
    ls_tile_sort_kernel<<<grid, dim3(LS_SORT_BLOCK), (size_t)LS_TILE * 16, st>>>(in_a, in_b, offsets, S, segstart,
                                                                                 min_len, a0, b0);
    PCFIT_LAUNCH_CHECK();
    double *src_a = a0, *src_b = b0, *dst_a = a1, *dst_b = b1;
    int where = 0;
    for (int64_t w = LS_TILE; w < max_len; w <<= 1) {
        ls_merge_kernel<<<grid, dim3(LS_BLOCK), 0, st>>>(src_a, src_b, offsets, S, segstart, min_len, w, dst_a, dst_b);
        PCFIT_LAUNCH_CHECK();
        std::swap(src_a, dst_a);
        std::swap(src_b, dst_b);
        where ^= 1;
    }
    *in_second = where;
    return PCFIT_OK;
}

// Distinct pairs of the sorted long slices (a, b) -> (ua, ub) at the slices' offsets, their number to
// n_unique.  small_ws as left by long_sort_dev.
int long_unique_dev(const double *a, const double *b, const int64_t *offsets, int64_t S, int64_t total,
                    int64_t max_len, int64_t min_len, void *small_ws, double *ua, double *ub, int64_t *n_unique,
                    hipStream_t st)
{
    if (S <= 0 || max_len <= min_len)
        return PCFIT_OK;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    int64_t *segstart = static_cast<int64_t *>(small_ws);
    int64_t *segcount = reinterpret_cast<int64_t *>(static_cast<char *>(small_ws) + up((size_t)(S + 2) * 8));
    const int64_t max_segs = total / SPLIT_SEG + S + 1;
    const dim3 grid((unsigned)max_segs);
    ls_unique_count_kernel<<<grid, dim3(LS_BLOCK), 0, st>>>(a, b, offsets, S, segstart, min_len, segcount);
    ls_unique_scan_kernel<<<dim3((unsigned)S), dim3(WAVE), 0, st>>>(offsets, segstart, min_len, segcount, n_unique);
    ls_unique_write_kernel<<<grid, dim3(LS_BLOCK), 0, st>>>(a, b, offsets, S, segstart, min_len, segcount, ua, ub);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit
