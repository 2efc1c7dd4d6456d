// Cutting the slices of a batch into segments of SPLIT_SEG points, one workgroup per segment
// (shared by split.hip and hull.hip).
#pragma once

#include "pcfit_internal.h"

namespace pcfit {

constexpr int SPLIT_SEG = 8192;        // points per segment
constexpr int SPLIT_BLOCK = 256;
constexpr int SPLIT_PART = 16;         // doubles per partial result

// segstart[s] = number of segments of the slices before s (one workgroup, tiles of 1024 slices)
static __global__ __launch_bounds__(1024) void split_map_kernel(const int64_t *__restrict__ offsets, int64_t S,
                                                        int64_t *__restrict__ segstart)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < S; base += 1024) {
        const int64_t s = base + threadIdx.x;
        const int64_t v = s < S ? (offsets[s + 1] - offsets[s] + SPLIT_SEG - 1) / SPLIT_SEG : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int64_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const int64_t incl = part[threadIdx.x];
        if (s < S)
            segstart[s] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023)
            carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        segstart[S] = carry;
}

// which slice and which of its segments a workgroup works on; false: none (the grid is an upper bound)
struct SegRef {
    int64_t slice, seg, first;          // seg: global segment id; first: index of its first point in the slice
    int32_t n;                          // points in the segment
    const double *xs, *ys;
};

__device__ __forceinline__ bool find_segment(const int64_t *__restrict__ segstart, const int64_t *__restrict__ offsets,
                                             int64_t S, const double *x, const double *y, SegRef &r)
{
    const int64_t seg = blockIdx.x;
    if (seg >= segstart[S])
        return false;
    int64_t lo = 0, hi = S - 1;         // last slice with segstart[s] <= seg
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (segstart[mid] <= seg)
            lo = mid;
        else
            hi = mid - 1;
    }
    // (an empty slice shares its segstart with the next one, so the last slice with segstart <= seg is
    // the one that owns the segment)
    const int64_t off = offsets[lo], len = offsets[lo + 1] - off;
    r.slice = lo;
    r.seg = seg;
    r.first = (seg - segstart[lo]) * SPLIT_SEG;
    const int64_t left = len - r.first;
    r.n = (int32_t)(left < SPLIT_SEG ? left : SPLIT_SEG);
    r.xs = x + off + r.first;
    r.ys = y + off + r.first;
    return true;
}

// f(px, py, i, valid) over the points of a segment, four per thread and trip (i: index in the segment)
template <class F>
__device__ __forceinline__ void segment_points(const SegRef &r, F &&f)
{
    constexpr int U = 4, B = SPLIT_BLOCK;
    int32_t first = 0;
    for (; first + U * B <= r.n; first += U * B) {
        double px[U], py[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            px[u] = r.xs[first + u * B + threadIdx.x];
            py[u] = r.ys[first + u * B + threadIdx.x];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            f(px[u], py[u], first + u * B + (int32_t)threadIdx.x, true);
    }
    if (first < r.n) {
        double px[U], py[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t i = first + u * B + (int32_t)threadIdx.x;
            px[u] = r.xs[i < r.n - 1 ? i : r.n - 1];
            py[u] = r.ys[i < r.n - 1 ? i : r.n - 1];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t i = first + u * B + (int32_t)threadIdx.x;
            f(px[u], py[u], i, i < r.n);
        }
    }
}

}  // namespace pcfit
