// Per-point residual pass and angular-bin boundary, one workgroup per slice
// (gfx950).  Both are a single sweep over the slice's points: HBM bound
// (16 bytes of x, y read per point; theta / dr written only when asked for).
//   residuals_kernel  - theta, dr = hypot - R, histogram of dr, Rmax / Rmin and
//                       where they occur   (gui.py:972-987, analyzer.py:83-92)
//   angle_max_kernel  - boundary_by_angle_max, boundary.py:7-42
#include <stdlib.h>

#include "pcfit_internal.h"

namespace pcfit {

constexpr int RES_BLOCK = 256;

// fit.hip: the residual pass of the fused fit kernel about given circles (statistics only)
int launch_residual_stats(const double *x, const double *y, const int64_t *offsets, int64_t S, const double *cxcyr,
                          int32_t H, const double *edges, int64_t *hist, double *rmm, int64_t *amm, hipStream_t st);

struct Extreme {
    double r;
    int64_t i;
};

// larger r first; among equal r the lower index (numpy argmax keeps the first)
__device__ __forceinline__ Extreme pick_max(Extreme a, Extreme b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.r > a.r || (b.r == a.r && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ Extreme pick_min(Extreme a, Extreme b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.r < a.r || (b.r == a.r && b.i < a.i)) return b;
    return a;
}

template <bool MAX>
__device__ __forceinline__ Extreme wave_extreme(Extreme v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Extreme o;
        o.r = __shfl_xor(v.r, m, WAVE);
        o.i = __shfl_xor(v.i, m, WAVE);
        v = MAX ? pick_max(v, o) : pick_min(v, o);
    }
    return v;
}

// dynamic LDS: (H+1) edges as doubles, then H int32 counters
__global__ __launch_bounds__(RES_BLOCK) void residuals_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets,
    const double *__restrict__ cxcyr, double *__restrict__ theta, double *__restrict__ dr,
    double *__restrict__ radius, int32_t H,
    const double *__restrict__ edges, int64_t *__restrict__ hist, double *__restrict__ rmax_rmin,
    int64_t *__restrict__ argmax_argmin)
{
    extern __shared__ double lds_raw[];
    __shared__ Extreme wmax[RES_BLOCK / WAVE], wmin[RES_BLOCK / WAVE];
    double *ledge = lds_raw;
    int32_t *lhist = reinterpret_cast<int32_t *>(lds_raw + (H > 0 ? H + 1 : 0));
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1], R = cxcyr[3 * s + 2];
    for (int b = threadIdx.x; b < H; b += RES_BLOCK)
        lhist[b] = 0;
    for (int b = threadIdx.x; b <= H && H > 0; b += RES_BLOCK)
        ledge[b] = edges[b];
    __syncthreads();
    const double e0 = H > 0 ? ledge[0] : 0.0, eH = H > 0 ? ledge[H] : 0.0;
    // numpy.histogram puts d into the bin b with edges[b] <= d < edges[b+1] (last bin closed): it
    // guesses b by scaling and repairs the guess against the edges.  Any guess within one bin of the
    // truth gives the same b, so the guess here multiplies by H / span instead of dividing.
    const double scale = (double)H / (eH - e0);
    Extreme mx{0.0, -1}, mn{0.0, -1};
    for (int64_t i = threadIdx.x; i < n; i += RES_BLOCK) {
        const double dx = x[off + i] - cx, dy = y[off + i] - cy;
        const double r = cr_hypot(dx, dy);
        const double d = r - R;
        if (theta) theta[off + i] = wrap_angle(dy, dx);
        if (dr) dr[off + i] = d;
        if (radius) radius[off + i] = r;
        if (mx.i < 0 ? (r == r) : (r > mx.r)) { mx.r = r; mx.i = i; }
        if (mn.i < 0 ? (r == r) : (r < mn.r)) { mn.r = r; mn.i = i; }
        if (H > 0 && d >= e0 && d <= eH) {
            int32_t b = (int32_t)((d - e0) * scale);
            b = b > H - 1 ? H - 1 : b;
            if (d < ledge[b]) b--;
            else if (b < H - 1 && d >= ledge[b + 1]) b++;
            atomicAdd(&lhist[b], 1);
        }
    }
    mx = wave_extreme<true>(mx);
    mn = wave_extreme<false>(mn);
    if ((threadIdx.x & 63) == 0) {
        wmax[threadIdx.x >> 6] = mx;
        wmin[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < H; b += RES_BLOCK)
        hist[s * H + b] = (int64_t)lhist[b];
    if (threadIdx.x == 0) {
        for (int w = 1; w < RES_BLOCK / WAVE; w++) {
            mx = pick_max(mx, wmax[w]);
            mn = pick_min(mn, wmin[w]);
        }
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        rmax_rmin[2 * s] = mx.i >= 0 ? mx.r : -inf;
        rmax_rmin[2 * s + 1] = mn.i >= 0 ? mn.r : inf;
        argmax_argmin[2 * s] = mx.i;
        argmax_argmin[2 * s + 1] = mn.i;
    }
}

// the two exact passes of the angular-bin boundary over the points of one slice: largest radius per bin (r >= 0,
// so its bit pattern orders like r; +1 keeps an all-zero radius apart from "empty"), then the first index that
// attains it
__device__ __forceinline__ int32_t exact_bin(double dx, double dy, double width, int32_t bins)
{
    const double fb = floor(wrap_angle(dy, dx) / width);
    return fb < 0.0 ? 0 : (fb > (double)(bins - 1) ? bins - 1 : (int32_t)fb);
}

__device__ void angle_max_exact_passes(const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                                       double cx, double cy, double width, int32_t bins, unsigned long long *best,
                                       unsigned int *who)
{
    for (int64_t i = threadIdx.x; i < n; i += RES_BLOCK) {
        const double dx = x[i] - cx, dy = y[i] - cy;
        const double r = cr_hypot(dx, dy);
        if (r == r && dx == dx && dy == dy)
            atomicMax(&best[exact_bin(dx, dy, width, bins)], (unsigned long long)__double_as_longlong(r) + 1ull);
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += RES_BLOCK) {
        const double dx = x[i] - cx, dy = y[i] - cy;
        const double r = cr_hypot(dx, dy);
        if (r == r && dx == dx && dy == dy) {
            const int32_t b = exact_bin(dx, dy, width, bins);
            if ((unsigned long long)__double_as_longlong(r) + 1ull == best[b])
                atomicMin(&who[b], (unsigned int)i);
        }
    }
    __syncthreads();
}

// dynamic LDS: bins uint64 (bit pattern of the largest radius) + bins int32 (index)
__global__ __launch_bounds__(RES_BLOCK) void angle_max_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets,
    const double *__restrict__ cxcyr, int32_t bins, int64_t *__restrict__ sel)
{
    extern __shared__ double lds_raw[];
    unsigned long long *best = reinterpret_cast<unsigned long long *>(lds_raw);
    unsigned int *who = reinterpret_cast<unsigned int *>(best + bins);
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1];
    const double width = TWO_PI / (double)bins;
    for (int b = threadIdx.x; b < bins; b += RES_BLOCK) {
        best[b] = 0ull;
        who[b] = 0xffffffffu;
    }
    __syncthreads();
    angle_max_exact_passes(x + off, y + off, n, cx, cy, width, bins, best, who);
    for (int b = threadIdx.x; b < bins; b += RES_BLOCK)
        sel[s * bins + b] = who[b] == 0xffffffffu ? -1 : (int64_t)who[b];
}

// The angle of (dx, dy) in [0, 2 pi) to within ANGLE_EPS of what wrap_angle gives: atan(a) = a P(a^2) on [0, 1]
// (least-squares fit of degree 9 in a^2, largest error 1.03e-9 over two million samples), the octant by selects.
// Not a number when both are zero.
constexpr double ANGLE_EPS = 4e-9;
__device__ __forceinline__ double quick_angle(double dx, double dy)
{
    const double ax = fabs(dx), ay = fabs(dy);
    const double big = fmax(ax, ay), small = fmin(ax, ay);
    double inv = __builtin_amdgcn_rcp(big);
    inv = fma(fma(-big, inv, 1.0), inv, inv);
    const double a = small * inv, t = a * a;
    double p = -0.0014725780479088223;
    p = fma(p, t, 0.0093928847604004759);
    p = fma(p, t, -0.028141839582183099);
    p = fma(p, t, 0.054646419595965126);
    p = fma(p, t, -0.081889499675001379);
    p = fma(p, t, 0.10868261333087868);
    p = fma(p, t, -0.14244972689331231);
    p = fma(p, t, 0.19996155516714859);
    p = fma(p, t, -0.33333164390779585);
    p = fma(p, t, 0.9999999778952402);
    double th = a * p;
    th = ay > ax ? 0.25 * TWO_PI - th : th;
    th = dx < 0.0 ? 0.5 * TWO_PI - th : th;
    th = dy < 0.0 ? TWO_PI - th : th;
    return th;
}

// boundary_by_angle_max for up to ANGLE_FAST_BINS bins without a hypot and an atan2 per point.  Pass 1 takes the
// radius from the reciprocal square root (within 4.7e-16) and the bin from quick_angle; a point whose quick angle
// comes within ANGLE_EPS of a bin's edge is left out here, every other point is in the bin it is counted in, so the
// largest quick radius of a bin (kept as a float rounded down) is a lower bound of the bin's largest radius.  Pass 2
// finds the candidates - the points left out, and those whose quick radius reaches their bin's bound - and lists
// them (one or two per bin); the exact code (correctly rounded hypot, atan2, numpy's bin rule) then runs over the
// list twice: largest radius per bin, first index that attains it.  A point that attains its bin's largest radius
// is always on the list, and everything on the list is judged exactly, so the result is that of the exact passes
// over all points.  A list that overflows (it cannot with distinct radii) sends the slice through those passes.
// Dynamic LDS: bins x (uint64 best, uint32 who, uint32 bound) + (2 bins + 256) x uint32 list.
constexpr int ANGLE_FAST_BINS = 4096;
__global__ __launch_bounds__(RES_BLOCK) void angle_max_fast_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets,
    const double *__restrict__ cxcyr, int32_t bins, int64_t *__restrict__ sel)
{
    extern __shared__ double lds_raw[];
    __shared__ unsigned int n_list;
    unsigned long long *best = reinterpret_cast<unsigned long long *>(lds_raw);
    unsigned int *who = reinterpret_cast<unsigned int *>(best + bins);
    unsigned int *bound = who + bins;
    unsigned int *list = bound + bins;
    const unsigned int cap = 2u * (unsigned int)bins + 256u;
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    const double *xs = x + off, *ys = y + off;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1];
    const double width = TWO_PI / (double)bins, inv_width = (double)bins / TWO_PI;
    const double margin = ANGLE_EPS * inv_width + 1e-9;                  // in bins
    for (int b = threadIdx.x; b < bins; b += RES_BLOCK) {
        best[b] = 0ull;
        who[b] = 0xffffffffu;
        bound[b] = 0u;
    }
    if (threadIdx.x == 0)
        n_list = 0u;
    __syncthreads();
    constexpr int U = 4;
    // quick radius and bin of point i; clear: the bin is certain.  (Negated compares: a NaN is not clear.)
    auto quick = [&](double px, double py, double &r1, int32_t &b) -> bool {
        const double dx = px - cx, dy = py - cy;
        const double d2 = fma(dx, dx, dy * dy);
        r1 = d2 * rsqrt_nr(d2);
        const double tb = quick_angle(dx, dy) * inv_width;
        const double fb = floor(tb);
        const bool clear = fabs((tb - fb) - 0.5) < 0.5 - margin && fb >= 0.0 && fb < (double)bins && r1 == r1;
        b = clear ? (int32_t)fb : 0;
        return clear;
    };
    const int64_t full = n / (U * RES_BLOCK) * (U * RES_BLOCK);
    // pass 1
    for (int64_t base = threadIdx.x; base < n; base += U * RES_BLOCK) {
        double px[U], py[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * RES_BLOCK;
            const int64_t ic = i < n ? i : n - 1;
            px[u] = xs[ic];
            py[u] = ys[ic];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double r1;
            int32_t b;
            const bool clear = quick(px[u], py[u], r1, b);
            if (clear && (base < full || base + u * RES_BLOCK < n))
                atomicMax(&bound[b], __float_as_uint(__double2float_rd(r1)) + 1u);
        }
    }
    __syncthreads();
    // pass 2
    for (int64_t base = threadIdx.x; base < n; base += U * RES_BLOCK) {
        double px[U], py[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * RES_BLOCK;
            const int64_t ic = i < n ? i : n - 1;
            px[u] = xs[ic];
            py[u] = ys[ic];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double r1;
            int32_t b;
            const bool clear = quick(px[u], py[u], r1, b);
            const unsigned int k = bound[b];
            // (bound - 1 is the float's bit pattern; an empty bin has no clear point, so nobody asks)
            const double thr = (double)__uint_as_float(k - 1u) * (1.0 - 2e-15);
            const bool cand = (!clear || !(r1 < thr)) && (base < full || base + u * RES_BLOCK < n);
            if (cand) {
                const unsigned int at = atomicAdd(&n_list, 1u);
                if (at < cap)
                    list[at] = (unsigned int)(base + u * RES_BLOCK);
            }
        }
    }
    __syncthreads();
    const unsigned int listed = n_list;
    if (listed > cap) {
        angle_max_exact_passes(xs, ys, n, cx, cy, width, bins, best, who);
    } else {
        for (unsigned int k = threadIdx.x; k < listed; k += RES_BLOCK) {
            const unsigned int i = list[k];
            const double dx = xs[i] - cx, dy = ys[i] - cy;
            const double r = cr_hypot(dx, dy);
            if (r == r && dx == dx && dy == dy)
                atomicMax(&best[exact_bin(dx, dy, width, bins)], (unsigned long long)__double_as_longlong(r) + 1ull);
        }
        __syncthreads();
        for (unsigned int k = threadIdx.x; k < listed; k += RES_BLOCK) {
            const unsigned int i = list[k];
            const double dx = xs[i] - cx, dy = ys[i] - cy;
            const double r = cr_hypot(dx, dy);
            if (r == r && dx == dx && dy == dy) {
                const int32_t b = exact_bin(dx, dy, width, bins);
                if ((unsigned long long)__double_as_longlong(r) + 1ull == best[b])
                    atomicMin(&who[b], i);
            }
        }
        __syncthreads();
    }
    for (int b = threadIdx.x; b < bins; b += RES_BLOCK)
        sel[s * bins + b] = who[b] == 0xffffffffu ? -1 : (int64_t)who[b];
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API int pcfit_residuals_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                        const double *cxcyr, double *theta, double *dr, double *radius,
                                        int32_t n_bins,
                                        const double *edges, int64_t *hist, double *rmax_rmin,
                                        int64_t *argmax_argmin, void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "residuals: n_slices out of range");
    PCFIT_REQUIRE(n_bins >= 0 && n_bins <= 8192, "residuals: n_bins must be 0..8192");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && cxcyr && rmax_rmin && argmax_argmin, "residuals: null pointer");
    PCFIT_REQUIRE(n_bins == 0 || (edges && hist), "residuals: histogram asked for without edges/hist");
    // Statistics only (histogram, extremes): the fused fit kernel's residual pass about the given circles - bins and
    // extremes from the reciprocal-square-root radius, the correctly rounded hypot only where that cannot decide,
    // the loads of two trips in flight.  (read at every call so that a test can compare the two kernels)
    const char *env = getenv("PCFIT_RESIDUALS_EXACT");
    if (!theta && !dr && !radius && n_bins <= 4096 && !(env && atoi(env)))
        return launch_residual_stats(x, y, offsets, n_slices, cxcyr, n_bins, edges, hist, rmax_rmin, argmax_argmin,
                                     as_stream(stream));
    const size_t lds = n_bins > 0 ? (size_t)(n_bins + 1) * 8 + (size_t)n_bins * 4 : 0;
    // more than 64 KB of dynamic LDS has to be granted explicitly (per device; idempotent)
    if (lds > 48 * 1024)
        PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&residuals_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(residuals_kernel, dim3((unsigned)n_slices), dim3(RES_BLOCK), lds, as_stream(stream), x, y,
                       offsets, cxcyr, theta, dr, radius, n_bins, edges, hist, rmax_rmin, argmax_argmin);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

PCFIT_API int pcfit_angle_max_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                        const double *cxcyr, int32_t bins, int64_t *sel, void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "angle_max: n_slices out of range");
    PCFIT_REQUIRE(bins >= 1 && bins <= 8192, "angle_max: bins must be 1..8192");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && cxcyr && sel, "angle_max: null pointer");
    // (read at every call so that a test can compare the two kernels)
    const char *env = getenv("PCFIT_ANGLE_MAX_EXACT");
    if (bins <= ANGLE_FAST_BINS && !(env && atoi(env))) {
        const size_t lds = (size_t)bins * 16 + ((size_t)2 * bins + 256) * 4;
        if (lds > 48 * 1024)
            PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&angle_max_fast_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(angle_max_fast_kernel, dim3((unsigned)n_slices), dim3(RES_BLOCK), lds, as_stream(stream), x,
                           y, offsets, cxcyr, bins, sel);
        PCFIT_LAUNCH_CHECK();
        return PCFIT_OK;
    }
    const size_t lds = (size_t)bins * 12;
    if (lds > 48 * 1024)
        PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&angle_max_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(angle_max_kernel, dim3((unsigned)n_slices), dim3(RES_BLOCK), lds, as_stream(stream), x, y,
                       offsets, cxcyr, bins, sel);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}
