// Per-point residual pass and angular-bin boundary, one workgroup per slice
// (gfx950).  Both are a single sweep over the slice's points: HBM bound
// (16 bytes of x, y read per point; theta / dr written only when asked for).
//   residuals_kernel  - theta, dr = hypot - R, histogram of dr, Rmax / Rmin and
//                       where they occur   (gui.py:972-987, analyzer.py:83-92)
//   angle_max_kernel  - boundary_by_angle_max, boundary.py:7-42
#include <stdlib.h>

#include "pcfit_internal.h"
#include "angle_math.h"

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

// boundary_by_angle_max in one sweep (AngleBins, angle_math.h).  Dynamic LDS: angle_bins_lds_bytes(bins).
__global__ __launch_bounds__(RES_BLOCK) void angle_max_fast_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets,
    const double *__restrict__ cxcyr, int32_t bins, int64_t *__restrict__ sel)
{
    extern __shared__ double lds_raw[];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    const double *xs = x + off, *ys = y + off;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1];
    AngleBins ab;
    ab.init(lds_raw, bins, cx, cy);
    __syncthreads();
    constexpr int U = 4;
    auto quick = [&](double px, double py, double &dx, double &dy) {
        dx = px - cx;
        dy = py - cy;
        const double d2 = fma(dx, dx, dy * dy);
        return d2 * rsqrt_nr(d2);
    };
    // (a slice of 2^31 points or more - none of the reference's scans comes near - takes the exact passes)
    if (n >= (1ll << 31)) {
        angle_max_exact_passes(xs, ys, n, cx, cy, ab.width, bins, ab.best, ab.who);
        for (int b = threadIdx.x; b < bins; b += RES_BLOCK)
            sel[s * bins + b] = ab.who[b] == 0xffffffffu ? -1 : (int64_t)ab.who[b];
        return;
    }
    const uint32_t n32 = (uint32_t)n;
    // the first stretch raises bounds only
    const uint32_t n_seed = n32 < ANGLE_SEED_POINTS ? n32 : ANGLE_SEED_POINTS;
    for (uint32_t i = threadIdx.x; i < n_seed; i += RES_BLOCK) {
        double dx, dy;
        const double r1 = quick(xs[i], ys[i], dx, dy);
        ab.seed(dx, dy, r1);
    }
    __syncthreads();
    // (uniform trip count; the loads of a trip are issued together, past the end they read the last point again)
    const uint32_t last = n32 ? n32 - 1 : 0;
    for (uint32_t base = threadIdx.x; base - threadIdx.x < n32; base += U * RES_BLOCK) {
        double px[U], py[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = base + u * RES_BLOCK;
            px[u] = xs[i < last ? i : last];
            py[u] = ys[i < last ? i : last];
        }
        AngleBins::Probe pr[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            double dx, dy;
            const double r1 = quick(px[u], py[u], dx, dy);
            pr[u] = ab.probe(dx, dy, r1, base + u * RES_BLOCK < n32);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            ab.settle(pr[u], base + u * RES_BLOCK, base + u * RES_BLOCK < n32);
    }
    ab.finish(xs, ys, n, sel + s * bins);
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
        const size_t lds = angle_bins_lds_bytes(bins);
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
