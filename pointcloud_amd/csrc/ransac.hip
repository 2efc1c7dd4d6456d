// Seeded 3-point RANSAC for a batch of slices (_ransac_circle,
// circle_fitting.py:112-143) on gfx950.
//
// Mapping: one LANE per hypothesis.  A 256-thread workgroup owns 256 trials of
// one slice; every lane builds its circle from the trial's index triple
// (circle_fitting.py:22-32, the 2x2 solve as LAPACK evaluates it), then the
// workgroup walks the slice's points once.  The point is the same for all 64
// lanes of a wave, so it arrives through scalar loads and the lane's private
// counter needs no cross-lane reduction.  The inlier test
//     |sqrt(dx*dx + dy*dy) - R| <= tol
// is monotone in d2 = dx*dx + dy*dy, so each lane first finds the exact
// interval [d_lo, d_hi] of doubles d2 that pass (walking a few ulps around
// (R -/+ tol)^2 with the very sqrt and subtraction of the test) and the inner
// loop is two subtractions, two products, one sum and two compares: no sqrt,
// same bits.  A lane whose interval search does not settle falls back to the
// literal test.
//
// The best trial of a slice is max over (score, earliest trial), an order-
// independent integer maximum, so the cross-workgroup combine is one 64-bit
// atomicMax per workgroup and the result is deterministic.  A second small
// kernel rebuilds the winning circle and writes the inlier mask.
#include <mutex>

#include "pcfit_internal.h"

namespace pcfit {

constexpr int RS_BLOCK = 256;

struct Circle {
    double cx, cy, r;
    bool ok;
};

// circle_fitting.py:22-32 for the trial's three points
__device__ __forceinline__ Circle circle_from_triple(const double *__restrict__ xs, const double *__restrict__ ys,
                                                     const int32_t *__restrict__ tri)
{
    const double x1 = xs[tri[0]], y1 = ys[tri[0]];
    const double x2 = xs[tri[1]], y2 = ys[tri[1]];
    const double x3 = xs[tri[2]], y3 = ys[tri[2]];
    const double a = 2 * (x2 - x1), b = 2 * (y2 - y1);
    const double c = ((x2 * x2 + y2 * y2) - x1 * x1) - y1 * y1;
    const double d = 2 * (x3 - x2), e = 2 * (y3 - y2);
    const double f = ((x3 * x3 + y3 * y3) - x2 * x2) - y2 * y2;
    Circle out;
    const double det = solve2x2(a, b, c, d, e, f, out.cx, out.cy);
    out.ok = fabs(det) >= 1e-12;          // false for NaN as well
    out.r = cr_hypot(x1 - out.cx, y1 - out.cy);
    return out;
}

// Exact set of d2 >= 0 with |sqrt(d2) - R| <= tol, as an interval of doubles.
// Returns false when the search did not close (caller uses the literal test).
__device__ __forceinline__ bool inlier_interval(double R, double tol, double &d_lo, double &d_hi)
{
    d_lo = 1.0;
    d_hi = 0.0;                            // empty
    if (!(tol >= 0.0) || !(R >= 0.0) || !isfinite(R) || !isfinite(tol))
        return tol < 0.0 ? true : false;   // negative tolerance: nothing passes
    // largest s with (s - R) <= tol
    double s_hi = R + tol;
    if (!isfinite(s_hi))
        return false;
    bool ok = false;
    for (int k = 0; k < 8; k++) {
        if ((s_hi - R) > tol) {
            s_hi = next_down(s_hi);
        } else if ((next_up(s_hi) - R) <= tol) {
            s_hi = next_up(s_hi);
        } else {
            ok = true;
            break;
        }
    }
    if (!ok)
        return false;
    // smallest s >= 0 with (s - R) >= -tol
    double s_lo = R - tol;
    if (s_lo <= 0.0) {
        s_lo = 0.0;
    } else {
        ok = false;
        for (int k = 0; k < 8; k++) {
            if ((s_lo - R) < -tol) {
                s_lo = next_up(s_lo);
            } else if (s_lo > 0.0 && (next_down(s_lo) - R) >= -tol) {
                s_lo = next_down(s_lo);
            } else {
                ok = true;
                break;
            }
        }
        if (!ok)
            return false;
    }
    // largest d with sqrt(d) <= s_hi
    double hi = s_hi * s_hi;
    if (!isfinite(hi))
        return false;
    ok = false;
    for (int k = 0; k < 12; k++) {
        if (sqrt(hi) > s_hi) {
            hi = next_down(hi);
        } else if (sqrt(next_up(hi)) <= s_hi) {
            hi = next_up(hi);
        } else {
            ok = true;
            break;
        }
    }
    if (!ok)
        return false;
    // smallest d >= 0 with sqrt(d) >= s_lo
    double lo = 0.0;
    if (s_lo > 0.0) {
        lo = s_lo * s_lo;
        ok = false;
        for (int k = 0; k < 12; k++) {
            if (sqrt(lo) < s_lo) {
                lo = next_up(lo);
            } else if (lo > 0.0 && sqrt(next_down(lo)) >= s_lo) {
                lo = next_down(lo);
            } else {
                ok = true;
                break;
            }
        }
        if (!ok)
            return false;
    }
    d_lo = lo;
    d_hi = hi;
    return true;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(v, m, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

// key = (score + 1) << 32 | (0xffffffff - trial): larger score first, then earlier trial
__global__ __launch_bounds__(RS_BLOCK) void ransac_score_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ triples, const int32_t *__restrict__ table_of_slice, int32_t trials, double tol,
    int use_range, double r_min, double r_max, int32_t tiles, unsigned long long *__restrict__ best)
{
    __shared__ unsigned long long wbest[RS_BLOCK / WAVE];
    const int64_t s = blockIdx.x / tiles;          // tiles of one slice are neighbours: its points stay in L2
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    if (n < 3)
        return;
    const double *__restrict__ xs = x + off;
    const double *__restrict__ ys = y + off;
    const int32_t t = (int32_t)(blockIdx.x % tiles) * RS_BLOCK + threadIdx.x;
    const bool live = t < trials;

    double cx = 0.0, cy = 0.0, R = 0.0, d_lo = 1.0, d_hi = 0.0;
    bool valid = false, slow = false;
    if (live) {
        const int32_t *tri = triples + ((int64_t)table_of_slice[s] * trials + t) * 3;
        const Circle c = circle_from_triple(xs, ys, tri);
        valid = c.ok && (!use_range || (r_min <= c.r && c.r <= r_max));
        if (valid) {
            cx = c.cx;
            cy = c.cy;
            R = c.r;
            slow = !inlier_interval(R, tol, d_lo, d_hi);
        }
    }
    int32_t count = 0;
    if (__any(slow)) {
        // literal test for the lanes that need it, interval test for the rest
        for (int64_t i = 0; i < n; i++) {
            const double dx = xs[i] - cx, dy = ys[i] - cy;
            const double d2 = dx * dx + dy * dy;
            const bool in = slow ? (fabs(sqrt(d2) - R) <= tol) : (d2 >= d_lo && d2 <= d_hi);
            count += in;
        }
    } else {
        // count += in, as one add-with-carry whose carry-in is the compare mask (a lane per hypothesis)
        // (the two compare masks are combined as scalars: a bool built from both would be made a
        // vector value first)
        auto tally = [&](double d2) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(d2 >= d_lo) & __builtin_amdgcn_ballot_w64(d2 <= d_hi);
            asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(count) : "s"(m) : "vcc");
        };
        const int32_t n32 = (int32_t)n;                 // slices hold fewer than 2^31 points: the score is an int32
        int32_t i = 0;
        for (; i + 8 <= n32; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const double dx = xs[i + k] - cx, dy = ys[i + k] - cy;
                const double d2 = dx * dx + dy * dy;
                tally(d2);
            }
        }
        for (; i < n32; i++) {
            const double dx = xs[i] - cx, dy = ys[i] - cy;
            const double d2 = dx * dx + dy * dy;
            tally(d2);
        }
    }
    unsigned long long key = 0;
    if (valid)
        key = ((unsigned long long)(uint32_t)(count + 1) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)t);
    key = wave_max_u64(key);
    if ((threadIdx.x & 63) == 0)
        wbest[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = wbest[0];
        for (int w = 1; w < RS_BLOCK / WAVE; w++)
            k = wbest[w] > k ? wbest[w] : k;
        if (k)
            atomicMax(best + s, k);
    }
}

__global__ __launch_bounds__(RS_BLOCK) void ransac_final_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ triples, const int32_t *__restrict__ table_of_slice, int32_t trials, double tol,
    double min_frac, const unsigned long long *__restrict__ best, double *__restrict__ model,
    int32_t *__restrict__ score, int32_t *__restrict__ best_trial, uint8_t *__restrict__ mask,
    int32_t *__restrict__ status)
{
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    const double *__restrict__ xs = x + off;
    const double *__restrict__ ys = y + off;
    const unsigned long long key = n >= 3 ? best[s] : 0ull;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    if (key == 0) {
        if (threadIdx.x == 0) {
            model[3 * s] = model[3 * s + 1] = model[3 * s + 2] = qnan;
            score[s] = -1;
            best_trial[s] = -1;
            status[s] = n < 3 ? PCFIT_ST_SPARSE : PCFIT_ST_RANSAC;
        }
        for (int64_t i = threadIdx.x; i < n; i += RS_BLOCK)
            mask[off + i] = 0;
        return;
    }
    const int32_t sc = (int32_t)(uint32_t)(key >> 32) - 1;
    const int32_t t = (int32_t)(0xffffffffu - (uint32_t)key);
    const int32_t *tri = triples + ((int64_t)table_of_slice[s] * trials + t) * 3;
    const Circle c = circle_from_triple(xs, ys, tri);
    for (int64_t i = threadIdx.x; i < n; i += RS_BLOCK) {
        const double dx = xs[i] - c.cx, dy = ys[i] - c.cy;
        mask[off + i] = (uint8_t)(fabs(sqrt(dx * dx + dy * dy) - c.r) <= tol);
    }
    if (threadIdx.x == 0) {
        model[3 * s] = c.cx;
        model[3 * s + 1] = c.cy;
        model[3 * s + 2] = c.r;
        score[s] = sc;
        best_trial[s] = t;
        int32_t need = (int32_t)(min_frac * (double)n);
        if (need < 3)
            need = 3;
        status[s] = sc < need ? PCFIT_ST_RANSAC : PCFIT_ST_OK;
    }
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API size_t pcfit_ransac_workspace_bytes(int64_t n_slices)
{
    return (size_t)(n_slices > 0 ? n_slices : 1) * sizeof(unsigned long long);
}

// Both entry points below: argument checks, then clear the per-slice best keys and run the two kernels.
static int ransac_enqueue(const double *x, const double *y, const int64_t *offsets, int64_t n_slices, double inlier_tol,
                          int32_t trials, double min_inliers_frac, const double *r_range, const int32_t *triples,
                          const int32_t *table_of_slice, double *model, int32_t *score, int32_t *best_trial,
                          uint8_t *mask, int32_t *status, unsigned long long *best, hipStream_t st)
{
    const int64_t tiles = (trials + RS_BLOCK - 1) / RS_BLOCK;
    PCFIT_HIP(hipMemsetAsync(best, 0, (size_t)n_slices * sizeof(unsigned long long), st));
    if (trials > 0) {
        dim3 grid((unsigned)(n_slices * tiles));
        hipLaunchKernelGGL(ransac_score_kernel, grid, dim3(RS_BLOCK), 0, st, x, y, offsets, triples, table_of_slice,
                           trials, inlier_tol, r_range ? 1 : 0, r_range ? r_range[0] : 0.0,
                           r_range ? r_range[1] : 0.0, (int32_t)tiles, best);
        PCFIT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ransac_final_kernel, dim3((unsigned)n_slices), dim3(RS_BLOCK), 0, st, x, y, offsets, triples,
                       table_of_slice, trials, inlier_tol, min_inliers_frac, best, model, score, best_trial, mask,
                       status);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

static int ransac_check_args(const double *x, const double *y, const int64_t *offsets, int64_t n_slices, int32_t trials,
                             const int32_t *triples, const int32_t *table_of_slice, double *model, int32_t *score,
                             int32_t *best_trial, uint8_t *mask, int32_t *status)
{
    PCFIT_REQUIRE(trials >= 0, "ransac: trials negative");
    const int64_t tiles = (trials + RS_BLOCK - 1) / RS_BLOCK;
    PCFIT_REQUIRE(n_slices >= 0 && n_slices * (tiles > 0 ? tiles : 1) < (1ll << 31),
                  "ransac: n_slices * ceil(trials/256) must stay below 2^31");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && model && score && best_trial && mask && status, "ransac: null pointer");
    PCFIT_REQUIRE(trials == 0 || (triples && table_of_slice), "ransac: null sampling table");
    return PCFIT_OK;
}

PCFIT_API int pcfit_ransac_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                     double inlier_tol, int32_t trials, double min_inliers_frac,
                                     const double *r_range, const int32_t *triples, const int32_t *table_of_slice,
                                     double *model, int32_t *score, int32_t *best_trial, uint8_t *mask,
                                     int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    PCFIT_TRY(ransac_check_args(x, y, offsets, n_slices, trials, triples, table_of_slice, model, score, best_trial,
                                mask, status));
    if (n_slices == 0)
        return PCFIT_OK;
    if (!workspace || workspace_bytes < pcfit_ransac_workspace_bytes(n_slices)) {
        set_error("ransac: workspace of %zu bytes needed", pcfit_ransac_workspace_bytes(n_slices));
        return PCFIT_ERR_WORKSPACE;
    }
    return ransac_enqueue(x, y, offsets, n_slices, inlier_tol, trials, min_inliers_frac, r_range, triples,
                          table_of_slice, model, score, best_trial, mask, status,
                          static_cast<unsigned long long *>(workspace), as_stream(stream));
}

// ---- the same call for callers that may not allocate and bring no workspace --------------------------
// The per-slice best keys (the only scratch the two kernels use) live in an array that is part of the
// library's device image: every GPU the library runs on holds its own copy at its own address until the
// process ends.  In host code the array's name is not a device address; the address on the current device
// is asked for once per device (hipGetSymbolAddress) and kept.
namespace {
constexpr int64_t RS_STATIC_SLICES = PCFIT_RANSAC_STATIC_MAX_SLICES;     // 512 KB per GPU
static __device__ __align__(256) unsigned long long scratch[RS_STATIC_SLICES];
constexpr int RS_MAX_DEVICES = 64;
std::mutex g_scratch_mu;
unsigned long long *g_scratch_ptr[RS_MAX_DEVICES] = {nullptr};

int ransac_scratch(unsigned long long **out)
{
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    PCFIT_REQUIRE(dev >= 0 && dev < RS_MAX_DEVICES, "ransac: device ordinal %d beyond %d", dev, RS_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        PCFIT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        PCFIT_REQUIRE(p != nullptr, "ransac: the scratch array has no address on device %d", dev);
        g_scratch_ptr[dev] = static_cast<unsigned long long *>(p);
    }
    *out = g_scratch_ptr[dev];
    return PCFIT_OK;
}
}  // namespace

PCFIT_API int pcfit_ransac_batch_static_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                            double inlier_tol, int32_t trials, double min_inliers_frac,
                                            const double *r_range, const int32_t *triples,
                                            const int32_t *table_of_slice, double *model, int32_t *score,
                                            int32_t *best_trial, uint8_t *mask, int32_t *status, void *stream)
{
    PCFIT_ENTER();
    PCFIT_TRY(ransac_check_args(x, y, offsets, n_slices, trials, triples, table_of_slice, model, score, best_trial,
                                mask, status));
    if (n_slices == 0)
        return PCFIT_OK;
    if (n_slices > RS_STATIC_SLICES) {
        set_error("ransac (static scratch): at most %lld slices per call, got %lld; use pcfit_ransac_batch_dev with a "
                  "workspace", (long long)RS_STATIC_SLICES, (long long)n_slices);
        return PCFIT_ERR_UNSUPPORTED;
    }
    unsigned long long *best = nullptr;
    PCFIT_TRY(ransac_scratch(&best));
    return ransac_enqueue(x, y, offsets, n_slices, inlier_tol, trials, min_inliers_frac, r_range, triples,
                          table_of_slice, model, score, best_trial, mask, status, best, as_stream(stream));
}
