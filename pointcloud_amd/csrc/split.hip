// Long slices spread over many workgroups (gfx950).
//
// The kernels of fit.hip and residual.hip give a slice to one workgroup, which
// fills the chip only when there are many slices.  A scan cut into a few dozen
// windows of a million points each needs the opposite: here every slice is cut
// into segments of SPLIT_SEG points, one workgroup per segment, and each pass is
// two launches - a partial pass over all segments of all slices, and a small
// per-slice kernel that combines the partial results in segment order (fixed
// order: bitwise reproducible) and does the serial work of the algorithm.
//
//   Kasa (circle_fitting.py:35-54)   sums -> means -> centred moments -> circle
//   LM   (circle_fitting.py:57-109)  one round per sweep: partial normal equations of every slice
//                                    that is still iterating, then the accept / reject / next-candidate
//                                    step of fit_math.h per slice; max_iter + 1 rounds are enqueued and
//                                    the workgroups of finished slices return at once
//   residual statistics (analyzer.py:83-92, gui.py:972-987) and the angular-bin boundary
//   (boundary.py:7-42): partial histograms / per-bin maxima merged with integer atomics
//   (order independent), extremes and first indices by a second, ordered step.
//
// All of it streams 16 bytes per point and pass from HBM or the Infinity Cache.
#include <algorithm>

#include "fit_math.h"
#include <stdlib.h>

#include "pcfit_internal.h"
#include "segments.h"

namespace pcfit {

// per slice, in the workspace
struct SplitSlice {
    LmState st;
    double tot2[32];                   // two areas of totals (12 sums + minimum)
    double mean[2];
    int32_t finished;                  // LM: result written
    int32_t pad;
};

struct SplitPlan {
    int64_t max_segs;
    size_t segstart_off, part_off, slice_off, bins_off, bytes;
};

static SplitPlan split_plan(int64_t total, int64_t S, int64_t bins)
{
    SplitPlan p;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    p.max_segs = total / SPLIT_SEG + S + 1;
    p.segstart_off = 0;
    p.part_off = up((size_t)(S + 2) * 8);          // segstart[S + 1], then the count of slices still iterating
    p.slice_off = p.part_off + up((size_t)p.max_segs * SPLIT_PART * 8);
    p.bins_off = p.slice_off + up((size_t)std::max<int64_t>(S, 1) * sizeof(SplitSlice));
    p.bytes = p.bins_off + up((size_t)std::max<int64_t>(S, 1) * (size_t)std::max<int64_t>(bins, 0) * 8);
    return p;
}

// MODE 0: sums of x, y.  MODE 1: centred moments about the slice mean.  MODE 2: LM sweep at the
// slice's trial parameters (slices that have finished cost nothing).
template <int MODE>
__global__ __launch_bounds__(SPLIT_BLOCK) void split_part_kernel(const double *__restrict__ x,
                                                                const double *__restrict__ y,
                                                                const int64_t *__restrict__ offsets, int64_t S,
                                                                const int64_t *__restrict__ segstart,
                                                                const SplitSlice *__restrict__ slices, LmParams prm,
                                                                double *__restrict__ part)
{
    __shared__ double scratch[13 * 16];
    __shared__ double tot[16];
    if constexpr (MODE == 2) {
        if (segstart[S + 1] == 0)       // every slice has finished: the rest of the enqueued rounds is idle
            return;
    }
    SegRef r;
    if (!find_segment(segstart, offsets, S, x, y, r))
        return;
    constexpr int NW = SPLIT_BLOCK / WAVE;
    double *out = part + r.seg * SPLIT_PART;
    if constexpr (MODE == 0) {
        double s1[2] = {0.0, 0.0};
        segment_points(r, [&](double px, double py, int32_t, bool valid) {
            s1[0] += valid ? px : 0.0;
            s1[1] += valid ? py : 0.0;
        });
        block_reduce_to_lds<2, false, NW>(s1, 0.0, scratch, tot);
        if (threadIdx.x < 2)
            out[threadIdx.x] = tot[threadIdx.x];
    } else if constexpr (MODE == 1) {
        const double mx = slices[r.slice].mean[0], my = slices[r.slice].mean[1];
        double m[7] = {0, 0, 0, 0, 0, 0, 0};
        segment_points(r, [&](double px, double py, int32_t, bool valid) {
            const double u = valid ? px - mx : 0.0, v = valid ? py - my : 0.0;
            const double uu = u * u, vv = v * v;
            m[0] += uu;
            m[1] += vv;
            m[2] = fma(u, v, m[2]);
            m[3] = fma(uu, u, m[3]);
            m[4] = fma(vv, v, m[4]);
            m[5] = fma(u, vv, m[5]);
            m[6] = fma(v, uu, m[6]);
        });
        block_reduce_to_lds<7, false, NW>(m, 0.0, scratch, tot);
        if (threadIdx.x < 7)
            out[threadIdx.x] = tot[threadIdx.x];
    } else {
        const LmState &st = slices[r.slice].st;
        if (st.done)
            return;
        const double px0 = st.tx, py0 = st.ty, pr = st.tr;
        const double delta = prm.delta, root_delta = sqrt(prm.delta);
        double acc[12];
#pragma unroll
        for (int k = 0; k < 12; k++)
            acc[k] = 0.0;
        double low = __longlong_as_double(0x7ff0000000000000ll);
        segment_points(r, [&](double px, double py, int32_t, bool valid) {
            lm_point_general(acc, low, px, py, px0, py0, pr, delta, root_delta, valid);
        });
        block_reduce_to_lds<12, true, NW>(acc, low, scratch, tot);
        if (threadIdx.x < 13)
            out[threadIdx.x] = tot[threadIdx.x];
    }
}

// Sum K partial values of a slice's segments in segment order: one wave per slice, lane l takes
// segments l, l + 64, ...; every lane returns the totals.
template <int K>
__device__ __forceinline__ void combine_sums(const double *__restrict__ part, int64_t seg0, int64_t nseg, double (&v)[K])
{
#pragma unroll
    for (int k = 0; k < K; k++)
        v[k] = 0.0;
    for (int64_t j = threadIdx.x; j < nseg; j += WAVE) {
#pragma unroll
        for (int k = 0; k < K; k++)
            v[k] += part[(seg0 + j) * SPLIT_PART + k];
    }
#pragma unroll
    for (int k = 0; k < K; k++)
        v[k] = wave_sum(v[k]);
}

__global__ __launch_bounds__(WAVE) void split_mean_kernel(const int64_t *__restrict__ offsets,
                                                         const int64_t *__restrict__ segstart,
                                                         const double *__restrict__ part, SplitSlice *__restrict__ slices)
{
    const int64_t s = blockIdx.x;
    const int64_t n = offsets[s + 1] - offsets[s];
    double v[2];
    combine_sums<2>(part, segstart[s], segstart[s + 1] - segstart[s], v);
    if (threadIdx.x == 0) {
        slices[s].mean[0] = n > 0 ? v[0] / (double)n : 0.0;
        slices[s].mean[1] = n > 0 ? v[1] / (double)n : 0.0;
    }
}

__global__ __launch_bounds__(WAVE) void split_kasa_kernel(const int64_t *__restrict__ offsets,
                                                         const int64_t *__restrict__ segstart,
                                                         const double *__restrict__ part,
                                                         const SplitSlice *__restrict__ slices,
                                                         double *__restrict__ cxcyr, int32_t *__restrict__ status)
{
    const int64_t s = blockIdx.x;
    const int64_t n = offsets[s + 1] - offsets[s];
    double m[7];
    combine_sums<7>(part, segstart[s], segstart[s + 1] - segstart[s], m);
    double c[3];
    int st = PCFIT_ST_SPARSE;
    c[0] = c[1] = c[2] = __longlong_as_double(0x7ff8000000000000ll);
    if (n >= 3) {
        // (suu svv suv suuu svvv suvv svuu -> the third moments as sum u z and sum v z, z = u^2 + v^2)
        const double m5[5] = {m[0], m[1], m[2], m[3] + m[5], m[4] + m[6]};
        st = kasa_from_moments(m5, slices[s].mean[0], slices[s].mean[1], (double)n, c);
    }
    if (threadIdx.x == 0) {
        cxcyr[3 * s] = c[0];
        cxcyr[3 * s + 1] = c[1];
        cxcyr[3 * s + 2] = c[2];
        if (status)
            status[s] = st;
    }
}

// LM: start every slice from init (slices with fewer than 3 points or a NaN start are finished at once,
// with init as the result, like the one-workgroup kernel after an unavailable Kasa fit)
__global__ void split_lm_init_kernel(const int64_t *__restrict__ offsets, int64_t S, const double *__restrict__ init,
                                     LmParams prm, SplitSlice *__restrict__ slices, double *__restrict__ out,
                                     int32_t *__restrict__ iters, const int32_t *__restrict__ skip_status,
                                     int64_t *__restrict__ active)
{
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= S)
        return;
    const double par[3] = {init[3 * s], init[3 * s + 1], init[3 * s + 2]};
    LmState st;
    lm_begin(st, par, prm);
    const bool skip = skip_status && skip_status[s] == PCFIT_ST_SPARSE;
    if (skip || offsets[s + 1] - offsets[s] <= 0)
        st.done = true;
    slices[s].st = st;
    slices[s].finished = 0;
    if (!st.done)
        atomicAdd(reinterpret_cast<unsigned long long *>(active), 1ull);
    if (st.done) {
        out[3 * s] = par[0];
        out[3 * s + 1] = par[1];
        out[3 * s + 2] = par[2];
        if (iters)
            for (int k = 0; k < 4; k++)
                iters[4 * s + k] = 0;
        slices[s].finished = 1;
    }
}

__global__ __launch_bounds__(WAVE) void split_lm_step_kernel(const int64_t *__restrict__ segstart,
                                                            const double *__restrict__ part, LmParams prm,
                                                            SplitSlice *__restrict__ slices, double *__restrict__ out,
                                                            int32_t *__restrict__ iters, int64_t *__restrict__ active)
{
    const int64_t s = blockIdx.x;
    if (*active == 0)
        return;
    SplitSlice &sl = slices[s];
    if (sl.finished)
        return;
    LmState st = sl.st;
    const int into = lm_into(st);
    // totals of the sweep just done: 12 sums in segment order, and the smallest distance
    const int64_t seg0 = segstart[s], nseg = segstart[s + 1] - seg0;
    double v[12];
    combine_sums<12>(part, seg0, nseg, v);
    double low = __longlong_as_double(0x7ff0000000000000ll);
    for (int64_t j = threadIdx.x; j < nseg; j += WAVE)
        low = fmin(low, part[(seg0 + j) * SPLIT_PART + 12]);
    low = wave_min(low);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++)
            sl.tot2[16 * into + k] = v[k];
        sl.tot2[16 * into + 12] = low;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lm_step(st, prm, sl.tot2, into);
        sl.st = st;
        if (st.done) {
            double par[3];
            int32_t it[4];
            lm_finish(st, par, it);
            out[3 * s] = par[0];
            out[3 * s + 1] = par[1];
            out[3 * s + 2] = par[2];
            if (iters)
                for (int k = 0; k < 4; k++)
                    iters[4 * s + k] = it[k];
            sl.finished = 1;
            atomicAdd(reinterpret_cast<unsigned long long *>(active), ~0ull);      // one slice fewer
        }
    }
}

// ---- residual statistics
struct SegExtreme {
    double rmax, rmin;
    int64_t imax, imin;                 // slice-local indices, -1: none
};
static_assert(sizeof(SegExtreme) <= SPLIT_PART * 8, "a segment's extremes fit its partial slot");

__device__ __forceinline__ void pick(SegExtreme &a, double r, int64_t i)
{
    if (a.imax < 0 ? (r == r) : (r > a.rmax || (r == a.rmax && i < a.imax))) { a.rmax = r; a.imax = i; }
    if (a.imin < 0 ? (r == r) : (r < a.rmin || (r == a.rmin && i < a.imin))) { a.rmin = r; a.imin = i; }
}

__device__ __forceinline__ void merge(SegExtreme &a, const SegExtreme &b)
{
    if (b.imax >= 0 && (a.imax < 0 || b.rmax > a.rmax || (b.rmax == a.rmax && b.imax < a.imax))) { a.rmax = b.rmax; a.imax = b.imax; }
    if (b.imin >= 0 && (a.imin < 0 || b.rmin < a.rmin || (b.rmin == a.rmin && b.imin < a.imin))) { a.rmin = b.rmin; a.imin = b.imin; }
}

__device__ __forceinline__ SegExtreme wave_merge(SegExtreme v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        SegExtreme o;
        o.rmax = __shfl_xor(v.rmax, m, WAVE);
        o.rmin = __shfl_xor(v.rmin, m, WAVE);
        o.imax = __shfl_xor(v.imax, m, WAVE);
        o.imin = __shfl_xor(v.imin, m, WAVE);
        merge(v, o);
    }
    return v;
}

// dynamic LDS: (H+1) edges as doubles, then H int32 counters.  hist must be zero on entry.
__global__ __launch_bounds__(SPLIT_BLOCK) void split_resid_part_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, const double *__restrict__ cxcyr, double *__restrict__ theta,
    double *__restrict__ dr, double *__restrict__ radius, int32_t H, const double *__restrict__ edges,
    unsigned long long *__restrict__ hist, double *__restrict__ part)
{
    extern __shared__ double lds_raw[];
    __shared__ SegExtreme wext[SPLIT_BLOCK / WAVE];
    SegRef r;
    if (!find_segment(segstart, offsets, S, x, y, r))
        return;
    double *ledge = lds_raw;
    int32_t *lhist = reinterpret_cast<int32_t *>(lds_raw + (H > 0 ? H + 1 : 0));
    for (int b = threadIdx.x; b < H; b += SPLIT_BLOCK)
        lhist[b] = 0;
    for (int b = threadIdx.x; b <= H && H > 0; b += SPLIT_BLOCK)
        ledge[b] = edges[b];
    __syncthreads();
    const int64_t s = r.slice;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1], R = cxcyr[3 * s + 2];
    const double e0 = H > 0 ? ledge[0] : 0.0, eH = H > 0 ? ledge[H] : 0.0;
    const double scale = (double)H / (eH - e0);         // (the bin rule of residual.hip)
    const int64_t out0 = offsets[s] + r.first;
    SegExtreme ext{0.0, 0.0, -1, -1};
    segment_points(r, [&](double px, double py, int32_t i, bool valid) {
        const double dx = px - cx, dy = py - cy;
        const double rr = cr_hypot(dx, dy);
        const double d = rr - R;
        if (valid) {
            if (theta) theta[out0 + i] = wrap_angle(dy, dx);
            if (dr) dr[out0 + i] = d;
            if (radius) radius[out0 + i] = rr;
            pick(ext, rr, r.first + i);
            if (H > 0 && d >= e0 && d <= eH) {
                int32_t b = (int32_t)((d - e0) * scale);
                b = b > H - 1 ? H - 1 : b;
                if (d < ledge[b]) b--;
                else if (b < H - 1 && d >= ledge[b + 1]) b++;
                atomicAdd(&lhist[b], 1);
            }
        }
    });
    ext = wave_merge(ext);
    if ((threadIdx.x & 63) == 0)
        wext[threadIdx.x >> 6] = ext;
    __syncthreads();
    for (int b = threadIdx.x; b < H; b += SPLIT_BLOCK)
        if (lhist[b])
            atomicAdd(&hist[s * H + b], (unsigned long long)lhist[b]);
    if (threadIdx.x == 0) {
        for (int w = 1; w < SPLIT_BLOCK / WAVE; w++)
            merge(ext, wext[w]);
        *reinterpret_cast<SegExtreme *>(part + r.seg * SPLIT_PART) = ext;
    }
}

__global__ __launch_bounds__(WAVE) void split_resid_comb_kernel(const int64_t *__restrict__ segstart,
                                                               const double *__restrict__ part,
                                                               double *__restrict__ rmax_rmin,
                                                               int64_t *__restrict__ argmax_argmin)
{
    const int64_t s = blockIdx.x;
    const int64_t seg0 = segstart[s], nseg = segstart[s + 1] - seg0;
    SegExtreme ext{0.0, 0.0, -1, -1};
    for (int64_t j = threadIdx.x; j < nseg; j += WAVE)
        merge(ext, *reinterpret_cast<const SegExtreme *>(part + (seg0 + j) * SPLIT_PART));
    ext = wave_merge(ext);
    if (threadIdx.x == 0) {
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        rmax_rmin[2 * s] = ext.imax >= 0 ? ext.rmax : -inf;
        rmax_rmin[2 * s + 1] = ext.imin >= 0 ? ext.rmin : inf;
        argmax_argmin[2 * s] = ext.imax;
        argmax_argmin[2 * s + 1] = ext.imin;
    }
}

// ---- angular-bin boundary: best[s][b] = bit pattern + 1 of the largest radius (0: empty), then the first
// index attaining it.  PASS 1 fills best (zero on entry), PASS 2 fills sel (INT64_MAX on entry).
template <int PASS>
__global__ __launch_bounds__(SPLIT_BLOCK) void split_amax_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, const double *__restrict__ cxcyr, int32_t bins,
    unsigned long long *__restrict__ best, long long *__restrict__ sel)
{
    extern __shared__ double lds_raw[];
    unsigned long long *lbest = reinterpret_cast<unsigned long long *>(lds_raw);
    SegRef r;
    if (!find_segment(segstart, offsets, S, x, y, r))
        return;
    const int64_t s = r.slice;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1];
    const double width = TWO_PI / (double)bins;
    auto bin_of = [&](double dx, double dy) -> int32_t {
        const double fb = floor(wrap_angle(dy, dx) / width);
        return fb < 0.0 ? 0 : (fb > (double)(bins - 1) ? bins - 1 : (int32_t)fb);
    };
    // (best holds bins + 1 values per slice: the bins, then the smallest of them)
    if constexpr (PASS == 1) {
        for (int b = threadIdx.x; b < bins; b += SPLIT_BLOCK)
            lbest[b] = 0ull;
    } else {
        for (int b = threadIdx.x; b <= bins; b += SPLIT_BLOCK)
            lbest[b] = best[s * (bins + 1) + b];
    }
    __syncthreads();
    segment_points(r, [&](double px, double py, int32_t i, bool valid) {
        const double dx = px - cx, dy = py - cy;
        const double rr = cr_hypot(dx, dy);
        if (valid && rr == rr && dx == dx && dy == dy) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(rr) + 1ull;
            if constexpr (PASS == 1) {
                atomicMax(&lbest[bin_of(dx, dy)], key);
            } else {
                // (a point closer in than the smallest of the slice's per-bin maxima is no bin's best:
                // most points stop here, before the arctangent)
                if (key >= lbest[bins]) {
                    const int32_t b = bin_of(dx, dy);
                    if (key == lbest[b])
                        atomicMin(&sel[s * bins + b], (long long)(r.first + i));
                }
            }
        }
    });
    if constexpr (PASS == 1) {
        __syncthreads();
        for (int b = threadIdx.x; b < bins; b += SPLIT_BLOCK)
            if (lbest[b])
                atomicMax(&best[s * (bins + 1) + b], lbest[b]);
    }
}

// best[s][bins] = the smallest of the slice's non-empty per-bin maxima
__global__ __launch_bounds__(SPLIT_BLOCK) void split_amax_floor_kernel(int32_t bins, unsigned long long *__restrict__ best)
{
    __shared__ unsigned long long wmin[SPLIT_BLOCK / WAVE];
    const int64_t s = blockIdx.x;
    unsigned long long m = ~0ull;
    for (int b = threadIdx.x; b < bins; b += SPLIT_BLOCK) {
        const unsigned long long v = best[s * (bins + 1) + b];
        m = v && v < m ? v : m;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const unsigned long long o = __shfl_xor(m, k, WAVE);
        m = o < m ? o : m;
    }
    if ((threadIdx.x & 63) == 0)
        wmin[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SPLIT_BLOCK / WAVE; w++)
            m = wmin[w] < m ? wmin[w] : m;
        best[s * (bins + 1) + bins] = m;
    }
}

__global__ void split_amax_finish_kernel(int64_t count, long long *__restrict__ sel)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < count && sel[i] == 0x7fffffffffffffffll)
        sel[i] = -1;
}

__global__ void split_fill_kernel(int64_t count, long long *__restrict__ p, long long v)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < count)
        p[i] = v;
}

struct SplitWs {
    SplitPlan plan;
    int64_t *segstart;
    double *part;
    SplitSlice *slices;
    unsigned long long *bins;
};

static int split_ws(void *workspace, size_t bytes, int64_t total, int64_t S, int64_t bins, SplitWs &w)
{
    w.plan = split_plan(total, S, bins);
    if (!workspace || bytes < w.plan.bytes) {
        set_error("split: workspace of %zu bytes needed (pcfit_split_workspace_bytes)", w.plan.bytes);
        return PCFIT_ERR_WORKSPACE;
    }
    char *p = static_cast<char *>(workspace);
    w.segstart = reinterpret_cast<int64_t *>(p + w.plan.segstart_off);
    w.part = reinterpret_cast<double *>(p + w.plan.part_off);
    w.slices = reinterpret_cast<SplitSlice *>(p + w.plan.slice_off);
    w.bins = reinterpret_cast<unsigned long long *>(p + w.plan.bins_off);
    return PCFIT_OK;
}

static int split_common(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t total,
                        const char *what)
{
    if (S < 0 || S >= (1ll << 31) || total < 0 || total / SPLIT_SEG + S + 1 >= (1ll << 31)) {
        set_error("%s: sizes out of range", what);
        return PCFIT_ERR_INVALID;
    }
    if (S > 0 && (!x || !y || !offsets)) {
        set_error("%s: null pointer", what);
        return PCFIT_ERR_INVALID;
    }
    return PCFIT_OK;
}

int split_kasa(const double *x, const double *y, const int64_t *offsets, int64_t S, const SplitWs &w, double *cxcyr,
               int32_t *status, hipStream_t st)
{
    const LmParams none{};
    const dim3 grid((unsigned)w.plan.max_segs);
    split_map_kernel<<<dim3(1), dim3(1024), 0, st>>>(offsets, S, w.segstart);
    split_part_kernel<0><<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, S, w.segstart, w.slices, none, w.part);
    split_mean_kernel<<<dim3((unsigned)S), dim3(WAVE), 0, st>>>(offsets, w.segstart, w.part, w.slices);
    split_part_kernel<1><<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, S, w.segstart, w.slices, none, w.part);
    split_kasa_kernel<<<dim3((unsigned)S), dim3(WAVE), 0, st>>>(offsets, w.segstart, w.part, w.slices, cxcyr, status);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int split_lm(const double *x, const double *y, const int64_t *offsets, int64_t S, const SplitWs &w, const double *init,
             const LmParams &prm, double *out, int32_t *iters, const int32_t *skip_status, hipStream_t st)
{
    const dim3 grid((unsigned)w.plan.max_segs);
    split_map_kernel<<<dim3(1), dim3(1024), 0, st>>>(offsets, S, w.segstart);
    int64_t *active = w.segstart + S + 1;
    PCFIT_HIP(hipMemsetAsync(active, 0, 8, st));
    split_lm_init_kernel<<<dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st>>>(offsets, S, init, prm, w.slices, out,
                                                                                   iters, skip_status, active);
    // One first sweep and at most max_iter candidate sweeps; once every slice has finished a round is two launches
    // that return at once. The rounds go out in batches of LM_BATCH. After each batch the number of unfinished
    // slices is copied to pinned host memory, and before batch k is enqueued the host looks at the count left by
    // batch k-2 (batch k-1 keeps the GPU busy meanwhile): zero ends the loop, so at most two batches of idle rounds
    // are launched instead of max_iter minus the rounds needed (typically 100 minus 10-25; about 8 us a round).
    // Under stream capture nothing can be read back, and every round is enqueued.
    constexpr int32_t LM_BATCH = 8;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    static const int allowed = [] { const char *e = getenv("PCFIT_LM_ADAPTIVE"); return e ? atoi(e) : 1; }();
    const bool adaptive = allowed && cap == hipStreamCaptureStatusNone && prm.max_iter + 1 > 3 * LM_BATCH;
    int64_t *seen = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (adaptive) {
        static thread_local int64_t *pinned = nullptr;               // two counters, kept for the thread's life
        if (!pinned)
            PCFIT_HIP(hipHostMalloc(reinterpret_cast<void **>(&pinned), 2 * sizeof(int64_t), hipHostMallocPortable));
        seen = pinned;
        PCFIT_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        const hipError_t e1 = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
        if (e1 != hipSuccess) {
            (void)hipEventDestroy(ev[0]);
            PCFIT_HIP(e1);
        }
    }
    int rc = PCFIT_OK;
    int32_t round = 0;
    int batch = 0;
    for (; round <= prm.max_iter && rc == PCFIT_OK; batch++) {
        if (adaptive && batch >= 2) {
            if (hipEventSynchronize(ev[batch & 1]) != hipSuccess) {
                set_error("lm_split: waiting for a batch of rounds failed: %s", hipGetErrorString(hipGetLastError()));
                rc = PCFIT_ERR_HIP;
                break;
            }
            if (seen[batch & 1] == 0)
                break;
        }
        const int32_t last = adaptive ? std::min<int32_t>(prm.max_iter, round + LM_BATCH - 1) : prm.max_iter;
        for (; round <= last; round++) {
            split_part_kernel<2><<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, S, w.segstart, w.slices, prm,
                                                                     w.part);
            split_lm_step_kernel<<<dim3((unsigned)S), dim3(WAVE), 0, st>>>(w.segstart, w.part, prm, w.slices, out,
                                                                           iters, active);
        }
        if (adaptive && (hipMemcpyAsync(&seen[batch & 1], active, sizeof(int64_t), hipMemcpyDeviceToHost, st) !=
                             hipSuccess ||
                         hipEventRecord(ev[batch & 1], st) != hipSuccess)) {
            set_error("lm_split: reading the count of unfinished slices failed: %s",
                      hipGetErrorString(hipGetLastError()));
            rc = PCFIT_ERR_HIP;
        }
    }
    if (adaptive) {
        // the counters belong to the thread, not to the call: the last copy into them has to land before another
        // call may use them (what is still queued by then is at most one batch of rounds)
        if (batch > 0 && hipEventSynchronize(ev[(batch - 1) & 1]) != hipSuccess && rc == PCFIT_OK) {
            set_error("lm_split: waiting for the last batch of rounds failed: %s",
                      hipGetErrorString(hipGetLastError()));
            rc = PCFIT_ERR_HIP;
        }
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
    }
    if (rc) return rc;
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API size_t pcfit_split_workspace_bytes(int64_t total_points, int64_t n_slices, int32_t n_bins)
{
    return split_plan(std::max<int64_t>(total_points, 0), std::max<int64_t>(n_slices, 0), n_bins > 0 ? n_bins + 1 : 0)
        .bytes;
}

PCFIT_API int pcfit_kasa_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                   int64_t total_points, double *cxcyr, int32_t *status, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    PCFIT_TRY(split_common(x, y, offsets, n_slices, total_points, "kasa_split"));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr, "kasa_split: null pointer");
    SplitWs w;
    PCFIT_TRY(split_ws(workspace, workspace_bytes, total_points, n_slices, 0, w));
    return split_kasa(x, y, offsets, n_slices, w, cxcyr, status, as_stream(stream));
}

PCFIT_API int pcfit_lm_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                 int64_t total_points, const double *init, double huber_delta, int32_t max_iter,
                                 double lm_lambda0, double tol_update, double tol_cost, double *out, int32_t *iters,
                                 void *workspace, size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    PCFIT_TRY(split_common(x, y, offsets, n_slices, total_points, "lm_split"));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(init && out, "lm_split: null pointer");
    PCFIT_REQUIRE(max_iter >= 0 && max_iter <= 10000, "lm_split: max_iter must be 0..10000");
    SplitWs w;
    PCFIT_TRY(split_ws(workspace, workspace_bytes, total_points, n_slices, 0, w));
    const LmParams prm{huber_delta, lm_lambda0, tol_update, tol_cost, max_iter};
    return split_lm(x, y, offsets, n_slices, w, init, prm, out, iters, nullptr, as_stream(stream));
}

PCFIT_API int pcfit_fit_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                  int64_t total_points, double huber_delta, int32_t max_iter, double lm_lambda0,
                                  double tol_update, double tol_cost, double *kasa_out, double *out, int32_t *iters,
                                  int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    PCFIT_TRY(split_common(x, y, offsets, n_slices, total_points, "fit_split"));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(kasa_out && out && status, "fit_split: kasa_out, out and status are required");
    PCFIT_REQUIRE(max_iter >= 0 && max_iter <= 10000, "fit_split: max_iter must be 0..10000");
    SplitWs w;
    PCFIT_TRY(split_ws(workspace, workspace_bytes, total_points, n_slices, 0, w));
    const LmParams prm{huber_delta, lm_lambda0, tol_update, tol_cost, max_iter};
    PCFIT_TRY(split_kasa(x, y, offsets, n_slices, w, kasa_out, status, as_stream(stream)));
    return split_lm(x, y, offsets, n_slices, w, kasa_out, prm, out, iters, status, as_stream(stream));
}

PCFIT_API int pcfit_residuals_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                        int64_t total_points, const double *cxcyr, double *theta, double *dr,
                                        double *radius, int32_t n_bins, const double *edges, int64_t *hist,
                                        double *rmax_rmin, int64_t *argmax_argmin, void *workspace,
                                        size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    PCFIT_TRY(split_common(x, y, offsets, n_slices, total_points, "residuals_split"));
    PCFIT_REQUIRE(n_bins >= 0 && n_bins <= 8192, "residuals_split: n_bins must be 0..8192");
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr && rmax_rmin && argmax_argmin, "residuals_split: null pointer");
    PCFIT_REQUIRE(n_bins == 0 || (edges && hist), "residuals_split: histogram asked for without edges/hist");
    SplitWs w;
    PCFIT_TRY(split_ws(workspace, workspace_bytes, total_points, n_slices, 0, w));
    hipStream_t st = as_stream(stream);
    if (n_bins > 0)
        PCFIT_HIP(hipMemsetAsync(hist, 0, (size_t)n_slices * n_bins * 8, st));
    const size_t lds = n_bins > 0 ? (size_t)(n_bins + 1) * 8 + (size_t)n_bins * 4 : 0;
    split_map_kernel<<<dim3(1), dim3(1024), 0, st>>>(offsets, n_slices, w.segstart);
    split_resid_part_kernel<<<dim3((unsigned)w.plan.max_segs), dim3(SPLIT_BLOCK), lds, st>>>(
        x, y, offsets, n_slices, w.segstart, cxcyr, theta, dr, radius, n_bins, edges,
        reinterpret_cast<unsigned long long *>(hist), w.part);
    split_resid_comb_kernel<<<dim3((unsigned)n_slices), dim3(WAVE), 0, st>>>(w.segstart, w.part, rmax_rmin,
                                                                             argmax_argmin);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

PCFIT_API int pcfit_angle_max_split_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                        int64_t total_points, const double *cxcyr, int32_t bins, int64_t *sel,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    PCFIT_TRY(split_common(x, y, offsets, n_slices, total_points, "angle_max_split"));
    PCFIT_REQUIRE(bins >= 1 && bins <= 8192, "angle_max_split: bins must be 1..8192");
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr && sel, "angle_max_split: null pointer");
    SplitWs w;
    PCFIT_TRY(split_ws(workspace, workspace_bytes, total_points, n_slices, bins + 1, w));
    hipStream_t st = as_stream(stream);
    const int64_t count = n_slices * (int64_t)bins;
    PCFIT_HIP(hipMemsetAsync(w.bins, 0, (size_t)n_slices * (bins + 1) * 8, st));
    split_fill_kernel<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st>>>(
        count, reinterpret_cast<long long *>(sel), 0x7fffffffffffffffll);
    split_map_kernel<<<dim3(1), dim3(1024), 0, st>>>(offsets, n_slices, w.segstart);
    const dim3 grid((unsigned)w.plan.max_segs);
    const size_t lds = (size_t)(bins + 1) * 8;
    split_amax_kernel<1><<<grid, dim3(SPLIT_BLOCK), lds, st>>>(x, y, offsets, n_slices, w.segstart, cxcyr, bins, w.bins,
                                                               reinterpret_cast<long long *>(sel));
    split_amax_floor_kernel<<<dim3((unsigned)n_slices), dim3(SPLIT_BLOCK), 0, st>>>(bins, w.bins);
    split_amax_kernel<2><<<grid, dim3(SPLIT_BLOCK), lds, st>>>(x, y, offsets, n_slices, w.segstart, cxcyr, bins, w.bins,
                                                               reinterpret_cast<long long *>(sel));
    split_amax_finish_kernel<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st>>>(
        count, reinterpret_cast<long long *>(sel));
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}
