// Ovality / k=2 Fourier amplitude (boundary.py:56-90), the small glue kernels
// of the hybrid fit (candidate gather, inlier compaction) and the host driver
// that chains unique -> hull -> RANSAC -> Kasa -> LM -> metrics for a batch of
// slices (fit_circle_hybrid_at_z, circle_fitting.py:146-232, followed by the
// per-slice metrics of analyzer.py:83-95).
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <tuple>
#include <vector>

#include "pcfit_internal.h"
#include "segments.h"

namespace pcfit {

constexpr int OV_BLOCK = 256;
constexpr int OV_CAP = 8192;

// One workgroup per slice.  Dynamic LDS: theta[cap], dev[cap] doubles, id[cap] uint16.
__global__ __launch_bounds__(OV_BLOCK) void ovality_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                          const int64_t *__restrict__ offsets,
                                                          const double *__restrict__ cxcyr, int cap,
                                                          double *__restrict__ scal, double *__restrict__ dev_uniform,
                                                          int64_t grid_len)
{
    extern __shared__ double lds_dyn[];
    __shared__ double red[4 * 16];
    double *th = lds_dyn, *dv = lds_dyn + cap;
    uint16_t *id = reinterpret_cast<uint16_t *>(lds_dyn + 2 * cap);
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1], R = cxcyr[3 * s + 2];
    const int tid = threadIdx.x;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    if (n < 1 || n > cap || !(cx == cx) || !(cy == cy) || !(R == R)) {
        if (tid < 5)
            scal[5 * s + tid] = qnan;
        if (dev_uniform)
            for (int64_t k = tid; k < grid_len; k += OV_BLOCK)
                dev_uniform[s * grid_len + k] = qnan;
        return;
    }
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double rmax = -inf, rmin = inf;
    for (int64_t i = tid; i < n; i += OV_BLOCK) {
        const double dx = x[off + i] - cx, dy = y[off + i] - cy;
        const double r = cr_hypot(dx, dy);
        th[i] = wrap_angle(dy, dx);
        dv[i] = r - R;
        id[i] = (uint16_t)i;
        rmax = fmax(rmax, r);
        rmin = fmin(rmin, r);
    }
    // max / min over the workgroup (order independent)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        rmax = fmax(rmax, __shfl_xor(rmax, m, WAVE));
        rmin = fmin(rmin, __shfl_xor(rmin, m, WAVE));
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = rmax;
        red[16 + (tid >> 6)] = rmin;
    }
    __syncthreads();
    for (int w = 0; w < OV_BLOCK / WAVE; w++) {
        rmax = fmax(rmax, red[w]);
        rmin = fmin(rmin, red[16 + w]);
    }
    // argsort by theta (ties by index): ascending-only bitonic network, virtual +inf padding
    int64_t P = 1;
    while (P < n) P <<= 1;
    for (int64_t k = 2; k <= P; k <<= 1) {
        for (int64_t d = k >> 1; d >= 1; d >>= 1) {
            const bool flip = d == (k >> 1);
            for (int64_t q = tid; q < (P >> 1); q += OV_BLOCK) {
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
                    const double a = th[i], b = th[j];
                    const uint16_t ia = id[i], ib = id[j];
                    if (b < a || (b == a && ib < ia)) {
                        const double da = dv[i], db = dv[j];
                        th[i] = b; th[j] = a;
                        dv[i] = db; dv[j] = da;
                        id[i] = ib; id[j] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    // periodic linear interpolation on K = max(1024, n) uniform angles + the k=2 DFT bin
    const int64_t K = n > 1024 ? n : 1024;
    const double step = TWO_PI / (double)K;
    double acc[2] = {0.0, 0.0};
    for (int64_t k = tid; k < K; k += OV_BLOCK) {
        const double t = (double)k * step;
        // largest j in [-1, n-1] with xp[j] <= t, xp[-1] = th[n-1] - 2pi
        int64_t lo = -1, hi = n - 1;
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo + 1) / 2;
            if (th[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const int64_t j = lo;
        const double xj = j < 0 ? th[n - 1] - TWO_PI : th[j];
        const double fj = j < 0 ? dv[n - 1] : dv[j];
        double v;
        if (xj == t) {
            v = fj;
        } else {
            const double xn = j + 1 < n ? th[j + 1] : th[0] + TWO_PI;
            const double fn = j + 1 < n ? dv[j + 1] : dv[0];
            const double slope = (fn - fj) / (xn - xj);
            v = slope * (t - xj) + fj;
        }
        if (dev_uniform)
            dev_uniform[s * grid_len + k] = v;
        const double ang = 2.0 * TWO_PI * (double)k / (double)K;
        acc[0] += v * cos(ang);
        acc[1] -= v * sin(ang);
    }
    // the rest of the row (rows are grid_len apart) holds no profile: NaN, whatever the buffer held
    if (dev_uniform)
        for (int64_t k = K + tid; k < grid_len; k += OV_BLOCK)
            dev_uniform[s * grid_len + k] = qnan;
    block_sum<2>(acc, red + 32);
    if (tid == 0) {
        scal[5 * s] = rmax;
        scal[5 * s + 1] = rmin;
        scal[5 * s + 2] = rmax - rmin;
        scal[5 * s + 3] = R > 0 ? 100.0 * (rmax - rmin) / R : qnan;
        scal[5 * s + 4] = sqrt(acc[0] * acc[0] + acc[1] * acc[1]) / ((double)K / 2.0);
    }
}

// ---- the same for candidate lists beyond the LDS (every distinct point of a long slice as candidates,
// circle_fitting.py:196-198).  theta and the point's index (as a double: exact below 2^53) are sorted as
// pairs by longsort.hip - ties by index, the rule of the kernel above -, then every tile of 8192 grid angles
// of a slice is interpolated by its own workgroup (K = n for these slices) and the partial DFT sums and
// extremes are combined per slice in tile order.
__global__ __launch_bounds__(SPLIT_BLOCK) void ov_long_prepare_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, int64_t min_len, const double *__restrict__ cxcyr,
    double *__restrict__ th, double *__restrict__ id)
{
    SegRef r;
    if (!find_segment(segstart, offsets, S, x, y, r))
        return;
    const int64_t off = offsets[r.slice];
    if (offsets[r.slice + 1] - off <= min_len)
        return;
    const double cx = cxcyr[3 * r.slice], cy = cxcyr[3 * r.slice + 1];
    for (int32_t i = threadIdx.x; i < r.n; i += SPLIT_BLOCK) {
        th[off + r.first + i] = wrap_angle(r.ys[i] - cy, r.xs[i] - cx);
        id[off + r.first + i] = (double)(r.first + i);
    }
}

// part: 4 doubles per tile = {sum v cos, -sum v sin, rmax, rmin}
__global__ __launch_bounds__(SPLIT_BLOCK) void ov_long_part_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, int64_t S,
    const int64_t *__restrict__ segstart, int64_t min_len, const double *__restrict__ cxcyr,
    const double *__restrict__ th_all, const double *__restrict__ id_all, double *__restrict__ part,
    double *__restrict__ dev_uniform, int64_t grid_len)
{
    __shared__ double red[4 * 16];
    SegRef r;
    if (!find_segment(segstart, offsets, S, x, y, r))
        return;
    const int64_t off = offsets[r.slice], n = offsets[r.slice + 1] - off;
    if (n <= min_len)
        return;
    const int64_t s = r.slice;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1], R = cxcyr[3 * s + 2];
    const double *th = th_all + off, *id = id_all + off, *xs = x + off, *ys = y + off;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    auto dev_at = [&](int64_t j) {
        const int64_t src = (int64_t)id[j];
        return cr_hypot(xs[src] - cx, ys[src] - cy) - R;
    };
    const int64_t K = n;                                   // max(1024, n) with n beyond the LDS capacity
    const double step = TWO_PI / (double)K;
    double acc[2] = {0.0, 0.0};
    double rmax = -inf, rmin = inf;
    for (int32_t i = threadIdx.x; i < r.n; i += SPLIT_BLOCK) {
        const double rr = cr_hypot(r.xs[i] - cx, r.ys[i] - cy);
        rmax = fmax(rmax, rr);
        rmin = fmin(rmin, rr);
        const int64_t k = r.first + i;
        const double t = (double)k * step;
        int64_t lo = -1, hi = n - 1;                       // largest j in [-1, n-1] with th[j] <= t
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo + 1) / 2;
            if (th[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const int64_t j = lo;
        const double xj = j < 0 ? th[n - 1] - TWO_PI : th[j];
        const double fj = j < 0 ? dev_at(n - 1) : dev_at(j);
        double v;
        if (xj == t) {
            v = fj;
        } else {
            const double xn = j + 1 < n ? th[j + 1] : th[0] + TWO_PI;
            const double fn = j + 1 < n ? dev_at(j + 1) : dev_at(0);
            const double slope = (fn - fj) / (xn - xj);
            v = slope * (t - xj) + fj;
        }
        if (dev_uniform)
            dev_uniform[s * grid_len + k] = v;
        const double ang = 2.0 * TWO_PI * (double)k / (double)K;
        acc[0] += v * cos(ang);
        acc[1] -= v * sin(ang);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        rmax = fmax(rmax, __shfl_xor(rmax, m, WAVE));
        rmin = fmin(rmin, __shfl_xor(rmin, m, WAVE));
    }
    if ((threadIdx.x & 63) == 0) {
        red[32 + (threadIdx.x >> 6)] = rmax;
        red[48 + (threadIdx.x >> 6)] = rmin;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        for (int w = 0; w < SPLIT_BLOCK / WAVE; w++) {
            rmax = fmax(rmax, red[32 + w]);
            rmin = fmin(rmin, red[48 + w]);
        }
        double *p = part + 4 * r.seg;
        p[0] = acc[0];
        p[1] = acc[1];
        p[2] = rmax;
        p[3] = rmin;
    }
}

__global__ __launch_bounds__(WAVE) void ov_long_final_kernel(const int64_t *__restrict__ offsets,
                                                            const int64_t *__restrict__ segstart, int64_t min_len,
                                                            const double *__restrict__ cxcyr,
                                                            const double *__restrict__ part, double *__restrict__ scal,
                                                            double *__restrict__ dev_uniform, int64_t grid_len)
{
    const int64_t s = blockIdx.x;
    const int64_t n = offsets[s + 1] - offsets[s];
    if (n <= min_len)
        return;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1], R = cxcyr[3 * s + 2];
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    if (dev_uniform)
        for (int64_t k = n + threadIdx.x; k < grid_len; k += WAVE)
            dev_uniform[s * grid_len + k] = qnan;
    if (!(cx == cx) || !(cy == cy) || !(R == R)) {
        if (threadIdx.x < 5)
            scal[5 * s + threadIdx.x] = qnan;
        if (dev_uniform)
            for (int64_t k = threadIdx.x; k < n; k += WAVE)
                dev_uniform[s * grid_len + k] = qnan;
        return;
    }
    if (threadIdx.x != 0)
        return;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double a0 = 0.0, a1 = 0.0, rmax = -inf, rmin = inf;
    for (int64_t g = segstart[s]; g < segstart[s + 1]; g++) {
        a0 += part[4 * g];
        a1 += part[4 * g + 1];
        rmax = fmax(rmax, part[4 * g + 2]);
        rmin = fmin(rmin, part[4 * g + 3]);
    }
    scal[5 * s] = rmax;
    scal[5 * s + 1] = rmin;
    scal[5 * s + 2] = rmax - rmin;
    scal[5 * s + 3] = R > 0 ? 100.0 * (rmax - rmin) / R : qnan;
    scal[5 * s + 4] = sqrt(a0 * a0 + a1 * a1) / ((double)n / 2.0);
}

// circle_fitting.py:14-19 for m points at px, py (global memory), evaluated by a
// 256-thread workgroup; the same value in every thread.  red: 48 doubles of LDS.
__device__ bool has_2d_variation_block(const double *__restrict__ px, const double *__restrict__ py, int64_t m,
                                       double eps, double *red)
{
    if (m < 3)
        return false;
    const int tid = threadIdx.x;
    double s1[2] = {0.0, 0.0};
    for (int64_t i = tid; i < m; i += 256) {
        s1[0] += px[i];
        s1[1] += py[i];
    }
    block_sum<2>(s1, red);
    const double mx = s1[0] / (double)m, my = s1[1] / (double)m;
    double sc[3] = {0.0, 0.0, 0.0};
    for (int64_t i = tid; i < m; i += 256) {
        const double u = px[i] - mx, v = py[i] - my;
        sc[0] += u * u;
        sc[1] += u * v;
        sc[2] += v * v;
    }
    block_sum<3>(sc, red);
    const double phi = 0.5 * atan2(2 * sc[1], sc[0] - sc[2]);
    const double c = cos(phi), sn = sin(phi);
    double pr[2] = {0.0, 0.0};
    for (int64_t i = tid; i < m; i += 256) {
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
    return s1v > eps * (s0 + eps);
}

// candidates of a slice: its hull vertices (in hull order) or all its distinct
// points; also the second rank check of circle_fitting.py:200 on them.
__global__ __launch_bounds__(256) void gather_candidates_kernel(
    const double *__restrict__ ux, const double *__restrict__ uy, const int64_t *__restrict__ offsets,
    const int64_t *__restrict__ hull_idx, const int64_t *__restrict__ cand_offsets, int use_hull,
    double *__restrict__ cand_x, double *__restrict__ cand_y, int32_t *__restrict__ status)
{
    __shared__ double red[3 * 16];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], co = cand_offsets[s], m = cand_offsets[s + 1] - co;
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < m; i += 256) {
        const int64_t src = use_hull ? hull_idx[off + i] : i;
        cand_x[co + i] = ux[off + src];
        cand_y[co + i] = uy[off + src];
    }
    if (m < 3)
        return;
    __syncthreads();
    const bool ok = has_2d_variation_block(cand_x + co, cand_y + co, m, 1e-12, red);
    if (tid == 0 && !ok)
        status[s] = PCFIT_ST_RANK;
}

__global__ __launch_bounds__(256) void rank_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                  const int64_t *__restrict__ offsets, double eps,
                                                  uint8_t *__restrict__ ok)
{
    __shared__ double red[3 * 16];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], m = offsets[s + 1] - off;
    const bool r = has_2d_variation_block(x + off, y + off, m, eps, red);
    if (threadIdx.x == 0)
        ok[s] = r ? 1 : 0;
}

// circle_fitting.py:22-32 for T independent triples {x1,y1,x2,y2,x3,y3}
__global__ void circle3_kernel(const double *__restrict__ tri, int64_t T, double *__restrict__ out,
                               uint8_t *__restrict__ ok)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= T)
        return;
    const double *p = tri + 6 * t;
    const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], x3 = p[4], y3 = p[5];
    const double a = 2 * (x2 - x1), b = 2 * (y2 - y1);
    const double c = ((x2 * x2 + y2 * y2) - x1 * x1) - y1 * y1;
    const double d = 2 * (x3 - x2), e = 2 * (y3 - y2);
    const double f = ((x3 * x3 + y3 * y3) - x2 * x2) - y2 * y2;
    double cx, cy;
    const double det = solve2x2(a, b, c, d, e, f, cx, cy);
    ok[t] = fabs(det) >= 1e-12 ? 1 : 0;
    out[3 * t] = cx;
    out[3 * t + 1] = cy;
    out[3 * t + 2] = cr_hypot(x1 - cx, y1 - cy);
}

// inliers of a slice, in candidate order (candidates[mask], circle_fitting.py:215)
__global__ __launch_bounds__(256) void compact_inliers_kernel(
    const double *__restrict__ cand_x, const double *__restrict__ cand_y, const int64_t *__restrict__ cand_offsets,
    const uint8_t *__restrict__ mask, const int64_t *__restrict__ inl_offsets, double *__restrict__ inl_x,
    double *__restrict__ inl_y)
{
    __shared__ int wsum[4];
    __shared__ int carry;
    const int64_t s = blockIdx.x;
    const int64_t co = cand_offsets[s], m = cand_offsets[s + 1] - co;
    const int64_t io = inl_offsets[s], want = inl_offsets[s + 1] - io;
    if (want == 0)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < m; base += 256) {
        const int64_t i = base + tid;
        const bool in = i < m && mask[co + i] != 0;
        const unsigned long long bal = __ballot(in);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0)
            wsum[wave] = __popcll(bal);
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; w++)
            pre += wsum[w];
        if (in) {
            inl_x[io + pre + before] = cand_x[co + i];
            inl_y[io + pre + before] = cand_y[co + i];
        }
        __syncthreads();
        if (tid == 0)
            carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

int launch_fit(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
               const struct LmParams &prm, double *kasa_out, double *out, int32_t *iters, int32_t *status,
               hipStream_t st);

}  // namespace pcfit

using namespace pcfit;

static int ovality_lds_launch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                              int64_t max_slice_len, const double *cxcyr, double *scal, double *dev_uniform,
                              int64_t grid_len, hipStream_t st)
{
    const int64_t longest = std::min<int64_t>(max_slice_len, OV_CAP);
    const int cap = (int)std::max<int64_t>(64, (longest + 63) / 64 * 64);
    PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ovality_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, OV_CAP * 18));
    ovality_kernel<<<dim3((unsigned)n_slices), dim3(OV_BLOCK), (size_t)cap * 18, st>>>(x, y, offsets, cxcyr, cap, scal,
                                                                                       dev_uniform, grid_len);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

PCFIT_API int pcfit_ovality_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                      int64_t max_slice_len, const double *cxcyr, double *scal, double *dev_uniform,
                                      int64_t grid_len, void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "ovality: n_slices out of range");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && cxcyr && scal, "ovality: null pointer");
    PCFIT_REQUIRE(max_slice_len >= 0, "ovality: max_slice_len must be given");
    if (max_slice_len > OV_CAP) {
        set_error("ovality: slices of more than %d points (got %lld) need pcfit_ovality_long_dev and its workspace",
                  OV_CAP, (long long)max_slice_len);
        return PCFIT_ERR_UNSUPPORTED;
    }
    PCFIT_REQUIRE(!dev_uniform || grid_len >= std::max<int64_t>(1024, max_slice_len),
                  "ovality: grid_len must be >= max(1024, longest slice)");
    return ovality_lds_launch(x, y, offsets, n_slices, max_slice_len, cxcyr, scal, dev_uniform, grid_len,
                              as_stream(stream));
}

namespace {
struct OvLongLayout {
    size_t b0_off, b1_off, part_off, ls_off, bytes;
    int64_t max_segs;
};
OvLongLayout ov_long_layout(int64_t total, int64_t S)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    OvLongLayout l;
    const size_t t = (size_t)std::max<int64_t>(total, 1);
    l.max_segs = total / SPLIT_SEG + S + 1;
    l.b0_off = 0;
    l.b1_off = up(t * 16);
    l.part_off = l.b1_off + up(t * 16);
    l.ls_off = l.part_off + up((size_t)l.max_segs * 32);
    l.bytes = l.ls_off + long_sort_workspace_bytes(total, S);
    return l;
}
}  // namespace

PCFIT_API size_t pcfit_ovality_workspace_bytes(int64_t total_points, int64_t n_slices)
{
    return ov_long_layout(std::max<int64_t>(total_points, 0), std::max<int64_t>(n_slices, 0)).bytes;
}

PCFIT_API int pcfit_ovality_long_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                     int64_t total_points, int64_t max_slice_len, const double *cxcyr, double *scal,
                                     double *dev_uniform, int64_t grid_len, void *workspace, size_t workspace_bytes,
                                     void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "ovality: n_slices out of range");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && cxcyr && scal, "ovality: null pointer");
    PCFIT_REQUIRE(max_slice_len >= 0 && total_points >= max_slice_len, "ovality: max_slice_len / total_points must be given");
    PCFIT_REQUIRE(!dev_uniform || grid_len >= std::max<int64_t>(1024, max_slice_len),
                  "ovality: grid_len must be >= max(1024, longest slice)");
    hipStream_t st = as_stream(stream);
    // slices that fit the LDS (the kernel writes NaN for the longer ones, which the second step replaces)
    PCFIT_TRY(ovality_lds_launch(x, y, offsets, n_slices, max_slice_len, cxcyr, scal, dev_uniform, grid_len, st));
    if (max_slice_len <= OV_CAP)
        return PCFIT_OK;
    const OvLongLayout l = ov_long_layout(total_points, n_slices);
    if (!workspace || workspace_bytes < l.bytes) {
        set_error("ovality: workspace of %zu bytes needed", l.bytes);
        return PCFIT_ERR_WORKSPACE;
    }
    PCFIT_REQUIRE(l.max_segs < (1ll << 31), "ovality: too many tiles");
    char *ws = static_cast<char *>(workspace);
    const size_t t = (size_t)total_points;
    double *b0a = reinterpret_cast<double *>(ws + l.b0_off), *b0b = b0a + t;
    double *b1a = reinterpret_cast<double *>(ws + l.b1_off), *b1b = b1a + t;
    double *part = reinterpret_cast<double *>(ws + l.part_off);
    int64_t *segstart = reinterpret_cast<int64_t *>(ws + l.ls_off);
    const dim3 grid((unsigned)l.max_segs);
    split_map_kernel<<<dim3(1), dim3(1024), 0, st>>>(offsets, n_slices, segstart);
    ov_long_prepare_kernel<<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, n_slices, segstart, OV_CAP, cxcyr, b1a,
                                                               b1b);
    PCFIT_LAUNCH_CHECK();
    int second = 0;
    PCFIT_TRY(long_sort_dev(b1a, b1b, b0a, b0b, b1a, b1b, offsets, n_slices, total_points, max_slice_len, OV_CAP,
                            ws + l.ls_off, &second, st));
    ov_long_part_kernel<<<grid, dim3(SPLIT_BLOCK), 0, st>>>(x, y, offsets, n_slices, segstart, OV_CAP, cxcyr,
                                                            second ? b1a : b0a, second ? b1b : b0b, part, dev_uniform,
                                                            grid_len);
    ov_long_final_kernel<<<dim3((unsigned)n_slices), dim3(WAVE), 0, st>>>(offsets, segstart, OV_CAP, cxcyr, part, scal,
                                                                          dev_uniform, grid_len);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

// The points of a slice on the host, sorted by (x, y) with copies left out, in place; returns how many are left.
// A counting sort on a bucket number that never decreases with x (two buckets a point over the range of x), then
// a comparison sort inside every bucket - a point or two as a rule: a quarter of the time of one comparison sort.
static int64_t sort_distinct_points(double *x, double *y, int64_t m)
{
    typedef std::pair<double, double> P;
    std::vector<P> pts((size_t)m);
    double lo = 0.0, hi = 0.0;
    bool finite = m > 0;
    for (int64_t i = 0; i < m; i++) {
        pts[(size_t)i] = {x[i], y[i]};
        finite = finite && std::isfinite(x[i]);
        lo = i == 0 || x[i] < lo ? x[i] : lo;
        hi = i == 0 || x[i] > hi ? x[i] : hi;
    }
    const double scale = finite && hi > lo ? (double)(2 * m) / (hi - lo) : 0.0;
    if (m >= 256 && m < (1ll << 30) && scale > 0.0 && std::isfinite(scale)) {
        const int64_t B = 2 * m + 1;
        auto bucket = [&](double v) { const int64_t b = (int64_t)((v - lo) * scale); return b < 0 ? 0 : b >= B ? B - 1 : b; };
        std::vector<int32_t> first((size_t)B + 1, 0);
        for (int64_t i = 0; i < m; i++) first[(size_t)bucket(pts[(size_t)i].first) + 1]++;
        for (int64_t b = 0; b < B; b++) first[(size_t)b + 1] += first[(size_t)b];
        std::vector<P> out((size_t)m);
        std::vector<int32_t> at(first.begin(), first.end() - 1);
        for (int64_t i = 0; i < m; i++) out[(size_t)at[(size_t)bucket(pts[(size_t)i].first)]++] = pts[(size_t)i];
        for (int64_t b = 0; b < B; b++)
            if (first[(size_t)b + 1] - first[(size_t)b] > 1) std::sort(out.begin() + first[(size_t)b], out.begin() + first[(size_t)b + 1]);
        pts.swap(out);
    } else
        std::sort(pts.begin(), pts.end());
    pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
    for (size_t i = 0; i < pts.size(); i++) { x[i] = pts[i].first; y[i] = pts[i].second; }
    return (int64_t)pts.size();
}

// hull_start entries of PCFIT_HULL_START_QHULL: the walks of those slices are turned to the vertex that Qhull's
// construction leaves first.  The construction is replayed on the device (qhull_dev.hip) from the points in
// (px, py): the slices' sorted distinct points (counts in n_pts / d_n_pts), or the slices as they are
// (d_n_pts null, n_pts = their lengths; the walks in dhi then index what the interior filter kept, in dux /
// duy, and the vertex is found there by its coordinates).  Slices the kernel hands back - decisions inside
// Qhull's roundoff bands, ties, capacities - have their points brought to the host and are replayed there (all
// selected slices with PCFIT_QHULL_ON_HOST=1 in the environment).
static int apply_qhull_starts_impl(const int64_t *hull_start, const int64_t *d_hull_start, const int64_t *offsets,
                                   const int64_t *n_pts, const int64_t *d_n_pts, int64_t S, int64_t total,
                                   const double *px, const double *py, const double *dux, const double *duy,
                                   const int64_t *doff, int64_t *dnh, int64_t *dhi, hipStream_t st, int64_t max_hull)
{
    bool any = false;
    for (int64_t s = 0; hull_start && s < S && !any; s++) any = hull_start[s] == PCFIT_HULL_START_QHULL;
    if (!any || total <= 0) return PCFIT_OK;
    const char *env = std::getenv("PCFIT_QHULL_ON_HOST");
    const bool on_host = env && env[0] == '1', raw = d_n_pts == nullptr;
    std::vector<int64_t> start((size_t)S, -1);
    std::vector<double> start_xy((size_t)S * 2, 0.0);
    DevBuf dstart, dxy, dscratch, dtmp, dverts, dvcount;
    PCFIT_TRY(dstart.alloc((size_t)S * 8));
    PCFIT_TRY(dxy.alloc((size_t)S * 16));
    if (max_hull < 0) {                                // (a caller that has not looked at the hulls' sizes)
        std::vector<int64_t> nh((size_t)S);
        PCFIT_HIP(hipMemcpyAsync(nh.data(), dnh, (size_t)S * 8, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipStreamSynchronize(st));
        max_hull = S ? *std::max_element(nh.begin(), nh.end()) : 0;
    }
    // The vertices Qhull ends with, per slice, as the replay lists them (the kernel for the slices it answers, the
    // host replay for the rest): the walks keep those and no others.
    const int vstride = (int)std::max<int64_t>(3, std::min<int64_t>(max_hull, (int64_t)1 << 20));
    PCFIT_TRY(dverts.alloc((size_t)S * vstride * 16));
    PCFIT_TRY(dvcount.alloc((size_t)S * 4));
    PCFIT_HIP(hipMemsetAsync(dvcount.p, 0xff, (size_t)S * 4, st));       // (-1: no list)
    if (!on_host) {
        PCFIT_TRY(dscratch.alloc((size_t)total * 48));
        PCFIT_TRY(qhull_start_dev(px, py, doff, d_n_pts, dnh, d_hull_start, S, total, dscratch.as<void>(),
                                  dstart.as<int64_t>(), dxy.as<double>(), st, max_hull, dux, duy, dhi, false,
                                  dverts.as<void>(), dvcount.as<int32_t>(), vstride));
        PCFIT_TRY(dstart.download(start.data(), (size_t)S * 8, st));
        PCFIT_HIP(hipStreamSynchronize(st));
    } else {
        for (int64_t s = 0; s < S; s++)
            if (hull_start[s] == PCFIT_HULL_START_QHULL) start[s] = QHULL_START_DEFER;
    }
    // what the device handed back: those slices' points, packed (sorted and made distinct here if they are not),
    // to the host replay
    std::vector<int64_t> which, coff(1, 0);
    for (int64_t s = 0; s < S; s++)
        if (start[s] == QHULL_START_DEFER) {
            start[s] = -1;
            if (n_pts[s] < 3) continue;
            which.push_back(s);
            coff.push_back(coff.back() + n_pts[s]);
        }
    if (!which.empty()) {
        PCFIT_TRY(dxy.download(start_xy.data(), (size_t)S * 16, st));
        // (through the thread's pinned scratch when it holds them: copies into pageable memory run at half the rate
        // and each of them blocks)
        const size_t bytes = (size_t)coff.back() * 8;
        double *pin = bytes <= ((size_t)64 << 20) ? static_cast<double *>(pinned_scratch(2 * bytes)) : nullptr;
        std::vector<double> spare(pin ? 0 : 2 * (size_t)coff.back());
        double *const hx = pin ? pin : spare.data(), *const hy = hx + coff.back();
        for (size_t k = 0; k < which.size(); k++) {
            const int64_t s = which[k];
            PCFIT_HIP(hipMemcpyAsync(hx + coff[k], px + offsets[s], (size_t)n_pts[s] * 8, hipMemcpyDeviceToHost, st));
            PCFIT_HIP(hipMemcpyAsync(hy + coff[k], py + offsets[s], (size_t)n_pts[s] * 8, hipMemcpyDeviceToHost, st));
        }
        PCFIT_HIP(hipStreamSynchronize(st));
        // (the slices go to host threads, as they do in the replay)
        std::vector<int64_t> counts(which.size());
        auto sort_slice = [&](size_t k) {
            int64_t m = coff[k + 1] - coff[k];
            if (raw) {
                m = sort_distinct_points(hx + coff[k], hy + coff[k], m);
            }
            counts[k] = m;
        };
        {
            const size_t n = which.size();
            const size_t T = raw ? std::min<size_t>(n, std::min(16u, std::max(1u, std::thread::hardware_concurrency()))) : 1;
            std::atomic<size_t> next{0};
            std::atomic<int> failed{0};
            auto work = [&]() {
                try {
                    for (size_t k; (k = next.fetch_add(1)) < n;) sort_slice(k);
                } catch (...) {
                    failed.store(1);
                }
            };
            std::vector<std::thread> pool;
            try {
                for (size_t t = 1; t < T; t++) pool.emplace_back(work);
            } catch (...) {                   // no more threads: the ones there are share the slices
            }
            work();
            for (auto &th : pool) th.join();
            if (failed.load()) throw std::bad_alloc();
        }
        std::vector<int64_t> cstart(which.size(), -1), cwalk((size_t)coff.back()), cn(which.size(), 0);
        PCFIT_TRY(qhull_start_host(hx, hy, coff.data(), counts.data(), (int64_t)which.size(), 0, nullptr,
                                   cstart.data(), nullptr, cwalk.data(), cn.data()));
        std::vector<double> vlist((size_t)2 * vstride);
        std::vector<int32_t> vn(which.size(), -1);
        for (size_t k = 0; k < which.size(); k++) {
            const int64_t s = which[k];
            start[s] = cstart[k];
            if (cstart[k] >= 0) {
                start_xy[2 * s] = hx[coff[k] + cstart[k]];
                start_xy[2 * s + 1] = hy[coff[k] + cstart[k]];
                if (cn[k] >= 3 && cn[k] <= vstride) {            // the replay's vertices, for the walk to keep
                    for (int64_t i = 0; i < cn[k]; i++) {
                        vlist[(size_t)2 * i] = hx[coff[k] + cwalk[(size_t)(coff[k] + i)]];
                        vlist[(size_t)2 * i + 1] = hy[coff[k] + cwalk[(size_t)(coff[k] + i)]];
                    }
                    PCFIT_HIP(hipMemcpy(dverts.as<char>() + (size_t)s * vstride * 16, vlist.data(), (size_t)cn[k] * 16,
                                        hipMemcpyHostToDevice));
                    vn[k] = (int32_t)cn[k];
                }
            }
        }
        // only the handed-back slices' entries changed: they go back one by one
        for (size_t k = 0; k < which.size(); k++) {
            PCFIT_HIP(hipMemcpyAsync(dxy.as<double>() + 2 * which[k], start_xy.data() + 2 * which[k], 16,
                                     hipMemcpyHostToDevice, st));
            PCFIT_HIP(hipMemcpyAsync(dvcount.as<int32_t>() + which[k], &vn[k], 4, hipMemcpyHostToDevice, st));
        }
        PCFIT_HIP(hipStreamSynchronize(st));            // (vn and start_xy: pageable, and vn ends here)
    }
    PCFIT_TRY(dstart.upload(start.data(), (size_t)S * 8, st));
    PCFIT_TRY(dtmp.alloc((size_t)total * 8));
    PCFIT_TRY(hull_rotate_dev(doff, dnh, dux, duy, dstart.as<int64_t>(), dxy.as<double>(), S, dhi, dtmp.as<int64_t>(), st,
                              dverts.p, dvcount.as<int32_t>(), vstride));
    PCFIT_HIP(hipStreamSynchronize(st));               // start (pageable) and the buffers end with this call
    return PCFIT_OK;
}

// (host containers may run out of memory: that is an error code, not an exception through the C ABI)
static int apply_qhull_starts(const int64_t *hull_start, const int64_t *d_hull_start, const int64_t *offsets,
                              const int64_t *n_pts, const int64_t *d_n_pts, int64_t S, int64_t total, const double *px,
                              const double *py, const double *dux, const double *duy, const int64_t *doff,
                              int64_t *dnh, int64_t *dhi, hipStream_t st, int64_t max_hull = -1)
{
    try {
        return apply_qhull_starts_impl(hull_start, d_hull_start, offsets, n_pts, d_n_pts, S, total, px, py, dux, duy,
                                       doff, dnh, dhi, st, max_hull);
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(st);                // nothing of this call may still be copying into freed vectors
        set_error("qhull order: %s", e.what());
        return PCFIT_ERR_INVALID;
    }
}

// ------------------------------------------------- host-pointer entry points
PCFIT_API int pcfit_unique_hull_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                      double *ux, double *uy, int64_t *n_unique, int64_t *hull_idx, int64_t *n_hull,
                                      int32_t *status, const int64_t *hull_start)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(n_unique && n_hull && status && (total == 0 || (x && y && ux && uy && hull_idx)),
                  "unique_hull: null pointer");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dux, duy, dnu, dhi, dnh, dst, dhs, dws;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dux.alloc(total * 8));
    PCFIT_TRY(duy.alloc(total * 8));
    PCFIT_TRY(dnu.alloc(n_slices * 8));
    PCFIT_TRY(dhi.alloc(total * 8));
    PCFIT_TRY(dnh.alloc(n_slices * 8));
    PCFIT_TRY(dst.alloc(n_slices * 4));
    if (hull_start) PCFIT_TRY(dhs.upload(hull_start, n_slices * 8, st));
    PCFIT_TRY(dws.alloc(pcfit_unique_hull_workspace_bytes(total, n_slices)));
    PCFIT_TRY(pcfit_unique_hull_batch_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, max_len,
                                          dux.as<double>(), duy.as<double>(), dnu.as<int64_t>(), dhi.as<int64_t>(),
                                          dnh.as<int64_t>(), dst.as<int32_t>(), dhs.as<int64_t>(), dws.p, dws.bytes,
                                          st));
    PCFIT_TRY(dnu.download(n_unique, n_slices * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    PCFIT_TRY(apply_qhull_starts(hull_start, dhs.as<int64_t>(), offsets, n_unique, dnu.as<int64_t>(), n_slices, total,
                                 dux.as<double>(), duy.as<double>(), dux.as<double>(), duy.as<double>(),
                                 doff.as<int64_t>(), dnh.as<int64_t>(), dhi.as<int64_t>(), st));
    PCFIT_TRY(dux.download(ux, total * 8, st));
    PCFIT_TRY(duy.download(uy, total * 8, st));
    PCFIT_TRY(dhi.download(hull_idx, total * 8, st));
    PCFIT_TRY(dnh.download(n_hull, n_slices * 8, st));
    PCFIT_TRY(dst.download(status, n_slices * 4, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_ovality_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                  const double *cxcyr, double *scal, double *dev_uniform, int64_t grid_len)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr && scal && (total == 0 || (x && y)), "ovality: null pointer");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dfit, dsc, ddev;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dfit.upload(cxcyr, n_slices * 24, st));
    PCFIT_TRY(dsc.alloc(n_slices * 40));
    if (dev_uniform) PCFIT_TRY(ddev.alloc((size_t)n_slices * grid_len * 8));
    DevBuf dws;
    if (max_len > OV_CAP) PCFIT_TRY(dws.alloc(pcfit_ovality_workspace_bytes(total, n_slices)));
    PCFIT_TRY(pcfit_ovality_long_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, total, max_len,
                                     dfit.as<double>(), dsc.as<double>(), ddev.as<double>(), grid_len, dws.p,
                                     dws.bytes, st));
    PCFIT_TRY(dsc.download(scal, n_slices * 40, st));
    if (dev_uniform) PCFIT_TRY(ddev.download(dev_uniform, (size_t)n_slices * grid_len * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

// Sampling tables for the candidate counts in `sizes` (each trials x 3), one after the other in `out`.
// A table depends on the seed state, the count and the number of trials only (every slice's RANSAC starts
// its own default_rng(seed), circle_fitting.py:124), and drawing one is a serial walk of the generator:
// tables are kept between calls (a bounded, mutex-guarded cache) and the missing ones are drawn by
// several host threads at once.
namespace {
struct TableKey {
    uint64_t st[4];
    int64_t n;
    int32_t trials;
    bool operator<(const TableKey &o) const
    {
        return std::tie(st[0], st[1], st[2], st[3], n, trials) < std::tie(o.st[0], o.st[1], o.st[2], o.st[3], o.n, o.trials);
    }
};
std::mutex g_tab_mu;
std::map<TableKey, std::shared_ptr<const std::vector<int32_t>>> g_tab;
size_t g_tab_bytes = 0;
constexpr size_t TAB_CACHE_BYTES = (size_t)256 << 20;
}  // namespace

static int sampling_tables(const uint64_t state[4], const std::vector<int64_t> &sizes, int32_t trials,
                           std::vector<int32_t> &out)
{
    const size_t K = sizes.size(), per = (size_t)trials * 3;
    std::vector<std::shared_ptr<const std::vector<int32_t>>> tab(K);
    std::vector<size_t> missing;
    {
        std::lock_guard<std::mutex> lock(g_tab_mu);
        for (size_t k = 0; k < K; k++) {
            const TableKey key{{state[0], state[1], state[2], state[3]}, sizes[k], trials};
            auto it = g_tab.find(key);
            if (it != g_tab.end())
                tab[k] = it->second;
            else
                missing.push_back(k);
        }
    }
    if (!missing.empty()) {
        std::vector<int> rc(missing.size(), PCFIT_OK);
        std::vector<std::shared_ptr<std::vector<int32_t>>> fresh(missing.size());
        const int T = (int)std::min<size_t>(missing.size(), std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
        auto work = [&](int t) {
            for (size_t j = t; j < missing.size(); j += T) {
                fresh[j] = std::make_shared<std::vector<int32_t>>(per);
                rc[j] = pcfit_choice3_table(state, sizes[missing[j]], trials, fresh[j]->data());
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < T; t++)
            pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool)
            th.join();
        for (int r : rc)
            if (r != PCFIT_OK)
                return r;                   // (the message was set by the thread that failed; report the code)
        std::lock_guard<std::mutex> lock(g_tab_mu);
        if (g_tab_bytes + missing.size() * per * 4 > TAB_CACHE_BYTES) {
            g_tab.clear();
            g_tab_bytes = 0;
        }
        for (size_t j = 0; j < missing.size(); j++) {
            tab[missing[j]] = fresh[j];
            if (per * 4 <= TAB_CACHE_BYTES / 4) {
                const TableKey key{{state[0], state[1], state[2], state[3]}, sizes[missing[j]], trials};
                if (g_tab.emplace(key, fresh[j]).second)
                    g_tab_bytes += per * 4;
            }
        }
    }
    out.resize(K * per);
    for (size_t k = 0; k < K; k++)
        if (per)
            memcpy(out.data() + k * per, tab[k]->data(), per * 4);
    return PCFIT_OK;
}

// The chain on slices that are already on the device (dxp, dyp, doffp; offsets = the same CSR offsets on
// the host).  ux / uy may be null (then the distinct points stay on the device); the candidate arrays hold
// cand_capacity entries.
static int hybrid_core_impl(const double *dxp, const double *dyp, const int64_t *doffp, const int64_t *offsets,
                            int64_t S, int64_t total, int64_t max_len, double inlier_tol, int32_t max_trials,
                            double min_inliers_frac, const double *r_range, double huber_delta, int32_t use_hull,
                            const int64_t *hull_start, int32_t hull_order, uint64_t seed, int32_t *status, double *fit,
                            double *ux, double *uy, int64_t *n_unique, int64_t *cand_offsets, double *cand_x,
                            double *cand_y, uint8_t *mask, double *radius, double *theta, int64_t *argmax_argmin,
                            double *ov_scal, double *dev_uniform, int64_t grid_len, int64_t cand_capacity,
                            hipStream_t st)
{
    PCFIT_REQUIRE(hull_order == PCFIT_HULL_ORDER_CANONICAL || hull_order == PCFIT_HULL_ORDER_QHULL,
                  "hybrid: hull_order must be PCFIT_HULL_ORDER_CANONICAL (0) or PCFIT_HULL_ORDER_QHULL (1)");
    // Small per-slice arrays travel packed, one copy per stage and direction: every copy from or to
    // pageable memory is a blocking call of some tens of microseconds, more than the kernels of a small scan.
    auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
    // host staging of the packed copies: the thread's pinned scratch (a vector when there is none)
    struct HostStage {
        std::vector<char> spare;
        char *at = nullptr;
        void resize(size_t n)
        {
            at = static_cast<char *>(pinned_scratch(n));
            if (!at) {
                spare.resize(n);
                at = spare.data();
            }
        }
        char *data() { return at; }
    } hbuf;

    // 1. distinct points, rank check, hull
    // Qhull's order for every slice and nobody asks for the distinct points: the hull comes by the short route
    // below (interior points of long slices dropped before the sort) and the replay of Qhull's construction
    // reads the slices as they are
    std::vector<int64_t> qhull_everywhere;
    bool qhull_raw = false;
    if (!hull_start && use_hull && hull_order == PCFIT_HULL_ORDER_QHULL) {
        qhull_everywhere.assign((size_t)S, PCFIT_HULL_START_QHULL);
        if (!ux && !uy) qhull_raw = true;
        else hull_start = qhull_everywhere.data();
    }
    DevBuf dux, duy, dhi, dhs, dws, d1;
    PCFIT_TRY(dux.alloc(total * 8));
    PCFIT_TRY(duy.alloc(total * 8));
    PCFIT_TRY(dhi.alloc(total * 8));
    const size_t nu_at = 0, nh_at = (size_t)S * 8, st1_at = (size_t)S * 16, d1_bytes = (size_t)S * 20;
    PCFIT_TRY(d1.alloc(d1_bytes));                     // n_unique | n_hull | status
    int64_t *dnu = reinterpret_cast<int64_t *>(d1.as<char>() + nu_at);
    int64_t *dnh = reinterpret_cast<int64_t *>(d1.as<char>() + nh_at);
    int32_t *dst1 = reinterpret_cast<int32_t *>(d1.as<char>() + st1_at);
    if (hull_start) PCFIT_TRY(dhs.upload(hull_start, S * 8, st));
    // hull candidates and nobody asks for the distinct points: the hull kernel may drop interior points first
    const int hull_only = (use_hull && !ux && !uy && !hull_start) ? 1 : 0;
    PCFIT_TRY(dws.alloc(unique_hull_workspace_bytes(total, S, hull_only)));
    PCFIT_TRY(unique_hull_dev(dxp, dyp, doffp, S, max_len, dux.as<double>(), duy.as<double>(), dnu, dhi.as<int64_t>(),
                              dnh, dst1, dhs.as<int64_t>(), dws.p, dws.bytes, st, hull_only));
    hbuf.resize(d1_bytes);
    PCFIT_TRY(d1.download(hbuf.data(), d1_bytes, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> n_hull(S);
    memcpy(n_unique, hbuf.data() + nu_at, (size_t)S * 8);
    memcpy(n_hull.data(), hbuf.data() + nh_at, (size_t)S * 8);
    memcpy(status, hbuf.data() + st1_at, (size_t)S * 4);
    const int64_t max_hull = S ? *std::max_element(n_hull.begin(), n_hull.end()) : 0;
    if (use_hull && qhull_raw) {
        std::vector<int64_t> lens((size_t)S);
        for (int64_t s = 0; s < S; s++) lens[(size_t)s] = offsets[s + 1] - offsets[s];
        PCFIT_TRY(apply_qhull_starts(qhull_everywhere.data(), nullptr, offsets, lens.data(), nullptr, S, total, dxp, dyp,
                                     dux.as<double>(), duy.as<double>(), doffp, dnh, dhi.as<int64_t>(), st, max_hull));
    } else if (use_hull)
        PCFIT_TRY(apply_qhull_starts(hull_start, dhs.as<int64_t>(), offsets, n_unique, dnu, S, total, dux.as<double>(),
                                     duy.as<double>(), dux.as<double>(), duy.as<double>(), doffp, dnh,
                                     dhi.as<int64_t>(), st, max_hull));
    if (use_hull && (qhull_raw || hull_start)) {
        // (walks in Qhull's order are brought to Qhull's vertex set too: the counts may have gone down)
        PCFIT_HIP(hipMemcpyAsync(n_hull.data(), dnh, (size_t)S * 8, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipStreamSynchronize(st));
    }

    // 2. candidates: hull vertices (circle_fitting.py:191-192) or all distinct points (:196-198)
    int64_t max_cand = 0;
    cand_offsets[0] = 0;
    for (int64_t s = 0; s < S; s++) {
        // fewer than 3 points before deduplication is a failure too (circle_fitting.py:178)
        if (offsets[s + 1] - offsets[s] < 3) status[s] = PCFIT_ST_SPARSE;
        int64_t m = 0;
        if (status[s] == PCFIT_ST_OK) m = (use_hull && n_hull[s] >= 3) ? n_hull[s] : n_unique[s];
        cand_offsets[s + 1] = cand_offsets[s] + m;
        max_cand = std::max(max_cand, m);
    }
    const int64_t n_cand = cand_offsets[S];
    if (n_cand > cand_capacity) {
        set_error("hybrid: the candidate arrays hold %lld entries, %lld needed", (long long)cand_capacity,
                  (long long)n_cand);
        return PCFIT_ERR_INVALID;
    }

    // 3. RANSAC on the candidates: one sampling table per distinct candidate count
    uint64_t state[4];
    PCFIT_TRY(pcfit_seed_state(seed, state));
    std::map<int64_t, int32_t> table_id;
    std::vector<int32_t> of_slice(S, 0);
    std::vector<int64_t> sizes;
    for (int64_t s = 0; s < S; s++) {
        const int64_t m = cand_offsets[s + 1] - cand_offsets[s];
        if (m < 3) continue;
        auto it = table_id.find(m);
        if (it == table_id.end()) {
            it = table_id.emplace(m, (int32_t)table_id.size()).first;
            sizes.push_back(m);
        }
        of_slice[s] = it->second;
    }
    std::vector<int32_t> tables;
    PCFIT_TRY(sampling_tables(state, sizes, max_trials, tables));
    // up: cand_offsets | status | table of slice
    DevBuf d2;
    const size_t co_at = 0, st2_at = (size_t)(S + 1) * 8, of_at = st2_at + (size_t)S * 4, d2_bytes = of_at + (size_t)S * 4;
    hbuf.resize(d2_bytes);
    memcpy(hbuf.data() + co_at, cand_offsets, (size_t)(S + 1) * 8);
    memcpy(hbuf.data() + st2_at, status, (size_t)S * 4);
    memcpy(hbuf.data() + of_at, of_slice.data(), (size_t)S * 4);
    PCFIT_TRY(d2.upload(hbuf.data(), d2_bytes, st));
    const int64_t *dco = reinterpret_cast<const int64_t *>(d2.as<char>() + co_at);
    int32_t *dst = reinterpret_cast<int32_t *>(d2.as<char>() + st2_at);
    const int32_t *dof = reinterpret_cast<const int32_t *>(d2.as<char>() + of_at);
    // everything that goes back at the end lives in one block:
    // fit | argmax_argmin | ov | cand_x | cand_y | radius | theta | mask
    DevBuf dout;
    const size_t C8 = (size_t)n_cand * 8;
    const size_t fit_at = 0, am_at = (size_t)S * 24, ov_at = am_at + (size_t)S * 16, cx_at = ov_at + (size_t)S * 40,
                 cy_at = cx_at + C8, rad_at = cy_at + C8, th_at = rad_at + C8, mask_at = th_at + C8,
                 out_bytes = up8(mask_at + (size_t)n_cand);
    PCFIT_TRY(dout.alloc(out_bytes));
    char *ob = dout.as<char>();
    double *dfit = reinterpret_cast<double *>(ob + fit_at), *dov = reinterpret_cast<double *>(ob + ov_at);
    int64_t *dam = reinterpret_cast<int64_t *>(ob + am_at);
    double *dcx = reinterpret_cast<double *>(ob + cx_at), *dcy = reinterpret_cast<double *>(ob + cy_at);
    double *drad = reinterpret_cast<double *>(ob + rad_at), *dth = reinterpret_cast<double *>(ob + th_at);
    uint8_t *dmask = reinterpret_cast<uint8_t *>(ob + mask_at);
    gather_candidates_kernel<<<dim3((unsigned)S), dim3(256), 0, st>>>(dux.as<double>(), duy.as<double>(), doffp,
                                                                     dhi.as<int64_t>(), dco, use_hull ? 1 : 0, dcx, dcy,
                                                                     dst);
    PCFIT_LAUNCH_CHECK();
    DevBuf dtab, dmodel, d3, drws;
    PCFIT_TRY(dtab.upload(tables.data(), tables.size() * 4, st));
    PCFIT_TRY(dmodel.alloc(S * 24));
    PCFIT_TRY(d3.alloc((size_t)S * 12));               // score | ransac status | best trial
    int32_t *dscore = d3.as<int32_t>(), *drst = dscore + S, *dbest = drst + S;
    PCFIT_TRY(drws.alloc(pcfit_ransac_workspace_bytes(S)));
    PCFIT_TRY(pcfit_ransac_batch_dev(dcx, dcy, dco, S, inlier_tol, max_trials, min_inliers_frac, r_range,
                                     dtab.as<int32_t>(), dof, dmodel.as<double>(), dscore, dbest, dmask, drst, drws.p,
                                     drws.bytes, st));
    std::vector<int32_t> back((size_t)S * 3);          // score | ransac status | status after the gather
    PCFIT_TRY(d3.download(back.data(), (size_t)S * 8, st));
    PCFIT_HIP(hipMemcpyAsync(back.data() + 2 * S, dst, (size_t)S * 4, hipMemcpyDeviceToHost, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    const int32_t *score = back.data(), *rst = score + S, *st2 = rst + S;

    // 4. inliers -> Kasa start -> LM (circle_fitting.py:215-222)
    std::vector<int64_t> inl_off(S + 1, 0);
    int64_t max_inl = 0;
    for (int64_t s = 0; s < S; s++) {
        if (status[s] == PCFIT_ST_OK && st2[s] != PCFIT_ST_OK) status[s] = st2[s];        // candidates rank deficient
        if (status[s] == PCFIT_ST_OK && rst[s] != PCFIT_ST_OK) status[s] = PCFIT_ST_RANSAC;
        const int64_t k = status[s] == PCFIT_ST_OK ? score[s] : 0;
        inl_off[s + 1] = inl_off[s] + k;
        max_inl = std::max(max_inl, k);
    }
    DevBuf dio, dix, diy;
    PCFIT_TRY(dio.upload(inl_off.data(), (S + 1) * 8, st));
    PCFIT_TRY(dix.alloc(inl_off[S] * 8));
    PCFIT_TRY(diy.alloc(inl_off[S] * 8));
    compact_inliers_kernel<<<dim3((unsigned)S), dim3(256), 0, st>>>(dcx, dcy, dco, dmask, dio.as<int64_t>(),
                                                                   dix.as<double>(), diy.as<double>());
    PCFIT_LAUNCH_CHECK();
    PCFIT_TRY(pcfit_fit_batch_dev(dix.as<double>(), diy.as<double>(), dio.as<int64_t>(), S, max_inl, huber_delta, 100,
                                  1e-2, 1e-10, 1e-12, nullptr, dfit, nullptr, nullptr, st));

    // 5. metrics on the candidates about the fitted circle (analyzer.py:83-95)
    DevBuf drm, ddev;
    PCFIT_TRY(drm.alloc(S * 16));
    if (dev_uniform) PCFIT_TRY(ddev.alloc((size_t)S * grid_len * 8));
    PCFIT_TRY(pcfit_residuals_batch_dev(dcx, dcy, dco, S, dfit, dth, nullptr, drad, 0, nullptr, nullptr,
                                        drm.as<double>(), dam, st));
    DevBuf dows;
    if (max_cand > OV_CAP) PCFIT_TRY(dows.alloc(pcfit_ovality_workspace_bytes(n_cand, S)));
    PCFIT_TRY(pcfit_ovality_long_dev(dcx, dcy, dco, S, n_cand, max_cand, dfit, dov, ddev.as<double>(), grid_len, dows.p,
                                     dows.bytes, st));

    // (with very many candidates - all distinct points of long slices - the sections go out one by one
    // instead: a staging copy of that size would cost more than the calls it saves)
    const bool packed = out_bytes <= ((size_t)8 << 20);
    if (packed) {
        hbuf.resize(out_bytes);
        PCFIT_TRY(dout.download(hbuf.data(), out_bytes, st));
    } else {
        PCFIT_HIP(hipMemcpyAsync(fit, dfit, (size_t)S * 24, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipMemcpyAsync(argmax_argmin, dam, (size_t)S * 16, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipMemcpyAsync(ov_scal, dov, (size_t)S * 40, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipMemcpyAsync(cand_x, dcx, C8, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipMemcpyAsync(cand_y, dcy, C8, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipMemcpyAsync(radius, drad, C8, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipMemcpyAsync(theta, dth, C8, hipMemcpyDeviceToHost, st));
        PCFIT_HIP(hipMemcpyAsync(mask, dmask, (size_t)n_cand, hipMemcpyDeviceToHost, st));
    }
    if (ux) PCFIT_TRY(dux.download(ux, total * 8, st));
    if (uy) PCFIT_TRY(duy.download(uy, total * 8, st));
    if (dev_uniform) PCFIT_TRY(ddev.download(dev_uniform, (size_t)S * grid_len * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    if (packed) {
        memcpy(fit, hbuf.data() + fit_at, (size_t)S * 24);
        memcpy(argmax_argmin, hbuf.data() + am_at, (size_t)S * 16);
        memcpy(ov_scal, hbuf.data() + ov_at, (size_t)S * 40);
        if (n_cand) {
            memcpy(cand_x, hbuf.data() + cx_at, C8);
            memcpy(cand_y, hbuf.data() + cy_at, C8);
            memcpy(radius, hbuf.data() + rad_at, C8);
            memcpy(theta, hbuf.data() + th_at, C8);
            memcpy(mask, hbuf.data() + mask_at, (size_t)n_cand);
        }
    }
    return PCFIT_OK;
}

static int hybrid_core(const double *dxp, const double *dyp, const int64_t *doffp, const int64_t *offsets, int64_t S,
                       int64_t total, int64_t max_len, double inlier_tol, int32_t max_trials, double min_inliers_frac,
                       const double *r_range, double huber_delta, int32_t use_hull, const int64_t *hull_start,
                       int32_t hull_order, uint64_t seed, int32_t *status, double *fit, double *ux, double *uy,
                       int64_t *n_unique, int64_t *cand_offsets, double *cand_x, double *cand_y, uint8_t *mask,
                       double *radius, double *theta, int64_t *argmax_argmin, double *ov_scal, double *dev_uniform,
                       int64_t grid_len, int64_t cand_capacity, hipStream_t st)
{
    try {
        return hybrid_core_impl(dxp, dyp, doffp, offsets, S, total, max_len, inlier_tol, max_trials, min_inliers_frac,
                                r_range, huber_delta, use_hull, hull_start, hull_order, seed, status, fit, ux, uy,
                                n_unique, cand_offsets, cand_x, cand_y, mask, radius, theta, argmax_argmin, ov_scal,
                                dev_uniform, grid_len, cand_capacity, st);
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(st);
        set_error("hybrid: %s", e.what());
        return PCFIT_ERR_INVALID;
    }
}

PCFIT_API int pcfit_hybrid_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                 double inlier_tol, int32_t max_trials, double min_inliers_frac,
                                 const double *r_range, double huber_delta, int32_t use_hull,
                                 const int64_t *hull_start, int32_t hull_order, uint64_t seed, int32_t *status,
                                 double *fit, double *ux,
                                 double *uy, int64_t *n_unique, int64_t *cand_offsets, double *cand_x,
                                 double *cand_y, uint8_t *mask, double *radius, double *theta,
                                 int64_t *argmax_argmin, double *ov_scal, double *dev_uniform, int64_t grid_len)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(hull_order == PCFIT_HULL_ORDER_CANONICAL || hull_order == PCFIT_HULL_ORDER_QHULL,
                  "hybrid: hull_order must be PCFIT_HULL_ORDER_CANONICAL (0) or PCFIT_HULL_ORDER_QHULL (1)");
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    const int64_t S = n_slices;
    PCFIT_REQUIRE(status && fit && n_unique && cand_offsets && argmax_argmin && ov_scal, "hybrid: null pointer");
    PCFIT_REQUIRE(total == 0 || (x && y && cand_x && cand_y && mask && radius && theta), "hybrid: null pointer");
    PCFIT_REQUIRE(max_trials >= 0, "hybrid: max_trials negative");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (S + 1) * 8, st));
    return hybrid_core(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), offsets, S, total, max_len, inlier_tol,
                       max_trials, min_inliers_frac, r_range, huber_delta, use_hull, hull_start, hull_order, seed, status,
                       fit, ux, uy, n_unique, cand_offsets, cand_x, cand_y, mask, radius, theta, argmax_argmin, ov_scal,
                       dev_uniform, grid_len, total, st);
}

namespace pcfit {

constexpr int ZR_BLOCK = 256;

// numpy's min / max of the z column: a NaN anywhere makes both NaN.  part: blocks x {min, max, saw NaN}
__global__ __launch_bounds__(ZR_BLOCK) void z_range_part_kernel(const double *__restrict__ xyz, int64_t n,
                                                               double *__restrict__ part)
{
    __shared__ double red[3 * (ZR_BLOCK / WAVE)];
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double lo = inf, hi = -inf, bad = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)ZR_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * ZR_BLOCK) {
        const double z = xyz[3 * i + 2];
        bad = (z == z) ? bad : 1.0;
        lo = z < lo ? z : lo;
        hi = z > hi ? z : hi;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ol = __shfl_xor(lo, m, WAVE), oh = __shfl_xor(hi, m, WAVE), ob = __shfl_xor(bad, m, WAVE);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
        bad = ob > bad ? ob : bad;
    }
    if ((threadIdx.x & 63) == 0) {
        red[3 * (threadIdx.x >> 6)] = lo;
        red[3 * (threadIdx.x >> 6) + 1] = hi;
        red[3 * (threadIdx.x >> 6) + 2] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ZR_BLOCK / WAVE; w++) {
            lo = red[3 * w] < lo ? red[3 * w] : lo;
            hi = red[3 * w + 1] > hi ? red[3 * w + 1] : hi;
            bad = red[3 * w + 2] > bad ? red[3 * w + 2] : bad;
        }
        part[3 * blockIdx.x] = lo;
        part[3 * blockIdx.x + 1] = hi;
        part[3 * blockIdx.x + 2] = bad;
    }
}

__global__ __launch_bounds__(64) void z_range_final_kernel(const double *__restrict__ part, int blocks,
                                                          double *__restrict__ out)
{
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double lo = inf, hi = -inf, bad = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 64) {
        lo = part[3 * b] < lo ? part[3 * b] : lo;
        hi = part[3 * b + 1] > hi ? part[3 * b + 1] : hi;
        bad = part[3 * b + 2] > bad ? part[3 * b + 2] : bad;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ol = __shfl_xor(lo, m, WAVE), oh = __shfl_xor(hi, m, WAVE), ob = __shfl_xor(bad, m, WAVE);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
        bad = ob > bad ? ob : bad;
    }
    if (threadIdx.x == 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        out[0] = bad > 0.0 ? qnan : lo;
        out[1] = bad > 0.0 ? qnan : hi;
    }
}

// bin the resident scan into the windows and run the chain on the slices
static int scan_hybrid(const double *dxyz, const double *dz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t S,
                       double inlier_tol, int32_t max_trials, double min_inliers_frac, const double *r_range,
                       double huber_delta, int32_t use_hull, int32_t hull_order, uint64_t seed, int64_t *offsets,
                       int32_t *status, double *fit, int64_t *n_unique, int64_t *cand_offsets, double *cand_x,
                       double *cand_y, uint8_t *mask, double *radius, double *theta, int64_t *argmax_argmin,
                       double *ov_scal, double *dev_uniform, int64_t grid_len, int64_t cand_capacity, hipStream_t st)
{
    PCFIT_REQUIRE(hull_order == PCFIT_HULL_ORDER_CANONICAL || hull_order == PCFIT_HULL_ORDER_QHULL,
                  "scan_hybrid: hull_order must be PCFIT_HULL_ORDER_CANONICAL (0) or PCFIT_HULL_ORDER_QHULL (1)");
    PCFIT_REQUIRE(n_points >= 0 && S >= 1, "scan_hybrid: need n_points >= 0 and n_slices >= 1");
    PCFIT_REQUIRE(z_lo && z_hi && offsets, "scan_hybrid: null pointer");
    PCFIT_REQUIRE(status && fit && n_unique && cand_offsets && argmax_argmin && ov_scal, "scan_hybrid: null pointer");
    PCFIT_REQUIRE(cand_capacity >= 0 && (cand_capacity == 0 || (cand_x && cand_y && mask && radius && theta)),
                  "scan_hybrid: null pointer");
    PCFIT_REQUIRE(max_trials >= 0, "scan_hybrid: max_trials negative");
    DevBuf doff, dx, dy, dperm;
    PCFIT_TRY(bin_on_device(dxyz, n_points, z_lo, z_hi, S, false, -1, offsets, doff, dx, dy, dperm, st, dz));
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, S, &total, &max_len));
    return hybrid_core(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), offsets, S, total, max_len, inlier_tol,
                       max_trials, min_inliers_frac, r_range, huber_delta, use_hull, nullptr, hull_order, seed, status,
                       fit, nullptr, nullptr, n_unique, cand_offsets, cand_x, cand_y, mask, radius, theta, argmax_argmin,
                       ov_scal, dev_uniform, grid_len, cand_capacity, st);
}

}  // namespace pcfit

PCFIT_API int pcfit_scan_hybrid_batch(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                                      int64_t n_slices, double inlier_tol, int32_t max_trials,
                                      double min_inliers_frac, const double *r_range, double huber_delta,
                                      int32_t use_hull, int32_t hull_order, uint64_t seed, int64_t *offsets,
                                      int32_t *status, double *fit,
                                      int64_t *n_unique, int64_t *cand_offsets, double *cand_x, double *cand_y,
                                      uint8_t *mask, double *radius, double *theta, int64_t *argmax_argmin,
                                      double *ov_scal, double *dev_uniform, int64_t grid_len, int64_t cand_capacity)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_points >= 0 && (xyz || n_points == 0), "scan_hybrid: null pointer");
    hipStream_t st = nullptr;
    DevBuf dxyz;                   // the scan goes to the device once; slices are formed and stay there
    PCFIT_TRY(dxyz.upload(xyz, (size_t)n_points * 24, st));
    return scan_hybrid(dxyz.as<double>(), nullptr, n_points, z_lo, z_hi, n_slices, inlier_tol, max_trials, min_inliers_frac,
                       r_range, huber_delta, use_hull, hull_order, seed, offsets, status, fit, n_unique, cand_offsets,
                       cand_x, cand_y, mask, radius, theta, argmax_argmin, ov_scal, dev_uniform, grid_len, cand_capacity,
                       st);
}

PCFIT_API int pcfit_scan_upload(const double *xyz, int64_t n_points, pcfit_scan **scan, double *z_min_max)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(scan && n_points >= 0 && (xyz || n_points == 0), "scan_upload: null pointer or negative size");
    *scan = nullptr;
    hipStream_t st = nullptr;
    std::unique_ptr<pcfit_scan> sc(new (std::nothrow) pcfit_scan);
    PCFIT_REQUIRE(sc != nullptr, "scan_upload: out of host memory");
    PCFIT_HIP(hipGetDevice(&sc->device));
    sc->n_points = n_points;
    PCFIT_TRY(sc->xyz.upload(xyz, (size_t)n_points * 24, st));
    if (z_min_max) {
        const double qnan = __builtin_nan("");
        z_min_max[0] = z_min_max[1] = qnan;
        if (n_points > 0) {
            const int blocks = (int)std::min<int64_t>(2048, (n_points + ZR_BLOCK - 1) / ZR_BLOCK);
            DevBuf part, out;
            PCFIT_TRY(part.alloc((size_t)blocks * 24));
            PCFIT_TRY(out.alloc(16));
            z_range_part_kernel<<<dim3(blocks), dim3(ZR_BLOCK), 0, st>>>(sc->xyz.as<double>(), n_points,
                                                                         part.as<double>());
            PCFIT_LAUNCH_CHECK();
            z_range_final_kernel<<<dim3(1), dim3(64), 0, st>>>(part.as<double>(), blocks, out.as<double>());
            PCFIT_LAUNCH_CHECK();
            PCFIT_TRY(out.download(z_min_max, 16, st));
        }
    }
    PCFIT_TRY(scan_z_column(*sc, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    *scan = sc.release();
    return PCFIT_OK;
}

PCFIT_API int pcfit_scan_free(pcfit_scan *scan)
{
    PCFIT_ENTER();
    delete scan;
    return PCFIT_OK;
}

PCFIT_API int pcfit_scan_hybrid_resident(const pcfit_scan *scan, const double *z_lo, const double *z_hi,
                                         int64_t n_slices, double inlier_tol, int32_t max_trials,
                                         double min_inliers_frac, const double *r_range, double huber_delta,
                                         int32_t use_hull, int32_t hull_order, uint64_t seed, int64_t *offsets,
                                         int32_t *status, double *fit, int64_t *n_unique, int64_t *cand_offsets, double *cand_x,
                                         double *cand_y, uint8_t *mask, double *radius, double *theta,
                                         int64_t *argmax_argmin, double *ov_scal, double *dev_uniform,
                                         int64_t grid_len, int64_t cand_capacity)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(scan, "scan_hybrid: scan is null");
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    PCFIT_REQUIRE(dev == scan->device, "scan_hybrid: the scan lives on device %d, the current device is %d",
                  scan->device, dev);
    return scan_hybrid(scan->xyz.as<double>(), scan->z.as<double>(), scan->n_points, z_lo, z_hi, n_slices, inlier_tol, max_trials,
                       min_inliers_frac, r_range, huber_delta, use_hull, hull_order, seed, offsets, status, fit, n_unique,
                       cand_offsets, cand_x, cand_y, mask, radius, theta, argmax_argmin, ov_scal, dev_uniform, grid_len,
                       cand_capacity, nullptr);
}

PCFIT_API int pcfit_rank_check_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                     double eps, uint8_t *ok)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(ok && (total == 0 || (x && y)), "rank_check: null pointer");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dok;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dok.alloc(n_slices));
    rank_kernel<<<dim3((unsigned)n_slices), dim3(256), 0, st>>>(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(),
                                                               eps, dok.as<uint8_t>());
    PCFIT_LAUNCH_CHECK();
    PCFIT_TRY(dok.download(ok, n_slices, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_circle3_batch(const double *triples, int64_t n_triples, double *out, uint8_t *ok)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_triples >= 0, "circle3: negative count");
    if (n_triples == 0) return PCFIT_OK;
    PCFIT_REQUIRE(triples && out && ok, "circle3: null pointer");
    hipStream_t st = nullptr;
    DevBuf dt, dout, dok;
    PCFIT_TRY(dt.upload(triples, n_triples * 48, st));
    PCFIT_TRY(dout.alloc(n_triples * 24));
    PCFIT_TRY(dok.alloc(n_triples));
    circle3_kernel<<<dim3((unsigned)((n_triples + 255) / 256)), dim3(256), 0, st>>>(dt.as<double>(), n_triples,
                                                                                  dout.as<double>(),
                                                                                  dok.as<uint8_t>());
    PCFIT_LAUNCH_CHECK();
    PCFIT_TRY(dout.download(out, n_triples * 24, st));
    PCFIT_TRY(dok.download(ok, n_triples, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}
