// Batched circle fits, one workgroup per slice (gfx950):
//   kasa_kernel  - algebraic fit   (fit_circle_pratt,      circle_fitting.py:35-54)
//   lm_kernel    - geometric refit (lm_circle_geometric,   circle_fitting.py:57-109)
//   fit_kernel   - both in one launch, the slice read from HBM once
//
// Layout: a slice of up to 256*PT points lives in the registers of a 256-thread
// workgroup (thread t holds points t, t+256, ...: coalesced 8-byte loads), so
// every pass of the iteration runs out of the register file; longer slices are
// re-read from global memory (L2 / Infinity Cache) pass after pass.  Sums use a
// fixed butterfly + wave order, so results are bitwise reproducible.  Nothing
// here is a dense contraction: the bound is the FP64 vector rate (sqrt and
// divide per point and pass), not MFMA and not HBM.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "pcfit_internal.h"

namespace pcfit {


// A slice as seen by one thread.  PT > 0: register resident (n <= 256*PT).
// PT == 0: streamed from global memory on every pass.
template <int PT, int FIT_BLOCK>
struct SliceView {
    const double *xs, *ys;
    int64_t n;
    double rx[PT > 0 ? PT : 1], ry[PT > 0 ? PT : 1];

    __device__ __forceinline__ void load(const double *x, const double *y, int64_t off, int64_t count)
    {
        xs = x + off;
        ys = y + off;
        n = count;
        if constexpr (PT > 0) {
#pragma unroll
            for (int k = 0; k < PT; k++) {
                const int64_t i = threadIdx.x + k * FIT_BLOCK;
                rx[k] = i < n ? xs[i] : 0.0;
                ry[k] = i < n ? ys[i] : 0.0;
            }
        }
    }

    // f(x, y, slice-local index)
    template <class F>
    __device__ __forceinline__ void each_indexed(F &&f) const
    {
        if constexpr (PT > 0) {
#pragma unroll
            for (int k = 0; k < PT; k++) {
                const int32_t i = (int32_t)threadIdx.x + k * FIT_BLOCK;
                if (i < n)
                    f(rx[k], ry[k], i);
                if ((k & 3) == 3)
                    __builtin_amdgcn_sched_barrier(0);      // keep the unrolled bodies from piling up in registers
            }
        } else {
            for (int64_t i = threadIdx.x; i < n; i += FIT_BLOCK)
                f(xs[i], ys[i], (int32_t)i);
        }
    }

    template <class F>
    __device__ __forceinline__ void each(F &&f) const
    {
        if constexpr (PT > 0) {
#pragma unroll
            for (int k = 0; k < PT; k++)
                if ((int64_t)(threadIdx.x + k * FIT_BLOCK) < n)
                    f(rx[k], ry[k]);
        } else {
            for (int64_t i = threadIdx.x; i < n; i += FIT_BLOCK)
                f(xs[i], ys[i]);
        }
    }
};

// circle_fitting.py:35-54.  Uniform result in every thread.
template <int PT, int FIT_BLOCK>
__device__ __forceinline__ int kasa_slice(const SliceView<PT, FIT_BLOCK> &P, double *lds, double out[3])
{
    out[0] = out[1] = out[2] = __longlong_as_double(0x7ff8000000000000ll);
    if (P.n < 3)
        return PCFIT_ST_SPARSE;
    const double n = (double)P.n;
    double s1[2] = {0.0, 0.0};
    P.each([&](double px, double py) {
        s1[0] += px;
        s1[1] += py;
    });
    block_sum<2>(s1, lds);
    const double mx = s1[0] / n, my = s1[1] / n;
    double m[7] = {0, 0, 0, 0, 0, 0, 0};   // suu svv suv suuu svvv suvv svuu
    P.each([&](double px, double py) {
        const double u = px - mx, v = py - my;
        const double uu = u * u, vv = v * v;
        m[0] += uu;
        m[1] += vv;
        m[2] = fma(u, v, m[2]);
        m[3] = fma(uu, u, m[3]);
        m[4] = fma(vv, v, m[4]);
        m[5] = fma(u, vv, m[5]);
        m[6] = fma(v, uu, m[6]);
    });
    block_sum<7>(m, lds);
    const double a = 2 * m[0], b = 2 * m[2], d = 2 * m[1];
    const double r0 = m[3] + m[5], r1 = m[4] + m[6];
    double uc, vc;
    const double det = solve2x2(a, b, r0, b, d, r1, uc, vc);
    int st = PCFIT_ST_OK;
    if (det == 0.0 || !isfinite(uc) || !isfinite(vc)) {
        // exactly singular normal equations (the reference then takes the
        // minimum-norm least-squares solution, circle_fitting.py:50-51)
        st = PCFIT_ST_SINGULAR;
        const double tr = a + d;
        if (!(tr > 0.0)) {
            uc = vc = 0.0;
        } else {
            // rank one: A = lam * w w^T with w the dominant eigenvector
            double wx, wy;
            if (a >= d) { wx = a; wy = b; } else { wx = b; wy = d; }
            const double nw = sqrt(wx * wx + wy * wy);
            wx /= nw;
            wy /= nw;
            const double lam = wx * (a * wx + b * wy) + wy * (b * wx + d * wy);
            const double proj = (wx * r0 + wy * r1) / lam;
            uc = wx * proj;
            vc = wy * proj;
        }
    }
    out[0] = mx + uc;
    out[1] = my + vc;
    out[2] = sqrt(uc * uc + vc * vc + (m[0] + m[1]) / n);
    return st;
}

// 3x3 Gaussian elimination with partial pivoting; false on an exactly zero pivot
__device__ __forceinline__ bool solve3x3(double A[3][3], double b[3], double x[3])
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int p = k;
#pragma unroll
        for (int i = k + 1; i < 3; i++)
            if (fabs(A[i][k]) > fabs(A[p][k]))
                p = i;
        if (p != k) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double t = A[k][j];
                A[k][j] = A[p][j];
                A[p][j] = t;
            }
            const double t = b[k];
            b[k] = b[p];
            b[p] = t;
        }
        const double piv = A[k][k];
        if (piv == 0.0)
            return false;
#pragma unroll
        for (int i = k + 1; i < 3; i++) {
            const double l = A[i][k] / piv;
#pragma unroll
            for (int j = k + 1; j < 3; j++)
                A[i][j] -= l * A[k][j];
            b[i] -= l * b[k];
        }
    }
#pragma unroll
    for (int k = 2; k >= 0; k--) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < 3; j++)
            s -= A[k][j] * x[j];
        x[k] = s / A[k][k];
    }
    return true;
}

struct LmParams {
    double delta, lam0, tol_update, tol_cost;
    int32_t max_iter;
};

// smallest value over the workgroup (order independent)
__device__ __forceinline__ double block_min(double v, double *scratch)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = fmin(v, __shfl_xor(v, m, WAVE));
    const int nw = blockDim.x >> 6;
    if (nw == 1)
        return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    for (int w = 0; w < nw; w++)
        v = fmin(v, scratch[w]);
    return v;
}

// circle_fitting.py:80-109.  Every accept / reject decision is the reference's
// (cost of the candidate against the last accepted cost).  Two things the
// literal loop spends most of its 100 iterations on are not recomputed:
//  * after a rejected step the parameters have not moved, so H and g are the
//    same bits as before and are kept; the sweep that evaluates a candidate's
//    cost also accumulates H and g at the candidate (same distances), so an
//    accepted step needs no further sweep;
//  * the loop ends as soon as no later candidate can be accepted.  The Huber
//    cost F is convex in the residuals and the residual of a point is convex in
//    the centre, which gives, for a step d from the current parameters p,
//        F(p + d) >= F(p) + g.d - |d_c|^2 * sum_i |psi(r_i)| / r_i
//    (psi = Huber influence, r_i the point's distance, valid while
//    |d_c| <= min_i r_i / 2).  When |g.d| plus that curvature term is below a
//    quarter of tol_cost the candidate cannot beat the last cost by tol_cost,
//    and neither can the shorter steps of the stronger damping that would
//    follow: the remaining iterations are all rejections.  The same holds
//    when a rejected candidate equals the current parameters bit for bit.
//    The iteration count the literal loop would have reached is reported.
template <int PT, int FIT_BLOCK>
__device__ __forceinline__ void lm_slice(const SliceView<PT, FIT_BLOCK> &P, const LmParams prm, double *lds, double par[3], int32_t iters[4])
{
    double cx = par[0], cy = par[1], rad = par[2];
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double lam = prm.lam0, prev = inf;
    const double delta = prm.delta;
    const double root_delta = sqrt(delta);
    // One sweep over the points at parameters (px0, py0, pr): the normal equations H, g of the
    // Huber-weighted problem, the curvature sum and the Huber cost, all from the same distances.
    //   acc[0..5] H00 H01 H02 H11 H12 H22, acc[6..8] g, acc[9] sum |psi|/r,
    //   acc[10] sum of squares of the small residuals, acc[11] linear part of the large ones
    // The totals of a sweep stay in LDS (tot, 12 doubles): they are the same for every thread, and
    // keeping two sets of them in registers would cost 48 VGPRs per lane.
    auto sweep = [&](double px0, double py0, double pr, double *tot, double &lo_ri) {
        double acc[12];
#pragma unroll
        for (int k = 0; k < 12; k++)
            acc[k] = 0.0;
        double low = inf;
        P.each([&](double px, double py) {
            const double dx = px - px0, dy = py - py0;
            const double d2 = fma(dx, dx, dy * dy);
            // ri = sqrt(d2) + 1e-16 and 1/ri (first order in the 1e-16)
            const double y0 = d2 > 0.0 ? rsqrt_nr(d2) : 0.0;
            const double ri = fma(d2, y0, 1e-16);
            const double inv = d2 > 0.0 ? fma(-1e-16 * y0, y0, y0) : 1e16;
            const double r = ri - pr;
            const double a = fabs(r);
            double j0 = -dx * inv, j1 = -dy * inv, j2 = -1.0, rw = r;
            double psi = a;
            if (a > delta) {
                const double sw = root_delta * rsqrt_nr(a + 1e-16);      // sqrt(delta / (a + 1e-16))
                j0 *= sw;
                j1 *= sw;
                j2 = -sw;
                rw = r * sw;
                psi = delta;
                acc[11] = fma(delta, a - 0.5 * delta, acc[11]);
            } else {
                acc[10] = fma(a, a, acc[10]);
            }
            acc[0] = fma(j0, j0, acc[0]);
            acc[1] = fma(j0, j1, acc[1]);
            acc[2] = fma(j0, j2, acc[2]);
            acc[3] = fma(j1, j1, acc[3]);
            acc[4] = fma(j1, j2, acc[4]);
            acc[5] = fma(j2, j2, acc[5]);
            acc[6] = fma(j0, rw, acc[6]);
            acc[7] = fma(j1, rw, acc[7]);
            acc[8] = fma(j2, rw, acc[8]);
            acc[9] = fma(psi, inv, acc[9]);
            low = fmin(low, ri);
        });
        block_sum_to_lds<12>(acc, lds, tot);
        lo_ri = block_min(low, lds);
    };

    double *hg = lds + 12 * 16, *trial = lds + 12 * 16 + 12;      // LDS: current and candidate totals
    double min_ri = 0.0, trial_min;
    double tx = cx, ty = cy, tr = rad;          // what the next sweep evaluates: first the start itself
    double step[3] = {0.0, 0.0, 0.0};
    bool first = true;
    int32_t it = 0, acc_steps = 0, n_first = 0, n_sweeps = 0;
    // (one call site for the sweep: its unrolled body is the bulk of the kernel's code)
    while (prm.max_iter > 0) {
        sweep(tx, ty, tr, trial, trial_min);
        bool adopt, done = false;
        if (first) {
            // H and g at the start; its cost is not used (the reference starts from cost_prev = inf)
            first = false;
            adopt = true;
            n_first++;
        } else {
            n_sweeps++;
            const double cost = 0.5 * trial[10] + trial[11];
            if (cost < prev - prm.tol_cost) {
                cx = tx;
                cy = ty;
                rad = tr;
                prev = cost;
                lam = lam / 3 > 1e-12 ? lam / 3 : 1e-12;
                acc_steps++;
                adopt = true;
                it++;
                done = sqrt(step[0] * step[0] + step[1] * step[1] + step[2] * step[2]) < prm.tol_update;
            } else {
                lam *= 3;
                adopt = false;
                it++;
                if (tx == cx && ty == cy && tr == rad) {
                    it = prm.max_iter;           // every later candidate is this one again
                    done = true;
                }
            }
        }
        if (adopt) {
            __syncthreads();                     // everyone has read trial[10..11] and the old totals
            if (threadIdx.x < 10)
                hg[threadIdx.x] = trial[threadIdx.x];
            __syncthreads();
            min_ri = trial_min;
        }
        if (done)
            break;
        // next candidate from the current H, g (a singular damped system costs an iteration, as in the reference)
        bool have = false;
        while (it < prm.max_iter) {
            const double g0 = hg[6], g1 = hg[7], g2 = hg[8];
            double A[3][3] = {{hg[0] + lam * hg[0], hg[1], hg[2]},
                              {hg[1], hg[3] + lam * hg[3], hg[4]},
                              {hg[2], hg[4], hg[5] + lam * hg[5]}};
            double b[3] = {g0, g1, g2};
            if (!solve3x3(A, b, step)) {
                lam *= 3;
                it++;
                continue;
            }
            step[0] = -step[0];
            step[1] = -step[1];
            step[2] = -step[2];
            if (prev < inf) {
                const double gd = fabs(g0 * step[0] + g1 * step[1] + g2 * step[2]);
                const double dc2 = step[0] * step[0] + step[1] * step[1];
                if (gd + hg[9] * dc2 < 0.25 * prm.tol_cost && 4.0 * dc2 <= min_ri * min_ri) {
                    it = prm.max_iter;           // nothing from here on can be accepted
                    break;
                }
            }
            tx = cx + step[0];
            ty = cy + step[1];
            tr = rad + step[2];
            if (!(tr > 1e-12))
                tr = 1e-12;
            have = true;
            break;
        }
        if (!have)
            break;
    }
    par[0] = cx;
    par[1] = cy;
    par[2] = rad;
    iters[0] = it;          // iterations the literal loop runs
    iters[1] = acc_steps;   // accepted steps
    iters[2] = n_first;     // sweeps that only built H and g (the first)
    iters[3] = n_sweeps;    // sweeps that evaluated a candidate (cost, and H and g for the next step)
}

template <int PT, int FIT_BLOCK>
__global__ __launch_bounds__(FIT_BLOCK, (FIT_BLOCK >= 1024 ? 4 : (FIT_BLOCK == 384 ? 3 : 2))) void kasa_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                        const int64_t *__restrict__ offsets, double *__restrict__ out,
                                                        int32_t *__restrict__ status)
{
    __shared__ double lds[12 * 16 + 24];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT, FIT_BLOCK> P;
    P.load(x, y, off, n);
    double c[3];
    const int st = kasa_slice(P, lds, c);
    if (threadIdx.x == 0) {
        out[3 * s] = c[0];
        out[3 * s + 1] = c[1];
        out[3 * s + 2] = c[2];
        if (status)
            status[s] = st;
    }
}

template <int PT, int FIT_BLOCK>
__global__ __launch_bounds__(FIT_BLOCK, (FIT_BLOCK >= 1024 ? 4 : (FIT_BLOCK == 384 ? 3 : 2))) void lm_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                      const int64_t *__restrict__ offsets,
                                                      const double *__restrict__ init, LmParams prm,
                                                      double *__restrict__ out, int32_t *__restrict__ iters)
{
    __shared__ double lds[12 * 16 + 24];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT, FIT_BLOCK> P;
    P.load(x, y, off, n);
    double c[3] = {init[3 * s], init[3 * s + 1], init[3 * s + 2]};
    int32_t it[4] = {0, 0, 0, 0};
    lm_slice(P, prm, lds, c, it);
    if (threadIdx.x == 0) {
        out[3 * s] = c[0];
        out[3 * s + 1] = c[1];
        out[3 * s + 2] = c[2];
        if (iters) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                iters[4 * s + k] = it[k];
        }
    }
}

// Kasa start, then LM, the slice read once.  status follows the Kasa fit; slices
// whose Kasa fit is unavailable (n < 3) get NaN.
template <int PT, int FIT_BLOCK>
__global__ __launch_bounds__(FIT_BLOCK, (FIT_BLOCK >= 1024 ? 4 : (FIT_BLOCK == 384 ? 3 : 2))) void fit_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                       const int64_t *__restrict__ offsets, LmParams prm,
                                                       double *__restrict__ kasa_out, double *__restrict__ out,
                                                       int32_t *__restrict__ iters, int32_t *__restrict__ status)
{
    __shared__ double lds[12 * 16 + 24];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT, FIT_BLOCK> P;
    P.load(x, y, off, n);
    double c[3];
    const int st = kasa_slice(P, lds, c);
    if (threadIdx.x == 0 && kasa_out) {
        kasa_out[3 * s] = c[0];
        kasa_out[3 * s + 1] = c[1];
        kasa_out[3 * s + 2] = c[2];
    }
    int32_t it[4] = {0, 0, 0, 0};
    if (st != PCFIT_ST_SPARSE)
        lm_slice(P, prm, lds, c, it);
    if (threadIdx.x == 0) {
        out[3 * s] = c[0];
        out[3 * s + 1] = c[1];
        out[3 * s + 2] = c[2];
        if (iters) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                iters[4 * s + k] = it[k];
        }
        if (status)
            status[s] = st;
    }
}

struct Extreme2 {
    double r;
    int32_t i;
};
__device__ __forceinline__ Extreme2 better_max(Extreme2 a, Extreme2 b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    return (b.r > a.r || (b.r == a.r && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ Extreme2 better_min(Extreme2 a, Extreme2 b)
{
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    return (b.r < a.r || (b.r == a.r && b.i < a.i)) ? b : a;
}

// Kasa -> LM -> residual statistics with the slice read from HBM once:
// histogram of dr = hypot - R (numpy.histogram rules, H bins, edges given),
// Rmax / Rmin and the first index attaining each (analyzer.py:83-92,
// gui.py:979-983).  Dynamic LDS: (H+1) doubles of edges, then H int32 counters.
template <int PT, int FIT_BLOCK>
__global__ __launch_bounds__(FIT_BLOCK, (FIT_BLOCK >= 1024 ? 4 : (FIT_BLOCK == 384 ? 3 : 2))) void fit_residual_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, LmParams prm,
    double *__restrict__ kasa_out, double *__restrict__ out, int32_t *__restrict__ iters,
    int32_t *__restrict__ status, int32_t H, const double *__restrict__ edges, int64_t *__restrict__ hist,
    double *__restrict__ rmax_rmin, int64_t *__restrict__ argmax_argmin)
{
    extern __shared__ double lds_dyn[];
    __shared__ double lds[12 * 16 + 24];
    __shared__ Extreme2 wmax[FIT_BLOCK / WAVE], wmin[FIT_BLOCK / WAVE];
    double *ledge = lds_dyn;
    int32_t *lhist = reinterpret_cast<int32_t *>(lds_dyn + (H > 0 ? H + 1 : 0));
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT, FIT_BLOCK> P;
    P.load(x, y, off, n);
    for (int b = threadIdx.x; b < H; b += FIT_BLOCK)
        lhist[b] = 0;
    for (int b = threadIdx.x; b <= H && H > 0; b += FIT_BLOCK)
        ledge[b] = edges[b];
    double c[3];
    const int st = kasa_slice(P, lds, c);
    if (threadIdx.x == 0 && kasa_out) {
        kasa_out[3 * s] = c[0];
        kasa_out[3 * s + 1] = c[1];
        kasa_out[3 * s + 2] = c[2];
    }
    int32_t it[4] = {0, 0, 0, 0};
    if (st != PCFIT_ST_SPARSE)
        lm_slice(P, prm, lds, c, it);
    __syncthreads();
    const double cx = c[0], cy = c[1], R = c[2];
    const double e0 = H > 0 ? ledge[0] : 0.0, eH = H > 0 ? ledge[H] : 0.0;
    // numpy.histogram puts d into the bin b with edges[b] <= d < edges[b+1] (last bin closed): it
    // guesses b by scaling and repairs the guess against the edges.  Any guess within one bin of the
    // truth gives the same b, so the guess here multiplies by H / span instead of dividing.
    const double scale = (double)H / (eH - e0);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    Extreme2 mx{-inf, -1}, mn{inf, -1};
    P.each_indexed([&](double px, double py, int32_t i) {
        const double r = cr_hypot(px - cx, py - cy);
        const double d = r - R;
        if (r > mx.r) { mx.r = r; mx.i = i; }          // strict: the first index of an extreme value stays
        if (r < mn.r) { mn.r = r; mn.i = i; }
        if (H > 0 && d >= e0 && d <= eH) {
            // (the LDS arrays are named through lds_dyn here: a captured pointer loses its address space)
            int32_t b = (int32_t)((d - e0) * scale);
            b = b > H - 1 ? H - 1 : b;
            if (d < lds_dyn[b]) b--;
            else if (b < H - 1 && d >= lds_dyn[b + 1]) b++;
            atomicAdd(reinterpret_cast<int32_t *>(lds_dyn + H + 1) + b, 1);
        }
    });
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Extreme2 o;
        o.r = __shfl_xor(mx.r, m, WAVE);
        o.i = __shfl_xor(mx.i, m, WAVE);
        mx = better_max(mx, o);
        o.r = __shfl_xor(mn.r, m, WAVE);
        o.i = __shfl_xor(mn.i, m, WAVE);
        mn = better_min(mn, o);
    }
    if ((threadIdx.x & 63) == 0) {
        wmax[threadIdx.x >> 6] = mx;
        wmin[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < H; b += FIT_BLOCK)
        hist[s * H + b] = (int64_t)lhist[b];
    if (threadIdx.x == 0) {
        for (int w = 1; w < FIT_BLOCK / WAVE; w++) {
            mx = better_max(mx, wmax[w]);
            mn = better_min(mn, wmin[w]);
        }
        rmax_rmin[2 * s] = mx.r;
        rmax_rmin[2 * s + 1] = mn.r;
        argmax_argmin[2 * s] = mx.i;
        argmax_argmin[2 * s + 1] = mn.i;
        out[3 * s] = cx;
        out[3 * s + 1] = cy;
        out[3 * s + 2] = R;
        if (iters) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                iters[4 * s + k] = it[k];
        }
        if (status)
            status[s] = st;
    }
}

// Launch shape for a batch whose longest slice has max_len points: threads per
// workgroup x points per thread (4 VGPRs each).  0 points per thread = stream
// the slice from global memory on every pass.
struct FitShape {
    int block, pt;
};

static FitShape pick_shape(int64_t max_len)
{
    if (max_len <= 0) return {256, 0};
    if (max_len <= 64) return {64, 1};
    if (max_len <= 256) return {256, 1};
    if (max_len <= 1024) return {256, 4};
    if (max_len <= 2048) return {256, 8};
    if (max_len <= 4096) return {512, 8};
    if (max_len <= 8192) {
        static const FitShape big = [] {
            // measured on MI355X, 10 000 slices x 8192 points: 256x32 beats 512x16 and 1024x8 (fewer waves
            // to reduce over, more independent points per lane); the variable is a tuning knob
            const char *e = getenv("PCFIT_FIT_SHAPE");
            if (e && !strcmp(e, "512x16")) return FitShape{512, 16};
            if (e && !strcmp(e, "384x22")) return FitShape{384, 22};
            if (e && !strcmp(e, "320x26")) return FitShape{320, 26};
            if (e && !strcmp(e, "1024x8")) return FitShape{1024, 8};
            return FitShape{256, 32};
        }();
        return big;
    }
    return {256, 0};
}

template <int B, int P>
struct ShapeTag {
    static constexpr int block = B, pt = P;
};

// calls f(ShapeTag<block, pt>)
template <class F>
static void dispatch_shape(FitShape sh, F &&f)
{
    const int key = sh.block * 100 + sh.pt;
    switch (key) {
    case 64 * 100 + 1: f(ShapeTag<64, 1>{}); break;
    case 256 * 100 + 1: f(ShapeTag<256, 1>{}); break;
    case 256 * 100 + 4: f(ShapeTag<256, 4>{}); break;
    case 256 * 100 + 8: f(ShapeTag<256, 8>{}); break;
    case 512 * 100 + 8: f(ShapeTag<512, 8>{}); break;
    case 256 * 100 + 32: f(ShapeTag<256, 32>{}); break;
    case 512 * 100 + 16: f(ShapeTag<512, 16>{}); break;
    case 384 * 100 + 22: f(ShapeTag<384, 22>{}); break;
    case 320 * 100 + 26: f(ShapeTag<320, 26>{}); break;
    case 1024 * 100 + 8: f(ShapeTag<1024, 8>{}); break;
    default: f(ShapeTag<256, 0>{}); break;
    }
}

int launch_kasa(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len, double *out,
                int32_t *status, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    dispatch_shape(pick_shape(max_len), [&](auto t) {
        using T = decltype(t);
        kasa_kernel<T::pt, T::block><<<dim3((unsigned)S), dim3(T::block), 0, st>>>(x, y, offsets, out, status);
    });
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_lm(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
              const double *init, const LmParams &prm, double *out, int32_t *iters, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    dispatch_shape(pick_shape(max_len), [&](auto t) {
        using T = decltype(t);
        lm_kernel<T::pt, T::block><<<dim3((unsigned)S), dim3(T::block), 0, st>>>(x, y, offsets, init, prm, out, iters);
    });
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_fit(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
               const LmParams &prm, double *kasa_out, double *out, int32_t *iters, int32_t *status, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    dispatch_shape(pick_shape(max_len), [&](auto t) {
        using T = decltype(t);
        fit_kernel<T::pt, T::block><<<dim3((unsigned)S), dim3(T::block), 0, st>>>(x, y, offsets, prm, kasa_out, out,
                                                                                  iters, status);
    });
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_fit_residual(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
                        const LmParams &prm, double *kasa_out, double *out, int32_t *iters, int32_t *status, int32_t H,
                        const double *edges, int64_t *hist, double *rmm, int64_t *amm, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    const size_t lds = H > 0 ? (size_t)(H + 1) * 8 + (size_t)H * 4 : 0;
    dispatch_shape(pick_shape(max_len), [&](auto t) {
        using T = decltype(t);
        fit_residual_kernel<T::pt, T::block><<<dim3((unsigned)S), dim3(T::block), lds, st>>>(
            x, y, offsets, prm, kasa_out, out, iters, status, H, edges, hist, rmm, amm);
    });
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API int pcfit_kasa_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                   int64_t max_slice_len, double *cxcyr, int32_t *status, void *stream)
{
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "kasa: n_slices out of range");
    PCFIT_REQUIRE(n_slices == 0 || (x && y && offsets && cxcyr), "kasa: null pointer");
    return launch_kasa(x, y, offsets, n_slices, max_slice_len, cxcyr, status, as_stream(stream));
}

PCFIT_API int pcfit_lm_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                 int64_t max_slice_len, const double *init, double huber_delta, int32_t max_iter,
                                 double lm_lambda0, double tol_update, double tol_cost, double *out, int32_t *iters,
                                 void *stream)
{
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "lm: n_slices out of range");
    PCFIT_REQUIRE(n_slices == 0 || (x && y && offsets && init && out), "lm: null pointer");
    PCFIT_REQUIRE(max_iter >= 0, "lm: max_iter negative");
    LmParams prm{huber_delta, lm_lambda0, tol_update, tol_cost, max_iter};
    return launch_lm(x, y, offsets, n_slices, max_slice_len, init, prm, out, iters, as_stream(stream));
}

PCFIT_API int pcfit_fit_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                  int64_t max_slice_len, double huber_delta, int32_t max_iter, double lm_lambda0,
                                  double tol_update, double tol_cost, double *kasa_out, double *out, int32_t *iters,
                                  int32_t *status, void *stream)
{
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "fit: n_slices out of range");
    PCFIT_REQUIRE(n_slices == 0 || (x && y && offsets && out), "fit: null pointer");
    PCFIT_REQUIRE(max_iter >= 0, "fit: max_iter negative");
    LmParams prm{huber_delta, lm_lambda0, tol_update, tol_cost, max_iter};
    return launch_fit(x, y, offsets, n_slices, max_slice_len, prm, kasa_out, out, iters, status, as_stream(stream));
}

PCFIT_API int pcfit_fit_residuals_batch_dev(const double *x, const double *y, const int64_t *offsets,
                                            int64_t n_slices, int64_t max_slice_len, double huber_delta,
                                            int32_t max_iter, double lm_lambda0, double tol_update, double tol_cost,
                                            double *kasa_out, double *out, int32_t *iters, int32_t *status,
                                            int32_t n_bins, const double *edges, int64_t *hist, double *rmax_rmin,
                                            int64_t *argmax_argmin, void *stream)
{
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "fit_residuals: n_slices out of range");
    PCFIT_REQUIRE(n_bins >= 0 && n_bins <= 4096, "fit_residuals: n_bins must be 0..4096");
    PCFIT_REQUIRE(n_slices == 0 || (x && y && offsets && out && rmax_rmin && argmax_argmin),
                  "fit_residuals: null pointer");
    PCFIT_REQUIRE(n_bins == 0 || (edges && hist), "fit_residuals: histogram asked for without edges/hist");
    PCFIT_REQUIRE(max_iter >= 0, "fit_residuals: max_iter negative");
    LmParams prm{huber_delta, lm_lambda0, tol_update, tol_cost, max_iter};
    return launch_fit_residual(x, y, offsets, n_slices, max_slice_len, prm, kasa_out, out, iters, status, n_bins,
                               edges, hist, rmax_rmin, argmax_argmin, as_stream(stream));
}
