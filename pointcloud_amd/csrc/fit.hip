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
#include <type_traits>

#include "pcfit_internal.h"

namespace pcfit {

constexpr int FIT_BLOCK = 256;

// A slice as seen by one thread.  PT > 0: register resident (n <= 256*PT).
// PT == 0: streamed from global memory on every pass.
template <int PT>
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
template <int PT>
__device__ int kasa_slice(const SliceView<PT> &P, double *lds, double out[3])
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
        m[2] += u * v;
        m[3] += uu * u;
        m[4] += vv * v;
        m[5] += u * v * v;
        m[6] += v * u * u;
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

// circle_fitting.py:80-109.  The reference rebuilds H and g in every iteration;
// after a rejected step the parameters have not moved, so H and g are the same
// bits as before and are kept instead.  Once a rejected candidate equals the
// current parameters bit for bit, every later (more strongly damped) candidate
// does too and is rejected with the very same cost, so the loop stops there and
// reports the iteration count the literal loop would have reached.
template <int PT>
__device__ void lm_slice(const SliceView<PT> &P, const LmParams prm, double *lds, double par[3], int32_t iters[4])
{
    double cx = par[0], cy = par[1], rad = par[2];
    double lam = prm.lam0, prev = __longlong_as_double(0x7ff0000000000000ll);
    const double delta = prm.delta;
    double hg[9];   // H00 H01 H02 H11 H12 H22 g0 g1 g2
    bool fresh = false;
    int32_t it = 0, acc = 0, n_hg = 0, n_cost = 0;
    for (; it < prm.max_iter; it++) {
        if (!fresh) {
            n_hg++;
#pragma unroll
            for (int k = 0; k < 9; k++)
                hg[k] = 0.0;
            P.each([&](double px, double py) {
                const double dx = px - cx, dy = py - cy;
                const double ri = sqrt(dx * dx + dy * dy) + 1e-16;
                const double r = ri - rad;
                const double a = fabs(r);
                const double inv = 1.0 / ri;
                double j0 = -dx * inv, j1 = -dy * inv, j2 = -1.0, rw = r;
                if (a > delta) {
                    const double sw = sqrt(delta / (a + 1e-16));
                    j0 *= sw;
                    j1 *= sw;
                    j2 = -sw;
                    rw = r * sw;
                }
                hg[0] += j0 * j0;
                hg[1] += j0 * j1;
                hg[2] += j0 * j2;
                hg[3] += j1 * j1;
                hg[4] += j1 * j2;
                hg[5] += j2 * j2;
                hg[6] += j0 * rw;
                hg[7] += j1 * rw;
                hg[8] += j2 * rw;
            });
            block_sum<9>(hg, lds);
            fresh = true;
        }
        double A[3][3] = {{hg[0] + lam * hg[0], hg[1], hg[2]},
                          {hg[1], hg[3] + lam * hg[3], hg[4]},
                          {hg[2], hg[4], hg[5] + lam * hg[5]}};
        double b[3] = {hg[6], hg[7], hg[8]}, step[3];
        if (!solve3x3(A, b, step)) {
            lam *= 3;
            continue;
        }
        step[0] = -step[0];
        step[1] = -step[1];
        step[2] = -step[2];
        const double tx = cx + step[0], ty = cy + step[1];
        double tr = rad + step[2];
        if (!(tr > 1e-12))
            tr = 1e-12;
        n_cost++;
        double c2[2] = {0.0, 0.0};   // quadratic part, linear part
        P.each([&](double px, double py) {
            const double dx = px - tx, dy = py - ty;
            const double a = fabs((sqrt(dx * dx + dy * dy) + 1e-16) - tr);
            if (a <= delta)
                c2[0] += a * a;
            else
                c2[1] += delta * (a - 0.5 * delta);
        });
        block_sum<2>(c2, lds);
        const double cost = 0.5 * c2[0] + c2[1];
        if (cost < prev - prm.tol_cost) {
            cx = tx;
            cy = ty;
            rad = tr;
            prev = cost;
            lam = lam / 3 > 1e-12 ? lam / 3 : 1e-12;
            fresh = false;
            acc++;
            if (sqrt(step[0] * step[0] + step[1] * step[1] + step[2] * step[2]) < prm.tol_update) {
                it++;
                break;
            }
        } else {
            lam *= 3;
            if (tx == cx && ty == cy && tr == rad) {
                it = prm.max_iter;
                break;
            }
        }
    }
    par[0] = cx;
    par[1] = cy;
    par[2] = rad;
    iters[0] = it;        // iterations of the literal loop
    iters[1] = acc;       // accepted steps
    iters[2] = n_hg;      // passes over the points that built H and g
    iters[3] = n_cost;    // passes over the points that evaluated a candidate's cost
}

template <int PT>
__global__ __launch_bounds__(FIT_BLOCK) void kasa_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                        const int64_t *__restrict__ offsets, double *__restrict__ out,
                                                        int32_t *__restrict__ status)
{
    __shared__ double lds[9 * 16];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT> P;
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

template <int PT>
__global__ __launch_bounds__(FIT_BLOCK) void lm_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                      const int64_t *__restrict__ offsets,
                                                      const double *__restrict__ init, LmParams prm,
                                                      double *__restrict__ out, int32_t *__restrict__ iters)
{
    __shared__ double lds[9 * 16];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT> P;
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
template <int PT>
__global__ __launch_bounds__(FIT_BLOCK) void fit_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                       const int64_t *__restrict__ offsets, LmParams prm,
                                                       double *__restrict__ kasa_out, double *__restrict__ out,
                                                       int32_t *__restrict__ iters, int32_t *__restrict__ status)
{
    __shared__ double lds[9 * 16];
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT> P;
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

// register budget: PT points per thread cost 4*PT VGPRs
static int pick_pt(int64_t max_len)
{
    if (max_len <= 0) return 0;
    if (max_len <= 1 * FIT_BLOCK) return 1;
    if (max_len <= 4 * FIT_BLOCK) return 4;
    if (max_len <= 8 * FIT_BLOCK) return 8;
    if (max_len <= 16 * FIT_BLOCK) return 16;
    if (max_len <= 32 * FIT_BLOCK) return 32;
    return 0;
}

// calls f(std::integral_constant<int, PT>) for the points-per-thread class
template <class F>
static void dispatch_pt(int pt, F &&f)
{
    switch (pt) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    case 16: f(std::integral_constant<int, 16>{}); break;
    case 32: f(std::integral_constant<int, 32>{}); break;
    default: f(std::integral_constant<int, 0>{}); break;
    }
}

int launch_kasa(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len, double *out,
                int32_t *status, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    dispatch_pt(pick_pt(max_len), [&](auto pt) {
        kasa_kernel<decltype(pt)::value><<<dim3((unsigned)S), dim3(FIT_BLOCK), 0, st>>>(x, y, offsets, out, status);
    });
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_lm(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
              const double *init, const LmParams &prm, double *out, int32_t *iters, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    dispatch_pt(pick_pt(max_len), [&](auto pt) {
        lm_kernel<decltype(pt)::value><<<dim3((unsigned)S), dim3(FIT_BLOCK), 0, st>>>(x, y, offsets, init, prm, out,
                                                                                      iters);
    });
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_fit(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
               const LmParams &prm, double *kasa_out, double *out, int32_t *iters, int32_t *status, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    dispatch_pt(pick_pt(max_len), [&](auto pt) {
        fit_kernel<decltype(pt)::value><<<dim3((unsigned)S), dim3(FIT_BLOCK), 0, st>>>(x, y, offsets, prm, kasa_out,
                                                                                       out, iters, status);
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
