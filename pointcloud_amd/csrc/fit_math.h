// Arithmetic shared by the circle-fit kernels: the per-point terms of an LM sweep, the LM control
// step (circle_fitting.py:80-109) and the closing algebra of the Kasa fit (circle_fitting.py:35-54).
// fit.hip runs them with a slice on chip, split.hip with a long slice spread over many workgroups.
#pragma once

#include "pcfit_internal.h"

namespace pcfit {

// 3x3 Gaussian elimination with partial pivoting (first largest entry of the
// column, as LAPACK's pivot search); false on an exactly zero pivot.  Row swaps
// are selects and each pivot is inverted once (Newton-refined reciprocal), as
// LAPACK scales a pivot column, so the whole solve is a short branch-free chain:
// every thread of the workgroup runs it between two sweeps.
__device__ __forceinline__ double rcp_nr(double a)
{
    double y = __builtin_amdgcn_rcp(a);
    double e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    e = fma(-a, y, 1.0);
    return fma(y, e, y);
}

__device__ __forceinline__ void swap_if(bool c, double &a, double &b)
{
    const double t = c ? b : a;
    b = c ? a : b;
    a = t;
}

__device__ __forceinline__ bool solve3x3(double A[3][3], double b[3], double x[3])
{
    // column 0
    const bool p1 = fabs(A[1][0]) > fabs(A[0][0]);
    const bool p2 = fabs(A[2][0]) > fmax(fabs(A[0][0]), fabs(A[1][0]));
    const bool s1 = p1 && !p2;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        swap_if(s1, A[0][j], A[1][j]);
        swap_if(p2, A[0][j], A[2][j]);
    }
    swap_if(s1, b[0], b[1]);
    swap_if(p2, b[0], b[2]);
    if (A[0][0] == 0.0)
        return false;
    const double i0 = rcp_nr(A[0][0]);
    const double l1 = A[1][0] * i0, l2 = A[2][0] * i0;
    A[1][1] -= l1 * A[0][1];
    A[1][2] -= l1 * A[0][2];
    b[1] -= l1 * b[0];
    A[2][1] -= l2 * A[0][1];
    A[2][2] -= l2 * A[0][2];
    b[2] -= l2 * b[0];
    // column 1
    const bool q = fabs(A[2][1]) > fabs(A[1][1]);
    swap_if(q, A[1][1], A[2][1]);
    swap_if(q, A[1][2], A[2][2]);
    swap_if(q, b[1], b[2]);
    if (A[1][1] == 0.0)
        return false;
    const double i1 = rcp_nr(A[1][1]);
    const double l = A[2][1] * i1;
    A[2][2] -= l * A[1][2];
    b[2] -= l * b[1];
    if (A[2][2] == 0.0)
        return false;
    x[2] = b[2] * rcp_nr(A[2][2]);
    x[1] = (b[1] - A[1][2] * x[2]) * i1;
    x[0] = (b[0] - A[0][1] * x[1] - A[0][2] * x[2]) * i0;
    return true;
}

struct LmParams {
    double delta, lam0, tol_update, tol_cost;
    int32_t max_iter;
};

// One point's terms of a sweep at parameters (px0, py0, pr), any point: distance zero, beyond the
// Huber threshold delta, or not part of the slice at all (valid false) - all by selects.
//   acc[0..5] H00 H01 H02 H11 H12 H22, acc[6..8] g, acc[9] sum |psi|/r,
//   acc[10] sum of squares of the small residuals, acc[11] linear part of the large ones; low = smallest distance
//   low / high = smallest / largest distance
__device__ __forceinline__ void lm_point_general(double (&acc)[12], double &low, double &high, double px, double py,
                                                 double px0, double py0, double pr, double delta, double root_delta,
                                                 bool valid)
{
    const double dx = px - px0, dy = py - py0;
    const double d2 = fma(dx, dx, dy * dy);
    const bool pos = d2 > 0.0;
    // ri = sqrt(d2) + 1e-16 and 1/ri (first order in the 1e-16)
    const double y0 = rsqrt_nr(d2);
    const double y = pos ? y0 : 0.0;
    const double ri = fma(d2, y, 1e-16);
    const double iv = pos ? fma(-1e-16 * y0, y0, y0) : 1e16;
    const double res = ri - pr;
    const double a = fabs(res);
    const bool big = a > delta;
    const double sw = root_delta * rsqrt_nr(a + 1e-16);      // sqrt(delta / (a + 1e-16))
    double j0 = -dx * iv, j1 = -dy * iv;
    j0 = big ? j0 * sw : j0;
    j1 = big ? j1 * sw : j1;
    double j2 = big ? -sw : -1.0;
    double rw = big ? res * sw : res;
    double psi = big ? delta : a;
    j0 = valid ? j0 : 0.0;
    j1 = valid ? j1 : 0.0;
    j2 = valid ? j2 : 0.0;
    rw = valid ? rw : 0.0;
    psi = valid ? psi : 0.0;
    const double c11 = fma(delta, a - 0.5 * delta, acc[11]), c10 = fma(a, a, acc[10]);
    acc[11] = (valid && big) ? c11 : acc[11];
    acc[10] = (valid && !big) ? c10 : acc[10];
    acc[0] = fma(j0, j0, acc[0]);
    acc[1] = fma(j0, j1, acc[1]);
    acc[2] = fma(j0, j2, acc[2]);
    acc[3] = fma(j1, j1, acc[3]);
    acc[4] = fma(j1, j2, acc[4]);
    acc[5] = fma(j2, j2, acc[5]);
    acc[6] = fma(j0, rw, acc[6]);
    acc[7] = fma(j1, rw, acc[7]);
    acc[8] = fma(j2, rw, acc[8]);
    acc[9] = fma(psi, iv, acc[9]);
    low = valid ? fmin(low, ri) : low;
    high = valid ? fmax(high, ri) : high;
}

__device__ __forceinline__ void lm_point_general(double (&acc)[12], double &low, double px, double py, double px0,
                                                 double py0, double pr, double delta, double root_delta, bool valid)
{
    double high = 0.0;
    lm_point_general(acc, low, high, px, py, px0, py0, pr, delta, root_delta, valid);
}

// The LM loop between two sweeps.  Sweeps deliver their totals (12 sums as above, then the smallest
// distance) into one of two 16-double areas at tot2; the area that holds H, g at the current
// parameters is never the one the next sweep writes.
struct LmState {
    double cx, cy, rad;         // current parameters
    double tx, ty, tr;          // what the next sweep evaluates: first the start itself
    double lam, prev, min_ri;
    double max_ri;              // (fit.hip only: its sweeps deliver the largest distance at tot[13])
    double step[3];
    int32_t it, acc_steps, n_first, n_sweeps;
    int32_t cur;                // which totals area holds H, g at (cx, cy, rad); -1: none yet
    bool first, done;
};

__device__ __forceinline__ void lm_begin(LmState &st, const double par[3], const LmParams &prm)
{
    st.cx = st.tx = par[0];
    st.cy = st.ty = par[1];
    st.rad = st.tr = par[2];
    st.lam = prm.lam0;
    st.prev = __longlong_as_double(0x7ff0000000000000ll);
    st.min_ri = st.max_ri = 0.0;
    st.step[0] = st.step[1] = st.step[2] = 0.0;
    st.it = st.acc_steps = st.n_first = st.n_sweeps = 0;
    st.cur = -1;
    st.first = true;
    st.done = prm.max_iter <= 0;
}

__device__ __forceinline__ int lm_into(const LmState &st) { return st.cur == 0 ? 1 : 0; }

// after the sweep at (tx, ty, tr) has left its totals in area `into`
__device__ __forceinline__ void lm_step(LmState &st, const LmParams &prm, const double *tot2, int into)
{
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const double *trial = tot2 + 16 * into;
    bool adopt;
    if (st.first) {
        // H and g at the start; its cost is not used (the reference starts from cost_prev = inf)
        st.first = false;
        adopt = true;
        st.n_first++;
    } else {
        st.n_sweeps++;
        const double cost = 0.5 * trial[10] + trial[11];
        if (cost < st.prev - prm.tol_cost) {
            st.cx = st.tx;
            st.cy = st.ty;
            st.rad = st.tr;
            st.prev = cost;
            st.lam = st.lam / 3 > 1e-12 ? st.lam / 3 : 1e-12;
            st.acc_steps++;
            adopt = true;
            st.it++;
            st.done = sqrt(st.step[0] * st.step[0] + st.step[1] * st.step[1] + st.step[2] * st.step[2]) < prm.tol_update;
        } else {
            st.lam *= 3;
            adopt = false;
            st.it++;
            if (st.tx == st.cx && st.ty == st.cy && st.tr == st.rad) {
                st.it = prm.max_iter;           // every later candidate is this one again
                st.done = true;
            }
        }
    }
    if (adopt) {
        st.cur = into;
        st.min_ri = trial[12];
        st.max_ri = trial[13];
    }
    if (st.done)
        return;
    // next candidate from the current H, g (a singular damped system costs an iteration, as in the reference)
    const double *hg = tot2 + 16 * st.cur;
    bool have = false;
    while (st.it < prm.max_iter) {
        const double g0 = hg[6], g1 = hg[7], g2 = hg[8];
        double A[3][3] = {{hg[0] + st.lam * hg[0], hg[1], hg[2]},
                          {hg[1], hg[3] + st.lam * hg[3], hg[4]},
                          {hg[2], hg[4], hg[5] + st.lam * hg[5]}};
        double b[3] = {g0, g1, g2};
        if (!solve3x3(A, b, st.step)) {
            st.lam *= 3;
            st.it++;
            continue;
        }
        st.step[0] = -st.step[0];
        st.step[1] = -st.step[1];
        st.step[2] = -st.step[2];
        if (st.prev < inf) {
            const double gd = fabs(g0 * st.step[0] + g1 * st.step[1] + g2 * st.step[2]);
            const double dc2 = st.step[0] * st.step[0] + st.step[1] * st.step[1];
            if (gd + hg[9] * dc2 < 0.25 * prm.tol_cost && 4.0 * dc2 <= st.min_ri * st.min_ri) {
                st.it = prm.max_iter;           // nothing from here on can be accepted
                break;
            }
        }
        st.tx = st.cx + st.step[0];
        st.ty = st.cy + st.step[1];
        st.tr = st.rad + st.step[2];
        if (!(st.tr > 1e-12))
            st.tr = 1e-12;
        have = true;
        break;
    }
    if (!have)
        st.done = true;
}

__device__ __forceinline__ void lm_finish(const LmState &st, double par[3], int32_t iters[4])
{
    par[0] = st.cx;
    par[1] = st.cy;
    par[2] = st.rad;
    iters[0] = st.it;          // iterations the literal loop runs
    iters[1] = st.acc_steps;   // accepted steps
    iters[2] = st.n_first;     // sweeps that only built H and g (the first)
    iters[3] = st.n_sweeps;    // sweeps that evaluated a candidate (cost, and H and g for the next step)
}

// Kasa: from the centred moments m = {suu svv suv suuu svvv suvv svuu} about (mx, my) of n points
// to the circle; returns the status.
__device__ __forceinline__ int kasa_from_moments(const double (&m)[7], double mx, double my, double n, double out[3])
{
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

}  // namespace pcfit
