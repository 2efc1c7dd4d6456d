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
//   acc[10] sum of squares of the small residuals, acc[11] linear part of the large ones;
//   with 15 accumulators also acc[12..14]: sums of q j0^2, q j0 j1 and q, q = residual / distance (the part of the
//   cost's second derivative that H leaves out; taken as if no point were beyond the threshold)
//   low / high = smallest / largest distance
template <int NA, bool MODEL = (NA >= 15)>
__device__ __forceinline__ void lm_point_general(double (&acc)[NA], double &low, double &high, double px, double py,
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
    // (the Huber weight as a factor that is exactly one for a small residual: three products in place of four
    // selects of doubles, same bits either way)
    const double f = big ? sw : 1.0;
    double j0 = (-dx * iv) * f, j1 = (-dy * iv) * f;
    double j2 = -f;
    double rw = res * f;
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
    if constexpr (NA >= 15 && MODEL) {
        const double q = valid ? res * iv : 0.0;
        const double t0 = (valid ? -dx * iv : 0.0) * q;
        acc[12] = fma(t0, -dx * iv, acc[12]);
        acc[13] = fma(t0, -dy * iv, acc[13]);
        acc[14] += q;
    }
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
    bool regular;               // (fit.hip only) the sweep at (cx, cy, rad) found every point at a distance above
                                // zero and within the Huber threshold, and left the three model sums
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
    st.regular = false;
    st.first = true;
    st.done = prm.max_iter <= 0;
}

__device__ __forceinline__ int lm_into(const LmState &st) { return st.cur == 0 ? 1 : 0; }

// The state as LM_STATE_DOUBLES doubles: fit.hip keeps it in LDS between the sweeps, where the wave that takes the
// next step finds it.
constexpr int LM_STATE_DOUBLES = 16;
__device__ __forceinline__ void lm_state_store(const LmState &st, double *p)
{
    p[0] = st.cx; p[1] = st.cy; p[2] = st.rad;
    p[3] = st.tx; p[4] = st.ty; p[5] = st.tr;
    p[6] = st.lam; p[7] = st.prev; p[8] = st.min_ri; p[9] = st.max_ri;
    p[10] = st.step[0]; p[11] = st.step[1]; p[12] = st.step[2];
    int32_t *q = reinterpret_cast<int32_t *>(p + 13);
    q[0] = st.it; q[1] = st.acc_steps; q[2] = st.n_first; q[3] = st.n_sweeps; q[4] = st.cur;
    q[5] = (st.first ? 1 : 0) | (st.done ? 2 : 0) | (st.regular ? 4 : 0);
}

__device__ __forceinline__ void lm_state_load(LmState &st, const double *p)
{
    st.cx = p[0]; st.cy = p[1]; st.rad = p[2];
    st.tx = p[3]; st.ty = p[4]; st.tr = p[5];
    st.lam = p[6]; st.prev = p[7]; st.min_ri = p[8]; st.max_ri = p[9];
    st.step[0] = p[10]; st.step[1] = p[11]; st.step[2] = p[12];
    const int32_t *q = reinterpret_cast<const int32_t *>(p + 13);
    st.it = q[0]; st.acc_steps = q[1]; st.n_first = q[2]; st.n_sweeps = q[3]; st.cur = q[4];
    st.first = (q[5] & 1) != 0; st.done = (q[5] & 2) != 0; st.regular = (q[5] & 4) != 0;
}

// Whether the candidate p + d (d = st.step, solved from H, g at the current parameters p in hg) can be taken as
// the loop's result without a sweep: the candidate is accepted, and every candidate after it is rejected.
// Needs hg from a sweep that found every point regular (distance above zero, residual within the Huber
// threshold) and that left, at hg[14..16], the sums of q j0^2, q j0 j1 and q (q = residual / distance).
//
// With all points regular the cost is F = 1/2 sum r_i^2, three times differentiable, with
//    grad F = g,   hess F = H + B,   B = sum q_i (I - u_i u_i^T) on the centre block (u_i the unit vector to the point),
// and a third derivative of at most L3 = sum (3 sqrt2 / rho_i + 3 |r_i| / rho_i^2) in norm.  So over the step d
//    F(p + d) <= F(p) + g.d + d.(H + B) d / 2 + L3 |d|^3 / 6                    (i)
//    |g(p + d) - (g + (H + B) d)| <= L3 |d|^2 / 2                                (ii)
// (i) below F(p) - tol_cost (less an allowance for the rounding of the two costs the loop would compare) means
// the loop accepts p + d.  At p + d every later candidate s = -(H' + mu D')^-1 g' has g'.s >= -g'.H'^-1 g' and
// |s| <= |g'| / lmin(H'); H' differs from H by at most dH in norm, so with (ii), the lower bound
//    F(p + d + s) >= F(p + d) + g'.s - |s_c|^2 sum |r_i| / rho_i
// (convexity, as in lm_step) shows that none of them gains a quarter of tol_cost: all are rejected.
// `slack`: how far the centre moves (the distances at p + d lie within that of those at p).
// (every quantity below is a bound, so square roots and quotients are taken from the hardware's reciprocal and
// reciprocal square root and widened by more than their error - a correctly rounded division or square root is
// a chain of some 25 dependent instructions, and a dozen of them sat between the last sweep and the residual pass)
__device__ __forceinline__ double sqrt_above(double v) { return v * __builtin_amdgcn_rsq(v + 1e-300) * (1.0 + 1e-6); }
__device__ __forceinline__ double rcp_above(double v) { return rcp_nr(v) * (1.0 + 1e-9); }       // v > 0
__device__ __forceinline__ bool lm_model_step_is_last(const LmState &st, const LmParams &prm, const double *hg,
                                                      double &slack)
{
    const double d0 = st.step[0], d1 = st.step[1], d2 = st.step[2];
    const double H00 = hg[0], H01 = hg[1], H02 = hg[2], H11 = hg[3], H12 = hg[4], H22 = hg[5];
    const double g0 = hg[6], g1 = hg[7], g2 = hg[8];
    const double n = H22, lo = hg[12], hi = hg[13], R = st.rad, F = st.prev;
    const double dc2 = d0 * d0 + d1 * d1, dc = sqrt_above(dc2), move = dc + fabs(d2);
    const double amax = fmax(hi - R, R - lo);
    // all points stay regular over this step and another no longer than it; distances stay above rho
    const double rho = lo - 2.0 * dc;
    if (!(amax + 3.0 * move <= prm.delta && 16.0 * dc <= lo && R + d2 > 1e-12 && n >= 3.0 && F >= 0.0))
        return false;
    const double inv_rho = rcp_above(rho);
    const double B00 = hg[16] - hg[14], B01 = -hg[15], B11 = hg[14];
    const double Hd0 = H00 * d0 + H01 * d1 + H02 * d2, Hd1 = H01 * d0 + H11 * d1 + H12 * d2,
                 Hd2 = H02 * d0 + H12 * d1 + H22 * d2;
    const double Bd0 = B00 * d0 + B01 * d1, Bd1 = B01 * d0 + B11 * d1;
    const double s1 = sqrt_above(2.0 * n * F) + 3.0 * n * move;  // sum |r_i| anywhere near (Cauchy-Schwarz at p)
    const double L3 = (4.25 * n + 3.0 * s1 * inv_rho) * inv_rho;
    const double dn2 = dc2 + d2 * d2, dn = sqrt_above(dn2);
    const double Ef = L3 * dn2 * dn * (1.0 / 6.0), Eg = 0.5 * L3 * dn2;
    const double dF = (g0 * d0 + g1 * d1 + g2 * d2) + 0.5 * (d0 * Hd0 + d1 * Hd1 + d2 * Hd2) +
                      0.5 * (d0 * Bd0 + d1 * Bd1) + Ef;
    // (the rounding of a cost summed over n / 256 terms per thread, then by tree, and of the reference's pairwise
    // sum over blocks of 128: at most (n / 256 + 200) roundings of F each way)
    if (!(dF < -(prm.tol_cost + 4.4e-16 * (n * (1.0 / 256.0) + 200.0) * F)))
        return false;
    // the gradient at p + d, to within Eg
    const double gh[3] = {g0 + Hd0 + Bd0, g1 + Hd1 + Bd1, g2 + Hd2};
    double A[3][3] = {{H00, H01, H02}, {H01, H11, H12}, {H02, H12, H22}};
    double b[3] = {gh[0], gh[1], gh[2]}, y[3];
    if (!solve3x3(A, b, y))
        return false;
    const double a = gh[0] * y[0] + gh[1] * y[1] + gh[2] * y[2];                    // gh.H^-1 gh
    const double yn = sqrt_above(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    const double gn = sqrt_above(gh[0] * gh[0] + gh[1] * gh[1] + gh[2] * gh[2]);
    // the smallest eigenvalue of H from below: the other two have a product of at most (trace / 2)^2
    const double det = H00 * (H11 * H22 - H12 * H12) - H01 * (H01 * H22 - H12 * H02) + H02 * (H01 * H12 - H11 * H02);
    const double ht = 0.5 * (H00 + H11 + H22);
    if (!(det > 0.0 && ht > 0.0))
        return false;
    const double lmin = det * rcp_nr(ht * ht) * (1.0 - 1e-9);
    const double dH = 12.0 * n * dc * inv_rho;       // each j j^T moves by at most 2 sqrt2 * 2 (2 dc) / rho
    if (!(a >= 0.0 && lmin > 2.0 * dH))
        return false;
    const double lm2 = lmin - dH, inv_lm2 = rcp_above(lm2), kap = 1.0 + dH * inv_lm2;
    const double sn = (gn + Eg) * inv_lm2;            // no later step is longer
    // g'.H'^-1 g' from above; the most any step from p + d can gain is half of g'.(H' + B')^-1 g' plus the third-
    // order term, B' being within dB of B in norm
    const double gHg = kap * (a + 2.0 * yn * Eg + Eg * Eg * rcp_above(lmin));
    const double Bn = fabs(B00) + 2.0 * fabs(B01) + fabs(B11);
    const double dB = n * inv_rho * (move + 5.0 * (amax + move) * dc * inv_rho);
    if (!(lm2 > 2.0 * (Bn + dB)))
        return false;
    const double gain = 0.5 * gHg * lm2 * rcp_above(lm2 - (Bn + dB)) + L3 * sn * sn * sn * (1.0 / 6.0);
    if (!(gain < 0.5 * prm.tol_cost && sn <= dc && 2.0 * sn <= rho))
        return false;
    slack = dc;
    return true;
}

// Whether no candidate from the current parameters p can be accepted any more, by the second-order model (all
// points regular; hg as for lm_model_step_is_last; gd = g.(H + lam D)^-1 g of the step just solved).  For any
// step s that keeps the points regular  F(p + s) >= F(p) + g.s + s.(H + B)s / 2 - L3 |s|^3 / 6
// >= F(p) - g.(H + B)^-1 g / 2 - L3 |s|^3 / 6,  and  g.H^-1 g <= (1 + lam max(D) / lmin(H)) gd.  A gain below
// half of tol_cost is a rejection whatever the rounding of the two costs.
__device__ __forceinline__ bool lm_regular_all_rejected(const LmState &st, const LmParams &prm, const double *hg,
                                                        double gd)
{
    const double H00 = hg[0], H01 = hg[1], H02 = hg[2], H11 = hg[3], H12 = hg[4], H22 = hg[5];
    const double g0 = hg[6], g1 = hg[7], g2 = hg[8];
    const double n = H22, lo = hg[12], hi = hg[13], R = st.rad, F = st.prev;
    const double det = H00 * (H11 * H22 - H12 * H12) - H01 * (H01 * H22 - H12 * H02) + H02 * (H01 * H12 - H11 * H02);
    const double ht = 0.5 * (H00 + H11 + H22);
    if (!(det > 0.0 && ht > 0.0 && n >= 3.0 && F >= 0.0))
        return false;
    const double lmin = det * rcp_nr(ht * ht) * (1.0 - 1e-9);
    const double gn = sqrt_above(g0 * g0 + g1 * g1 + g2 * g2);
    const double sn = gn * rcp_above(lmin);                  // no step from here is longer, whatever the damping
    const double amax = fmax(hi - R, R - lo);
    if (!(amax + 1.5 * sn <= prm.delta && 4.0 * sn <= lo))
        return false;
    const double inv_rho = rcp_above(lo - sn);
    const double Bn = fabs(hg[16] - hg[14]) + 2.0 * fabs(hg[15]) + fabs(hg[14]);
    if (!(lmin > 2.0 * Bn))
        return false;
    const double dmax = fmax(H00, fmax(H11, H22));
    const double s1 = sqrt_above(2.0 * n * F) + 1.5 * n * sn;
    const double L3 = (4.25 * n + 3.0 * s1 * inv_rho) * inv_rho;
    const double gain = 0.5 * gd * (1.0 + st.lam * dmax * rcp_above(lmin)) * lmin * rcp_above(lmin - Bn) +
                        L3 * sn * sn * sn * (1.0 / 6.0);
    return gain < 0.5 * prm.tol_cost;
}

// after the sweep at (tx, ty, tr) has left its totals in area `into`.  AREA: doubles per totals area.  MODEL:
// the totals carry the model sums (`regular` says whether the sweep found every point regular), and a last
// accepted step is taken without its sweep where lm_model_step_is_last allows; min_ri / max_ri are then bounds
// (min_ri at least the smallest distance, max_ri at most the largest).
template <int AREA = 16, bool MODEL = false>
__device__ __forceinline__ void lm_step(LmState &st, const LmParams &prm, const double *tot2, int into,
                                        bool regular = false)
{
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const double *trial = tot2 + AREA * into;
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
        st.regular = regular;
    }
    if (st.done)
        return;
    // next candidate from the current H, g (a singular damped system costs an iteration, as in the reference)
    const double *hg = tot2 + AREA * st.cur;
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
            if constexpr (MODEL) {
                // (the second test first: it is the one that holds on the usual course - start, one accepted
                // candidate, one modelled step - and the two cannot hold together, one proving that this
                // candidate is accepted, the other that none is; so their order decides nothing but the cost)
                double slack;
                const bool last = st.regular && lm_model_step_is_last(st, prm, hg, slack);
                if (!last && st.regular && lm_regular_all_rejected(st, prm, hg, gd)) {
                    st.it = prm.max_iter;
                    break;
                }
                if (last) {
                    // (how far the centre really goes: far from the origin the rounding of the two sums can exceed
                    // the step itself, and the bounds below must hold for the centre that is returned)
                    const double ncx = st.cx + st.step[0], ncy = st.cy + st.step[1];
                    const double gx = ncx - st.cx, gy = ncy - st.cy;
                    slack = fmax(slack, sqrt_above(gx * gx + gy * gy));
                    st.cx = ncx;
                    st.cy = ncy;
                    st.rad = st.rad + st.step[2];
                    st.acc_steps++;
                    st.min_ri += slack * (1.0 + 1e-9);
                    st.max_ri -= slack * (1.0 + 1e-9);
                    st.regular = false;
                    st.it = prm.max_iter;
                    break;
                }
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

// Kasa: from the centred moments m = {suu svv suv suz svz} (z = u^2 + v^2) about (mx, my) of n points
// to the circle; returns the status.
__device__ __forceinline__ int kasa_from_moments(const double (&m)[5], double mx, double my, double n, double out[3])
{
    const double a = 2 * m[0], b = 2 * m[2], d = 2 * m[1];
    const double r0 = m[3], r1 = m[4];
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
