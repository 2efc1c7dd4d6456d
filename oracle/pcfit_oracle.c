/*
 * pcfit_oracle.c - plain C restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into
 * oracle/_build/libpcfit_oracle.so and loaded by tests/, the smoke check and
 * the cpu_baseline leg of bench.py.  The shipped library never links it.
 *
 * Every floating-point operation is written out (build with
 * -ffp-contract=off; fma() appears only where it is meant), so that the HIP
 * kernels can be compared with this file bit for bit.  File:line citations
 * name the reference lines a function restates (paths relative to the
 * reference checkout).  Third-party arithmetic the reference leans on is
 * restated from its published algorithm:
 *   - numpy Generator.choice / PCG64 / SeedSequence (numpy 2.2.6),
 *   - LAPACK dgesv on a 2x2 system as OpenBLAS 0.3.29 evaluates it (partial
 *     pivoting, reciprocal-scaled multiplier, fused multiply-add in the
 *     triangular solves) - matches np.linalg.solve bit for bit on 20 000
 *     random systems in the build container,
 *   - numpy.histogram / numpy.interp index rules.
 * Parity status: pinned by the .npz files under tests/golden (outputs of the real reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

static const double TWO_PI = 6.283185307179586476925286766559;

/* ------------------------------------------------------------------ */
/* numpy SeedSequence -> PCG64 state                                    */
/* ------------------------------------------------------------------ */
static uint32_t ss_hashmix(uint32_t v, uint32_t *hc)
{
    v ^= *hc;
    *hc *= 0x931e8875u;
    v *= *hc;
    v ^= v >> 16;
    return v;
}

static uint32_t ss_mix(uint32_t x, uint32_t y)
{
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
    r ^= r >> 16;
    return r;
}

/* out[0..3] = state_hi, state_lo, inc_hi, inc_lo of PCG64(SeedSequence(seed)) */
API void orc_seed_pcg64(uint64_t seed, uint64_t out[4])
{
    uint32_t ent[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    int nent = ent[1] ? 2 : 1;                  /* numpy drops leading zero words (keeps one) */
    uint32_t pool[4], hc = 0x43b0d7e5u;
    for (int i = 0; i < 4; i++)
        pool[i] = ss_hashmix(i < nent ? ent[i] : 0u, &hc);
    for (int s = 0; s < 4; s++)
        for (int d = 0; d < 4; d++)
            if (s != d)
                pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], &hc));
    uint32_t w[8], hb = 0x8b51f9ddu;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= 0x58f38dedu;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    uint64_t q[4];
    for (int i = 0; i < 4; i++)
        q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    /* pcg64_set_seed(seed = q0:q1, inc = q2:q3); srandom: state=0; inc=(initseq<<1)|1; step; state+=initstate; step */
    unsigned __int128 initstate = ((unsigned __int128)q[0] << 64) | q[1];
    unsigned __int128 initseq = ((unsigned __int128)q[2] << 64) | q[3];
    const unsigned __int128 mult = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    unsigned __int128 inc = (initseq << 1) | 1, st = 0;
    st = st * mult + inc;
    st += initstate;
    st = st * mult + inc;
    out[0] = (uint64_t)(st >> 64);
    out[1] = (uint64_t)st;
    out[2] = (uint64_t)(inc >> 64);
    out[3] = (uint64_t)inc;
}

typedef struct {
    unsigned __int128 state, inc;
    int has32;
    uint32_t hold;
} pcg_t;

static uint64_t pcg_next64(pcg_t *g)
{
    const unsigned __int128 mult = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    g->state = g->state * mult + g->inc;
    uint64_t hi = (uint64_t)(g->state >> 64), lo = (uint64_t)g->state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

static uint32_t pcg_next32(pcg_t *g)
{
    if (g->has32) {
        g->has32 = 0;
        return g->hold;
    }
    uint64_t r = pcg_next64(g);
    g->has32 = 1;
    g->hold = (uint32_t)(r >> 32);
    return (uint32_t)r;
}

/* uniform integer on [0, top] (Lemire, 32-bit); top == 0 consumes nothing */
static uint32_t pcg_bounded(pcg_t *g, uint32_t top)
{
    if (top == 0)
        return 0;
    if (top == 0xFFFFFFFFu)
        return pcg_next32(g);
    uint32_t span = top + 1;
    uint64_t m = (uint64_t)pcg_next32(g) * span;
    uint32_t low = (uint32_t)m;
    if (low < span) {
        uint32_t thr = (uint32_t)((0xFFFFFFFFull - top) % span);
        while (low < thr) {
            m = (uint64_t)pcg_next32(g) * span;
            low = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

/* one Generator.choice(n, size=3, replace=False): Floyd's sampling over a
 * 4-slot open-addressed set, then a two-step tail shuffle */
static void pcg_choice3(pcg_t *g, int64_t n, int32_t out[3])
{
    int64_t set[4] = { -1, -1, -1, -1 };
    for (int k = 0; k < 3; k++) {
        int64_t j = n - 3 + k;
        int64_t v = (int64_t)pcg_bounded(g, (uint32_t)j);
        int loc = (int)(v & 3);
        while (set[loc] != -1 && set[loc] != v)
            loc = (loc + 1) & 3;
        if (set[loc] == -1) {
            set[loc] = v;
            out[k] = (int32_t)v;
        } else {
            loc = (int)(j & 3);
            while (set[loc] != -1)
                loc = (loc + 1) & 3;
            set[loc] = j;
            out[k] = (int32_t)j;
        }
    }
    for (int i = 2; i >= 1; i--) {
        uint32_t j = pcg_bounded(g, (uint32_t)i);
        int32_t t = out[i];
        out[i] = out[j];
        out[j] = t;
    }
}

/* circle_fitting.py:124-126: the triples of trials 0..trials-1 for a given n */
API void orc_choice3_table(const uint64_t st[4], int64_t n, int32_t trials, int32_t *out)
{
    pcg_t g;
    g.state = ((unsigned __int128)st[0] << 64) | st[1];
    g.inc = ((unsigned __int128)st[2] << 64) | st[3];
    g.has32 = 0;
    g.hold = 0;
    for (int32_t t = 0; t < trials; t++)
        pcg_choice3(&g, n, out + 3 * t);
}

/* hypot rounded to nearest (double-double sum of squares, sqrt, one exact-
 * residual Newton step).  The reference calls libm's hypot through numpy, which
 * on glibc 2.35 is within one ulp and correctly rounded for ~99.5 % of inputs;
 * this form is used so that the radii here and on the GPU are the same bits. */
static double cr_hypot(double x, double y)
{
    double p1 = x * x, e1 = fma(x, x, -p1);
    double p2 = y * y, e2 = fma(y, y, -p2);
    double s = p1 + p2;
    double bv = s - p1;
    double es = (p1 - (s - bv)) + (p2 - bv);
    double tail = es + (e1 + e2);
    double r = sqrt(s);
    if (!(r > 0.0) || !isfinite(r))
        return r;
    double rr = r * r, er = fma(r, r, -rr);
    double resid = ((s - rr) - er) + tail;
    return r + resid / (2.0 * r);
}

/* ------------------------------------------------------------------ */
/* 2x2 solve as dgesv evaluates it                                      */
/* ------------------------------------------------------------------ */
/* rows (a b | c), (d e | f).  Returns the product of the pivots (the
 * determinant up to the sign of the row swap) and the solution in x1, x2. */
static double solve2x2(double a, double b, double c, double d, double e, double f, double *x1, double *x2)
{
    if (fabs(d) > fabs(a)) {
        double t;
        t = a; a = d; d = t;
        t = b; b = e; e = t;
        t = c; c = f; f = t;
    }
    double l = d * (1.0 / a);
    double u22 = e - l * b;
    double y2 = fma(-l, c, f);
    *x2 = y2 / u22;
    *x1 = fma(-b, *x2, c) / a;
    return a * u22;
}

/* circle_fitting.py:22-32.  p = {x1,y1,x2,y2,x3,y3}; 1 = circle in out, 0 = rejected.
 * The determinant test uses the pivot product where numpy exponentiates a sum
 * of logs; the two differ by rounding only, i.e. only for |det| within a few
 * ulp of 1e-12. */
API int orc_circle_3pt(const double *p, double out[3])
{
    double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], x3 = p[4], y3 = p[5];
    double a = 2 * (x2 - x1), b = 2 * (y2 - y1);
    double c = ((x2 * x2 + y2 * y2) - x1 * x1) - y1 * y1;
    double d = 2 * (x3 - x2), e = 2 * (y3 - y2);
    double f = ((x3 * x3 + y3 * y3) - x2 * x2) - y2 * y2;
    double cx, cy;
    double det = solve2x2(a, b, c, d, e, f, &cx, &cy);
    if (!(fabs(det) >= 1e-12))
        return 0;
    out[0] = cx;
    out[1] = cy;
    out[2] = cr_hypot(x1 - cx, y1 - cy);
    return 1;
}

/* ------------------------------------------------------------------ */
/* RANSAC                                       circle_fitting.py:112-143 */
/* ------------------------------------------------------------------ */
/* status: 0 ok, 1 fewer than 3 points, 3 consensus below the minimum.
 * r_range may be NULL.  mask has n bytes.  model/score/best_trial describe
 * the earliest trial of maximal score. */
API int orc_ransac(const double *x, const double *y, int64_t n, double tol, int32_t trials, double min_frac,
                   const double *r_range, const uint64_t st[4], double model[3], int32_t *score_out,
                   uint8_t *mask, int32_t *best_trial)
{
    *score_out = -1;
    *best_trial = -1;
    if (n < 3)
        return 1;
    int32_t *tab = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(trials > 0 ? trials : 1));
    orc_choice3_table(st, n, trials, tab);
    int32_t best = -1;
    for (int32_t t = 0; t < trials; t++) {
        const int32_t *k = tab + 3 * t;
        double p[6] = { x[k[0]], y[k[0]], x[k[1]], y[k[1]], x[k[2]], y[k[2]] };
        double c[3];
        if (!orc_circle_3pt(p, c))
            continue;
        if (r_range && !(r_range[0] <= c[2] && c[2] <= r_range[1]))
            continue;
        int32_t s = 0;
        for (int64_t i = 0; i < n; i++) {
            double dx = x[i] - c[0], dy = y[i] - c[1];
            double err = fabs(sqrt(dx * dx + dy * dy) - c[2]);
            s += (err <= tol);
        }
        if (s > best) {
            best = s;
            *best_trial = t;
            model[0] = c[0];
            model[1] = c[1];
            model[2] = c[2];
        }
    }
    free(tab);
    *score_out = best;
    int32_t need = (int32_t)(min_frac * (double)n);
    if (need < 3)
        need = 3;
    if (best < need)
        return 3;
    for (int64_t i = 0; i < n; i++) {
        double dx = x[i] - model[0], dy = y[i] - model[1];
        mask[i] = (uint8_t)(fabs(sqrt(dx * dx + dy * dy) - model[2]) <= tol);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Kasa fit                                      circle_fitting.py:35-54 */
/* ------------------------------------------------------------------ */
/* status: 0 ok, 1 fewer than 3 points, 4 singular normal equations */
API int orc_kasa(const double *x, const double *y, int64_t n, double out[3])
{
    if (n < 3)
        return 1;
    double sx = 0, sy = 0;
    for (int64_t i = 0; i < n; i++) {
        sx += x[i];
        sy += y[i];
    }
    double mx = sx / (double)n, my = sy / (double)n;
    double suu = 0, svv = 0, suv = 0, suuu = 0, svvv = 0, suvv = 0, svuu = 0;
    for (int64_t i = 0; i < n; i++) {
        double u = x[i] - mx, v = y[i] - my;
        suu += u * u;
        svv += v * v;
        suv += u * v;
        suuu += u * u * u;
        svvv += v * v * v;
        suvv += u * v * v;
        svuu += v * u * u;
    }
    double uc, vc;
    double det = solve2x2(2 * suu, 2 * suv, suuu + suvv, 2 * suv, 2 * svv, svvv + svuu, &uc, &vc);
    if (det == 0.0 || !isfinite(uc) || !isfinite(vc))
        return 4;
    out[0] = mx + uc;
    out[1] = my + vc;
    out[2] = sqrt(uc * uc + vc * vc + (suu + svv) / (double)n);
    return 0;
}

/* ------------------------------------------------------------------ */
/* LM + Huber                                   circle_fitting.py:57-109 */
/* ------------------------------------------------------------------ */
/* Gaussian elimination with partial pivoting on a 3x3 system; 0 when a pivot is exactly zero */
static int solve3x3(double A[3][3], double b[3], double x[3])
{
    int perm[3] = { 0, 1, 2 };
    for (int k = 0; k < 3; k++) {
        int p = k;
        for (int i = k + 1; i < 3; i++)
            if (fabs(A[perm[i]][k]) > fabs(A[perm[p]][k]))
                p = i;
        int t = perm[k];
        perm[k] = perm[p];
        perm[p] = t;
        double piv = A[perm[k]][k];
        if (piv == 0.0)
            return 0;
        for (int i = k + 1; i < 3; i++) {
            double l = A[perm[i]][k] / piv;
            for (int j = k + 1; j < 3; j++)
                A[perm[i]][j] -= l * A[perm[k]][j];
            b[perm[i]] -= l * b[perm[k]];
        }
    }
    for (int k = 2; k >= 0; k--) {
        double s = b[perm[k]];
        for (int j = k + 1; j < 3; j++)
            s -= A[perm[k]][j] * x[j];
        x[k] = s / A[perm[k]][k];
    }
    return 1;
}

static double huber_cost(const double *x, const double *y, int64_t n, double cx, double cy, double rad, double delta)
{
    double quad = 0, lin = 0;
    for (int64_t i = 0; i < n; i++) {
        double dx = x[i] - cx, dy = y[i] - cy;
        double a = fabs((sqrt(dx * dx + dy * dy) + 1e-16) - rad);
        if (a <= delta)
            quad += a * a;
        else
            lin += delta * (a - 0.5 * delta);
    }
    return 0.5 * quad + lin;
}

/* Literal loop: every iteration rebuilds the normal equations and evaluates the
 * candidate's cost.  iters = iterations run, accepts = steps accepted. */
API int orc_lm(const double *x, const double *y, int64_t n, const double init[3], double delta, int32_t max_iter,
               double lam0, double tol_update, double tol_cost, double out[3], int32_t *iters, int32_t *accepts)
{
    double cx = init[0], cy = init[1], rad = init[2];
    double lam = lam0, prev = INFINITY;
    int32_t it = 0, acc = 0;
    for (; it < max_iter; it++) {
        double H[3][3] = { { 0 } }, g[3] = { 0, 0, 0 };
        for (int64_t i = 0; i < n; i++) {
            double dx = x[i] - cx, dy = y[i] - cy;
            double ri = sqrt(dx * dx + dy * dy) + 1e-16;
            double r = ri - rad;
            double a = fabs(r);
            double w = (a > delta) ? delta / (a + 1e-16) : 1.0;
            double sw = sqrt(w);
            double j0 = (-dx / ri) * sw, j1 = (-dy / ri) * sw, j2 = -1.0 * sw, rw = r * sw;
            H[0][0] += j0 * j0; H[0][1] += j0 * j1; H[0][2] += j0 * j2;
            H[1][1] += j1 * j1; H[1][2] += j1 * j2; H[2][2] += j2 * j2;
            g[0] += j0 * rw; g[1] += j1 * rw; g[2] += j2 * rw;
        }
        H[1][0] = H[0][1]; H[2][0] = H[0][2]; H[2][1] = H[1][2];
        double A[3][3], b[3], step[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                A[i][j] = H[i][j];
            A[i][i] = H[i][i] + lam * H[i][i];
            b[i] = g[i];
        }
        if (!solve3x3(A, b, step)) {
            lam *= 3;
            continue;
        }
        step[0] = -step[0]; step[1] = -step[1]; step[2] = -step[2];
        double tx = cx + step[0], ty = cy + step[1], tr = rad + step[2];
        if (!(tr > 1e-12))
            tr = 1e-12;
        double cost = huber_cost(x, y, n, tx, ty, tr, delta);
        if (cost < prev - tol_cost) {
            cx = tx; cy = ty; rad = tr;
            prev = cost;
            lam = lam / 3 > 1e-12 ? lam / 3 : 1e-12;
            acc++;
            if (sqrt(step[0] * step[0] + step[1] * step[1] + step[2] * step[2]) < tol_update) {
                it++;
                break;
            }
        } else {
            lam *= 3;
        }
    }
    out[0] = cx; out[1] = cy; out[2] = rad;
    *iters = it;
    *accepts = acc;
    return 0;
}

/* ------------------------------------------------------------------ */
/* slice binning                                        analyzer.py:49-52 */
/* ------------------------------------------------------------------ */
/* z has stride zs doubles.  offsets: S+1.  perm: capacity cap (member indices,
 * slice after slice, original order inside a slice).  Returns total members
 * (may exceed cap: then only counts are valid). */
API int64_t orc_bin_slices(const double *z, int64_t n, int64_t zs, const double *lo, const double *hi, int64_t S,
                           int64_t *offsets, int64_t *perm, int64_t cap)
{
    int64_t tot = 0;
    for (int64_t s = 0; s < S; s++) {
        offsets[s] = tot;
        double a = lo[s], b = hi[s];
        for (int64_t i = 0; i < n; i++) {
            double v = z[i * zs];
            if (v >= a && v <= b) {
                if (tot < cap)
                    perm[tot] = i;
                tot++;
            }
        }
    }
    offsets[S] = tot;
    return tot;
}

/* ------------------------------------------------------------------ */
/* unique rows, rank check, convex hull      circle_fitting.py:14-19,182-192 */
/* ------------------------------------------------------------------ */
typedef struct { double x, y; } pt_t;

static int pt_cmp(const void *pa, const void *pb)
{
    const pt_t *a = (const pt_t *)pa, *b = (const pt_t *)pb;
    if (a->x < b->x) return -1;
    if (a->x > b->x) return 1;
    if (a->y < b->y) return -1;
    if (a->y > b->y) return 1;
    return 0;
}

/* np.unique(xy, axis=0): rows sorted by x then y, duplicates dropped.  Returns the count. */
API int64_t orc_sort_unique(const double *x, const double *y, int64_t n, double *xo, double *yo)
{
    pt_t *p = (pt_t *)malloc(sizeof(pt_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        p[i].x = x[i];
        p[i].y = y[i];
    }
    qsort(p, (size_t)n, sizeof(pt_t), pt_cmp);
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        if (m && p[i].x == xo[m - 1] && p[i].y == yo[m - 1])
            continue;
        xo[m] = p[i].x;
        yo[m] = p[i].y;
        m++;
    }
    free(p);
    return m;
}

/* circle_fitting.py:14-19 restated without an SVD: the singular values of the
 * centred n x 2 matrix are the root sums of squares of its projections on the
 * principal axes of the 2x2 scatter matrix.  Projecting (instead of taking the
 * small eigenvalue of the scatter matrix) keeps the small one accurate to
 * ~1e-16 of the large one, which the 1e-12 threshold needs. */
API int orc_has_2d_variation(const double *x, const double *y, int64_t n, double eps)
{
    if (n < 3)
        return 0;
    double sx = 0, sy = 0;
    for (int64_t i = 0; i < n; i++) {
        sx += x[i];
        sy += y[i];
    }
    double mx = sx / (double)n, my = sy / (double)n;
    double sxx = 0, sxy = 0, syy = 0;
    for (int64_t i = 0; i < n; i++) {
        double u = x[i] - mx, v = y[i] - my;
        sxx += u * u;
        sxy += u * v;
        syy += v * v;
    }
    double phi = 0.5 * atan2(2 * sxy, sxx - syy);
    double c = cos(phi), s = sin(phi);
    double big = 0, small = 0;
    for (int64_t i = 0; i < n; i++) {
        double u = x[i] - mx, v = y[i] - my;
        double a = c * u + s * v, b = c * v - s * u;
        big += a * a;
        small += b * b;
    }
    double s0 = sqrt(big), s1 = sqrt(small);
    if (s1 > s0) {
        double t = s0;
        s0 = s1;
        s1 = t;
    }
    return s1 > eps * (s0 + eps);
}

static void two_sum(double a, double b, double *s, double *e)
{
    double t = a + b;
    double bv = t - a;
    double av = t - bv;
    *s = t;
    *e = (a - av) + (b - bv);
}

/* exact sign of (bx-ax)(cy-ay) - (by-ay)(cx-ax) */
static int orient_sign(double ax, double ay, double bx, double by, double cx, double cy)
{
    double l = (bx - ax) * (cy - ay), r = (by - ay) * (cx - ax);
    double det = l - r;
    double bound = 3.3306690738754716e-16 * (fabs(l) + fabs(r));
    if (det > bound)
        return 1;
    if (det < -bound)
        return -1;
    /* det = bx*cy - bx*ay - ax*cy - by*cx + by*ax + ay*cx, each product exact as hi+lo */
    double pa[6] = { bx, -bx, -ax, -by, by, ay };
    double pb[6] = { cy, ay, cy, cx, ax, cx };
    double exp_[16];
    int len = 0;
    for (int k = 0; k < 6; k++) {
        double hi = pa[k] * pb[k];
        double lo = fma(pa[k], pb[k], -hi);
        double term[2] = { lo, hi };
        for (int t = 0; t < 2; t++) {
            double q = term[t];
            for (int i = 0; i < len; i++) {
                double s, e;
                two_sum(q, exp_[i], &s, &e);
                exp_[i] = e;
                q = s;
            }
            exp_[len++] = q;
        }
    }
    for (int i = len - 1; i >= 0; i--) {
        if (exp_[i] > 0) return 1;
        if (exp_[i] < 0) return -1;
    }
    return 0;
}

/* Strictly convex hull of sorted, duplicate-free points: indices, counter-
 * clockwise from index 0.  Same vertex set and cyclic order as
 * ConvexHull(xy).vertices (circle_fitting.py:189-190), canonical start. */
API int64_t orc_hull(const double *x, const double *y, int64_t n, int64_t *idx)
{
    if (n < 3) {
        for (int64_t i = 0; i < n; i++)
            idx[i] = i;
        return n;
    }
    int64_t *st = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * n + 2));
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        while (k >= 2 && orient_sign(x[st[k - 2]], y[st[k - 2]], x[st[k - 1]], y[st[k - 1]], x[i], y[i]) <= 0)
            k--;
        st[k++] = i;
    }
    int64_t floor_ = k + 1;
    for (int64_t i = n - 2; i >= 0; i--) {
        while (k >= floor_ && orient_sign(x[st[k - 2]], y[st[k - 2]], x[st[k - 1]], y[st[k - 1]], x[i], y[i]) <= 0)
            k--;
        st[k++] = i;
    }
    k--;
    memcpy(idx, st, sizeof(int64_t) * (size_t)k);
    free(st);
    return k;
}

/* ------------------------------------------------------------------ */
/* polar coordinates, boundaries, residuals      boundary.py, gui.py:972-987 */
/* ------------------------------------------------------------------ */
static double wrap_angle(double dy, double dx)
{
    return fmod(atan2(dy, dx) + TWO_PI, TWO_PI);
}

/* boundary.py:19-22 / analyzer.py:83-86 */
API void orc_polar(const double *x, const double *y, int64_t n, double cx, double cy, double *theta, double *r)
{
    for (int64_t i = 0; i < n; i++) {
        double dx = x[i] - cx, dy = y[i] - cy;
        theta[i] = wrap_angle(dy, dx);
        r[i] = cr_hypot(dx, dy);
    }
}

/* boundary.py:24-35: sel[b] = first index of maximal radius in angular bin b, or -1 */
API void orc_angle_max(const double *x, const double *y, int64_t n, double cx, double cy, int32_t bins, int64_t *sel)
{
    double *best = (double *)malloc(sizeof(double) * (size_t)bins);
    double width = TWO_PI / (double)bins;
    for (int32_t b = 0; b < bins; b++)
        sel[b] = -1;
    for (int64_t i = 0; i < n; i++) {
        double dx = x[i] - cx, dy = y[i] - cy;
        double th = wrap_angle(dy, dx), r = cr_hypot(dx, dy);
        double fb = floor(th / width);
        int32_t b = fb < 0 ? 0 : (fb > (double)(bins - 1) ? bins - 1 : (int32_t)fb);
        if (sel[b] < 0 || r > best[b]) {
            sel[b] = i;
            best[b] = r;
        }
    }
    free(best);
}

/* Residual pass over one slice: theta, dr = hypot - R (either may be NULL),
 * numpy.histogram counts of dr over the H bins whose H+1 edges are given,
 * Rmax/Rmin and the first index attaining each.  gui.py:979-983, analyzer.py:85-90. */
API void orc_residuals(const double *x, const double *y, int64_t n, double cx, double cy, double rad, double *theta,
                       double *dr, int32_t H, const double *edges, int64_t *hist, double rmaxmin[2],
                       int64_t argmaxmin[2])
{
    for (int32_t b = 0; b < H; b++)
        hist[b] = 0;
    double rmax = -INFINITY, rmin = INFINITY;
    int64_t imax = -1, imin = -1;
    for (int64_t i = 0; i < n; i++) {
        double dx = x[i] - cx, dy = y[i] - cy;
        double r = cr_hypot(dx, dy), d = r - rad;
        if (theta) theta[i] = wrap_angle(dy, dx);
        if (dr) dr[i] = d;
        if (r > rmax) { rmax = r; imax = i; }
        if (r < rmin) { rmin = r; imin = i; }
        if (H > 0 && d >= edges[0] && d <= edges[H]) {
            int64_t b = (int64_t)(((d - edges[0]) / (edges[H] - edges[0])) * (double)H);
            if (b == H) b--;
            if (d < edges[b]) b--;
            else if (d >= edges[b + 1] && b != H - 1) b++;
            hist[b]++;
        }
    }
    rmaxmin[0] = rmax; rmaxmin[1] = rmin;
    argmaxmin[0] = imax; argmaxmin[1] = imin;
}

typedef struct { double th, dev, r; int64_t i; } prof_t;

static int prof_cmp(const void *pa, const void *pb)
{
    const prof_t *a = (const prof_t *)pa, *b = (const prof_t *)pb;
    if (a->th < b->th) return -1;
    if (a->th > b->th) return 1;
    return (a->i > b->i) - (a->i < b->i);
}

/* boundary.py:56-90.  scal = {Rmax, Rmin, ovality_abs, ovality_pct, k2_amp};
 * dev_u (K = max(1024, n) doubles) may be NULL.  The k=2 amplitude is the one
 * DFT bin, summed directly. */
API void orc_ovality(const double *x, const double *y, int64_t n, double cx, double cy, double rad, double scal[5],
                     double *dev_u)
{
    prof_t *p = (prof_t *)malloc(sizeof(prof_t) * (size_t)(n + 2));
    double rmax = -INFINITY, rmin = INFINITY;
    for (int64_t i = 0; i < n; i++) {
        double dx = x[i] - cx, dy = y[i] - cy;
        prof_t *q = p + 1 + i;
        q->th = wrap_angle(dy, dx);
        q->r = cr_hypot(dx, dy);
        q->dev = q->r - rad;
        q->i = i;
        if (q->r > rmax) rmax = q->r;
        if (q->r < rmin) rmin = q->r;
    }
    qsort(p + 1, (size_t)n, sizeof(prof_t), prof_cmp);
    p[0] = p[n];
    p[0].th = p[n].th - TWO_PI;
    p[n + 1] = p[1];
    p[n + 1].th = p[1].th + TWO_PI;
    int64_t K = n > 1024 ? n : 1024;
    double step = TWO_PI / (double)K;
    double re = 0, im = 0;
    int64_t j = 0;
    for (int64_t k = 0; k < K; k++) {
        double t = (double)k * step;
        while (j + 1 < n + 1 && p[j + 1].th <= t)
            j++;
        double v;
        if (p[j].th == t) {
            v = p[j].dev;
        } else {
            double slope = (p[j + 1].dev - p[j].dev) / (p[j + 1].th - p[j].th);
            v = slope * (t - p[j].th) + p[j].dev;
        }
        if (dev_u) dev_u[k] = v;
        double ang = 2.0 * TWO_PI * (double)k / (double)K;
        re += v * cos(ang);
        im -= v * sin(ang);
    }
    free(p);
    scal[0] = rmax;
    scal[1] = rmin;
    scal[2] = rmax - rmin;
    scal[3] = rad > 0 ? 100.0 * (rmax - rmin) / rad : NAN;
    scal[4] = sqrt(re * re + im * im) / ((double)K / 2.0);
}

/* ------------------------------------------------------------------ */
/* batched fit + residual pass over CSR slices (CPU baseline timing)    */
/* ------------------------------------------------------------------ */
/* Kasa -> LM (reference defaults) -> residual histogram, slice after slice. */
API void orc_fit_residual_batch(const double *x, const double *y, const int64_t *offsets, int64_t S, int32_t H,
                                const double *edges, double *fit /* S*3 */, int64_t *hist /* S*H */,
                                double *rmaxmin /* S*2 */, int64_t *argmaxmin /* S*2 */)
{
    for (int64_t s = 0; s < S; s++) {
        const double *xs = x + offsets[s], *ys = y + offsets[s];
        int64_t n = offsets[s + 1] - offsets[s];
        double k[3] = { NAN, NAN, NAN }, f[3] = { NAN, NAN, NAN };
        int32_t it, acc;
        if (orc_kasa(xs, ys, n, k) == 0)
            orc_lm(xs, ys, n, k, 1.0, 100, 1e-2, 1e-10, 1e-12, f, &it, &acc);
        fit[3 * s] = f[0]; fit[3 * s + 1] = f[1]; fit[3 * s + 2] = f[2];
        orc_residuals(xs, ys, n, f[0], f[1], f[2], NULL, NULL, H, edges, hist + s * H, rmaxmin + 2 * s,
                      argmaxmin + 2 * s);
    }
}
