// Batched circle fits, one workgroup per slice (gfx950):
//   kasa_kernel  - algebraic fit   (fit_circle_pratt,      circle_fitting.py:35-54)
//   lm_kernel    - geometric refit (lm_circle_geometric,   circle_fitting.py:57-109)
//   fit_kernel   - both in one launch, the slice read from HBM once
//   fit_residual_kernel - both plus the residual statistics, still one read
//
// A slice stays on chip for all passes of the iteration.  Thread t of a
// workgroup of FIT_BLOCK threads owns points t, t+FIT_BLOCK, ... (coalesced
// loads): the first PT_R of them live in its registers, the next PT_L in LDS.
// Slices of up to 4096 points are all registers (16 points = 64 VGPRs per
// thread); up to 8192 points half of the slice goes to LDS (64 KB), which
// leaves the register file and the LDS of a CU room for two workgroups: while
// one sits in a reduction or solves its 3x3 system the other one's sweep keeps
// the vector units busy.  Longer slices are re-read from global memory (L2 /
// Infinity Cache) pass after pass.  Sums use a fixed butterfly + wave order, so
// results are bitwise reproducible.  Nothing here is a dense contraction: the
// bound is the FP64 vector rate (a reciprocal square root and some forty
// multiply-adds per point and pass), not MFMA and not HBM.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "pcfit_internal.h"
#include "fit_math.h"
#include "angle_math.h"

namespace pcfit {

// dynamic LDS of the kernels in this file
extern __shared__ __attribute__((aligned(16))) double lds_dyn[];

// Optional phase timing (tools only; -DPCFIT_PHASE_CLOCK): thread 0 of every
// workgroup adds the clocks spent between marks to g_phase[k].
#ifdef PCFIT_PHASE_CLOCK
__device__ unsigned long long g_phase[16];
// (one workgroup in 64 reports: ten thousand workgroups adding to the same few words would be what is measured)
struct Phase {
    long long last;
    __device__ __forceinline__ bool on() const { return threadIdx.x == 0 && (blockIdx.x & 63) == 0; }
    __device__ __forceinline__ void init() { last = on() ? wall_clock64() : 0; }
    __device__ __forceinline__ void mark(int k)
    {
        if (on()) {
            const long long now = wall_clock64();
            atomicAdd(&g_phase[k], (unsigned long long)(now - last));
            last = now;
        }
    }
};
#else
struct Phase {
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void mark(int) {}
};
#endif

// How a workgroup walks over the points of its slice.  Thread t visits points
// t, t+STRIDE, t+2*STRIDE, ... in that order, GROUP of them at a time:
//     full_groups(fg):  fg(px[GROUP], py[GROUP], i0)   points i0 + u*STRIDE, u < GROUP,
//                       for every group in which all threads hold GROUP points of the slice;
//     ragged_group(fg): the same call, at most once, for the group at the end of the slice in
//                       which only the points with an index below n count (the coordinates of
//                       the others are zeros or a repeat of the last point).
// The bulk of a slice thus runs through bodies without any per-point branch, so
// that the dependency chains of the GROUP points interleave.
//
// Storage: points [0, PT_R*FIT_BLOCK) in registers, the next PT_L*FIT_BLOCK as
// (x, y) pairs in LDS at lds_dyn[0 ..); with PT_R == PT_L == 0 the slice is
// streamed from global memory on every pass.  A slice longer than the storage
// is cut there (callers pick the storage from the longest slice of the batch).
template <int PT_R, int PT_L, int FIT_BLOCK>
struct SliceView {
    static constexpr bool STREAM = PT_R == 0 && PT_L == 0;
    // (with 64 VGPRs of points per thread a group of 4 no longer fits the register file beside them)
    static constexpr int GROUP = (STREAM || (PT_R % 4 == 0 && PT_L % 4 == 0)) ? 4 : (PT_R % 2 == 0 && PT_L % 2 == 0) ? 2 : 1;
    static constexpr int STRIDE = FIT_BLOCK;
    static constexpr int LDS_DOUBLES = 2 * PT_L * FIT_BLOCK;       // what the points take of lds_dyn
    const double *xs, *ys;
    int32_t n;
    double rx[PT_R > 0 ? PT_R : 1], ry[PT_R > 0 ? PT_R : 1];
    // This thread's share of the power sums of u = x - x0, v = y - y0 about the slice's first point (x0, y0),
    // taken while the slice is loaded (not STREAM): su sv suu sz suv suz svz with z = u^2 + v^2 (the fit needs
    // the third moments only as sum u z and sum v z).  The algebraic fit needs moments about the mean; these
    // give them without a pass of their own and without waiting for the mean (kasa_slice), and the first point
    // is as near the data as any shift known before the loads return.
    static constexpr int NSUM = 7;
    double m9[NSUM];
    double x0, y0;

    __device__ __forceinline__ static void power_sums(double (&q)[NSUM], double u, double v)
    {
        const double uu = u * u, z = fma(v, v, uu);
        q[0] += u;
        q[1] += v;
        q[2] += uu;
        q[3] += z;
        q[4] = fma(u, v, q[4]);
        q[5] = fma(u, z, q[5]);
        q[6] = fma(v, z, q[6]);
    }

    __device__ __forceinline__ void take(double vx, double vy, bool valid)
    {
        power_sums(m9, valid ? vx - x0 : 0.0, valid ? vy - y0 : 0.0);
    }

    __device__ __forceinline__ void load(const double *x, const double *y, int64_t off, int64_t count)
    {
        xs = x + off;
        ys = y + off;
        // n = min(count, what the storage holds), spelled with 32-bit compares: hipcc 7.2 turns the
        // uniform 64-bit form of this clamp into a scalar select on a stale condition code
        constexpr uint32_t CAP = STREAM ? 0x7fffffffu : (uint32_t)(PT_R + PT_L) * FIT_BLOCK;
        const uint32_t hi = (uint32_t)((uint64_t)count >> 32), lo = (uint32_t)count;
        n = (hi != 0 || lo > CAP) ? (int32_t)CAP : (int32_t)lo;
        // Loads come first in a workgroup's life and the CU's other workgroup is usually in the middle of
        // its arithmetic: raise the priority of these waves so that their loads get issued at once.
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int k = 0; k < NSUM; k++)
            m9[k] = 0.0;
        x0 = n > 0 ? xs[0] : 0.0;           // (uniform addresses: scalar loads)
        y0 = n > 0 ? ys[0] : 0.0;
        // All loads of a thread are issued before anything looks at what they bring (the scheduling fence keeps
        // the compiler from weaving the arithmetic in between, which left eight loads in flight per wave and a
        // workgroup waiting four times as long for its 128 KB): the memory system then sees the whole slice at once.
        if (!STREAM && n == (PT_R + PT_L) * FIT_BLOCK) {
            // a slice that fills the storage (byte offsets in 32 bits - the storage holds far less than 4 GB -
            // so that a load's address is the uniform base plus one vector register, not a 64-bit vector sum)
            // (the thread's own bytes lie below FIT_BLOCK * 8, a power of two: the two parts of an offset share
            // no bit, and joined by "or" the compiler sees that nothing carries)
            static_assert((FIT_BLOCK & (FIT_BLOCK - 1)) == 0, "offsets are joined bit by bit");
            auto at = [](const double *base, uint32_t k) {
                const uint32_t bytes = (threadIdx.x * 8u) | (k * (uint32_t)FIT_BLOCK * 8u);
                return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + bytes);
            };
#pragma unroll
            for (int k = 0; k < PT_R; k++) {
                rx[k] = at(xs, k);
                ry[k] = at(ys, k);
            }
            double tx[PT_L > 0 ? PT_L : 1], ty[PT_L > 0 ? PT_L : 1];
#pragma unroll
            for (int k = 0; k < PT_L; k++) {
                tx[k] = at(xs, PT_R + k);
                ty[k] = at(ys, PT_R + k);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < PT_R; k++)
                take(rx[k], ry[k], true);
#pragma unroll
            for (int k = 0; k < PT_L; k++) {
                const int32_t j = (int32_t)threadIdx.x + k * FIT_BLOCK;
                lds_dyn[2 * j] = tx[k];
                lds_dyn[2 * j + 1] = ty[k];
                take(tx[k], ty[k], true);
            }
        } else if (n > 0) {
            // every load is unconditional (a point past the end reads the slice's last one and is then
            // replaced by zero), so that all of them can be in flight together
            const int32_t last = n - 1;
            // how many of this thread's points lie inside the slice (point k of the thread does when k < mine:
            // one register for all the tests below instead of an index per point)
            const int32_t mine = (int32_t)threadIdx.x < n ? (last - (int32_t)threadIdx.x) / FIT_BLOCK + 1 : 0;
#pragma unroll
            for (int k = 0; k < PT_R; k++) {
                const int32_t i = threadIdx.x + k * FIT_BLOCK;
                rx[k] = xs[i < last ? i : last];
                ry[k] = ys[i < last ? i : last];
            }
            // the LDS part travels through registers
            double tx[PT_L > 0 ? PT_L : 1], ty[PT_L > 0 ? PT_L : 1];
#pragma unroll
            for (int k = 0; k < PT_L; k++) {
                const int32_t i = threadIdx.x + (PT_R + k) * FIT_BLOCK;
                tx[k] = xs[i < last ? i : last];
                ty[k] = ys[i < last ? i : last];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < PT_R; k++) {
                take(rx[k], ry[k], k < mine);
                rx[k] = k < mine ? rx[k] : 0.0;
                ry[k] = k < mine ? ry[k] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < PT_L; k++) {
                const int32_t j = (int32_t)threadIdx.x + k * FIT_BLOCK;
                lds_dyn[2 * j] = tx[k];
                lds_dyn[2 * j + 1] = ty[k];
                take(tx[k], ty[k], PT_R + k < mine);
            }
        } else {
#pragma unroll
            for (int k = 0; k < PT_R; k++)
                rx[k] = ry[k] = 0.0;
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (PT_L > 0)
            __syncthreads();
    }

    // how many points full_groups() hands to each thread
    __device__ __forceinline__ int32_t full_points_per_thread() const { return n / (GROUP * FIT_BLOCK) * GROUP; }

    // halt (streamed slices only): when the pointed-to flag is set in any lane of the wave after a group, the
    // wave leaves the remaining groups alone
    template <class FG>
    __device__ __forceinline__ void full_groups(FG &&fg, const bool *halt = nullptr) const
    {
        constexpr int U = GROUP;
        if constexpr (STREAM) {
            // the loads of the two trips after the current one are in flight while it is worked on: one trip's
            // arithmetic is far shorter than a trip to HBM.  Three sets of registers take turns (a loop of three
            // trips, so that no set has to be copied into another, which would wait for its loads).
            const int32_t trips = n / (U * FIT_BLOCK);
            double ax[U], ay[U], bx[U], by[U], cx[U], cy[U];
            if (trips == 0)
                return;
            // (every fetch is issued whatever t: past the end it reads the last trip again.  A load behind a
            // branch leaves the compiler unsure how many are in flight, and it then waits for all of them.)
            auto fetch = [&](double (&qx)[U], double (&qy)[U], int32_t t) {
                const int32_t at = (t < trips ? t : trips - 1) * (U * FIT_BLOCK) + (int32_t)threadIdx.x;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    qx[u] = xs[at + u * FIT_BLOCK];
                    qy[u] = ys[at + u * FIT_BLOCK];
                }
            };
            fetch(ax, ay, 0);
            fetch(bx, by, 1);
            for (int32_t t = 0; t < trips; t += 3) {
                if (halt && __builtin_amdgcn_ballot_w64(*halt) != 0ull)
                    break;
                fetch(cx, cy, t + 2);
                fg(ax, ay, t * (U * FIT_BLOCK) + (int32_t)threadIdx.x);
                fetch(ax, ay, t + 3);
                if (t + 1 < trips)
                    fg(bx, by, (t + 1) * (U * FIT_BLOCK) + (int32_t)threadIdx.x);
                fetch(bx, by, t + 4);
                if (t + 2 < trips)
                    fg(cx, cy, (t + 2) * (U * FIT_BLOCK) + (int32_t)threadIdx.x);
            }
        } else {
#pragma unroll
            for (int g = 0; g < PT_R / U; g++) {
                if ((g + 1) * U * FIT_BLOCK <= n) {
                    double px[U], py[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        px[u] = rx[g * U + u];
                        py[u] = ry[g * U + u];
                    }
                    fg(px, py, g * U * FIT_BLOCK + (int32_t)threadIdx.x);
                }
            }
            if constexpr (PT_L > 0) {
                // a rolled loop with a uniform trip count; the pairs of the next trip are read before
                // the current ones are worked on
                const int32_t in_lds = n - PT_R * FIT_BLOCK;
                const int32_t trips = in_lds > 0 ? in_lds / (U * FIT_BLOCK) : 0;
                double qx[U], qy[U];
                int32_t j = threadIdx.x;
                if (trips > 0) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        qx[u] = lds_dyn[2 * (j + u * FIT_BLOCK)];
                        qy[u] = lds_dyn[2 * (j + u * FIT_BLOCK) + 1];
                    }
                }
                for (int32_t t = 0; t < trips; t++, j += U * FIT_BLOCK) {
                    double px[U], py[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        px[u] = qx[u];
                        py[u] = qy[u];
                    }
                    const int32_t nxt = t + 1 < trips ? j + U * FIT_BLOCK : j;      // the last trip reads itself again
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        qx[u] = lds_dyn[2 * (nxt + u * FIT_BLOCK)];
                        qy[u] = lds_dyn[2 * (nxt + u * FIT_BLOCK) + 1];
                    }
                    fg(px, py, PT_R * FIT_BLOCK + j);
                }
            }
        }
    }

    // the first full group alone (what full_groups() would hand out first); false when the slice has none
    template <class FG>
    __device__ __forceinline__ bool first_group(FG &&fg) const
    {
        constexpr int U = GROUP;
        if (U * FIT_BLOCK > n)
            return false;
        double px[U], py[U];
        if constexpr (STREAM) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                px[u] = xs[u * FIT_BLOCK + threadIdx.x];
                py[u] = ys[u * FIT_BLOCK + threadIdx.x];
            }
        } else {
            static_assert(PT_R >= U, "the first group lives in registers");
#pragma unroll
            for (int u = 0; u < U; u++) {
                px[u] = rx[u];
                py[u] = ry[u];
            }
        }
        fg(px, py, (int32_t)threadIdx.x);
        return true;
    }

    template <class FG>
    __device__ __forceinline__ void ragged_group(FG &&fg) const
    {
        constexpr int U = GROUP;
        const int32_t first = n / (U * FIT_BLOCK) * (U * FIT_BLOCK);
        if (first == n)
            return;
        double px[U], py[U];
        if constexpr (STREAM) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int32_t i = first + u * FIT_BLOCK + (int32_t)threadIdx.x;
                px[u] = xs[i < n - 1 ? i : n - 1];
                py[u] = ys[i < n - 1 ? i : n - 1];
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
                px[u] = py[u] = 0.0;
#pragma unroll
            for (int g = 0; g < PT_R / U; g++) {
                if (first == g * U * FIT_BLOCK) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        px[u] = rx[g * U + u];
                        py[u] = ry[g * U + u];
                    }
                }
            }
            if constexpr (PT_L > 0) {
                if (first >= PT_R * FIT_BLOCK) {
                    const int32_t j = first - PT_R * FIT_BLOCK + (int32_t)threadIdx.x;
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        px[u] = lds_dyn[2 * (j + u * FIT_BLOCK)];
                        py[u] = lds_dyn[2 * (j + u * FIT_BLOCK) + 1];
                    }
                }
            }
        }
        fg(px, py, first + (int32_t)threadIdx.x);
    }
};

// f(x, y, i, valid) for every point of the slice and, in the ragged group, the slots past its end
template <class View, class F>
__device__ __forceinline__ void each_point(const View &P, F &&f)
{
    P.full_groups([&](const double (&px)[View::GROUP], const double (&py)[View::GROUP], int32_t i0) {
#pragma unroll
        for (int u = 0; u < View::GROUP; u++)
            f(px[u], py[u], i0 + u * View::STRIDE, true);
    });
    P.ragged_group([&](const double (&px)[View::GROUP], const double (&py)[View::GROUP], int32_t i0) {
#pragma unroll
        for (int u = 0; u < View::GROUP; u++) {
            const int32_t i = i0 + u * View::STRIDE;
            f(px[u], py[u], i, i < P.n);
        }
    });
}

// Static LDS of a fit kernel, in doubles: two scratch areas for the workgroup
// reductions (up to 16 values x 16 waves each, used alternately) and two areas for
// totals (12 sums, the smallest and the largest distance, three model sums), and 16 doubles through which the
// first wave hands the others what it has worked out alone (the Kasa circle; the LM loop's next candidate, in
// the end its result).
constexpr int RED_SCR = 16 * 16, RED_AREA = 24, RED_TOT = 2 * RED_SCR, RED_SHARE = RED_TOT + 2 * RED_AREA,
              RED_STATE = RED_SHARE + 16, RED_SIZE = RED_STATE + LM_STATE_DOUBLES;

// Where the next reduction of a workgroup puts its partial sums.  Alternating
// between two scratch areas (and never writing totals that may still be read)
// lets a reduction do without a barrier on entry.
// Which wave does the work that falls to one wave - the circle from the Kasa moments, every step of the LM loop
// between two sweeps (several hundred dependent instructions each): a different wave each time (turn), starting
// from a wave that differs from workgroup to workgroup.  The four waves of a workgroup sit on the four SIMDs of
// its CU (the first wave on any of them: tools/ubench/wave_simd.hip); with one fixed wave for all of this work
// that wave's SIMD carried a third more instructions for the slice than the other three.
struct RedCtx {
    double *lds;
    double *lohi;               // where a sweep's reduction leaves the quads' extreme distances (LM kernels only)
    int count;
    int first_owner, turn, waves;
    __device__ __forceinline__ double *scratch() { return lds + (count++ & 1) * RED_SCR; }
    __device__ __forceinline__ double *totals(int which) { return lds + RED_TOT + RED_AREA * which; }
    __device__ __forceinline__ double *share() { return lds + RED_SHARE; }
    __device__ __forceinline__ double *state() { return lds + RED_STATE; }
    __device__ __forceinline__ int owner() const { return (first_owner + turn) % waves; }
    __device__ __forceinline__ void next_owner() { turn++; }
};

// doubles of LDS an LM kernel of B threads sets aside for RedCtx::lohi
constexpr int red_lohi_doubles(int B) { return 2 * (B / WAVE) * 16; }

__device__ __forceinline__ RedCtx make_red(double *lds, double *lohi = nullptr)
{
    const int waves = (int)(blockDim.x >> 6);
    return RedCtx{lds, lohi, 0, (int)(blockIdx.x % (unsigned)waves), 0, waves};
}

// circle_fitting.py:35-54.  Uniform result in every thread.
template <int NW, class View>
__device__ __forceinline__ int kasa_slice(const View &P, RedCtx &red, double out[3])
{
    out[0] = out[1] = out[2] = __longlong_as_double(0x7ff8000000000000ll);
    if (P.n < 3)
        return PCFIT_ST_SPARSE;
    const double n = (double)P.n;
    const int owner = red.owner();
    red.next_owner();
    const bool lead = (int)(threadIdx.x >> 6) == owner;
    double m[5];   // sum of u^2, v^2, u v, u z, v z about the mean (z = u^2 + v^2)
    double mx = 0.0, my = 0.0;
    // the power sums about the slice's first point: they came with the loads, or (a slice that is streamed) a
    // pass over the points takes them; one reduction, then the moments about the mean by the binomial
    // identities (a, b = the mean, seen from the first point)
    constexpr int NS = View::NSUM;
    double q[NS];
    if constexpr (View::STREAM) {
#pragma unroll
        for (int k = 0; k < NS; k++)
            q[k] = 0.0;
        const double x0 = P.x0, y0 = P.y0;
        each_point(P, [&](double px, double py, int32_t, bool valid) {
            View::power_sums(q, valid ? px - x0 : 0.0, valid ? py - y0 : 0.0);
        });
    } else {
#pragma unroll
        for (int k = 0; k < NS; k++)
            q[k] = P.m9[k];
    }
    double *tot = red.totals(1);
    block_reduce_to_lds<NS, false, NW, false>(q, 0.0, red.scratch(), tot, owner);
    if (lead) {
        const double su = tot[0], sv = tot[1], suu = tot[2], sz = tot[3], suv = tot[4], suz = tot[5], svz = tot[6];
        const double svv = sz - suu;
        const double a = su / n, b = sv / n;
        mx = P.x0 + a;
        my = P.y0 + b;
        const double zc = sz - a * su - b * sv;             // sum of z about the mean
        const double ab2 = a * a + b * b;
        m[0] = suu - a * su;
        m[1] = svv - b * sv;
        m[2] = suv - a * sv;
        m[3] = suz - 2.0 * a * suu - 2.0 * b * suv + ab2 * su - a * zc;
        m[4] = svz - 2.0 * b * svv - 2.0 * a * suv + ab2 * sv - b * zc;
    }
    // The circle from the moments (divisions, a square root: some 250 instructions) is the same in every thread.
    // The first wave - whose lanes have just written the totals, so it need not wait for anyone - works it out
    // and the others take it from LDS: their SIMDs are free for the workgroup next door meanwhile.
    double *share = red.share();
    if (lead) {
        const int st = kasa_from_moments(m, mx, my, n, out);
        if ((threadIdx.x & 63) == 0) {
            share[0] = out[0];
            share[1] = out[1];
            share[2] = out[2];
            share[3] = (double)st;
        }
    }
    __syncthreads();
    out[0] = share[0];
    out[1] = share[1];
    out[2] = share[2];
    return (int)share[3];
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
template <int NW, class View>
__device__ __forceinline__ bool lm_slice(const View &P, const LmParams prm, RedCtx &red, double par[3],
                                         int32_t iters[4], Phase &ph, double reach[2])
{
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const double delta = prm.delta;
    const double root_delta = sqrt(delta);
    // One sweep over the points at parameters (px0, py0, pr): the normal equations H, g of the
    // Huber-weighted problem, the curvature sum and the Huber cost, all from the same distances.
    //   acc[0..5] H00 H01 H02 H11 H12 H22, acc[6..8] g, acc[9] sum |psi|/r,
    //   acc[10] sum of squares of the small residuals, acc[11] linear part of the large ones
    //   acc[12..14] sums of q j0^2, q j0 j1, q (q = residual / distance): what lm_model_step_is_last needs
    // The totals of a sweep stay in LDS (tot: 12 sums, the smallest and the largest distance, the three model
    // sums): they are the same for every thread, and keeping two sets of them in registers would cost 48 VGPRs
    // per lane.
    //
    // Two forms of the pass.  The short one takes every point for a regular one: at a distance
    // above zero, within the Huber threshold, inside the slice; no branch, no select, and it notes
    // whether any point was not regular.  The general one handles all points with selects.  For a
    // regular point both give the same bits.  A sweep runs the short form over the full groups (and
    // the general one over the ragged group); when a point turns out not to be regular the sweep
    // is run again in the general form, and so are all later sweeps of this slice.  The totals are
    // left for the first wave to read (no barrier after they are written).
    // The first sweep (H and g at the start) leaves the three model sums out - five instructions per point: they
    // are read only when the candidate that follows is rejected, one slice in fifty, and that slice then goes on
    // without the second-order model (its totals are marked as not regular).
    bool general = false;
    auto sweep = [&](double px0, double py0, double pr, double *tot, auto with_model) {
        constexpr bool MODEL_SUMS = decltype(with_model)::value;
        constexpr int U = View::GROUP;
        double acc[15];
        double low, high;
        // (the model sums: for the ragged end of a sweep in the short form that takes them; a sweep wholly in the
        // general form is never marked regular and nobody reads them)
        auto any_point = [&](const double (&px)[U], const double (&py)[U], int32_t i0, bool check, auto model_tag) {
#pragma unroll
            for (int u = 0; u < U; u++)
                lm_point_general<15, decltype(model_tag)::value>(acc, low, high, px[u], py[u], px0, py0, pr, delta,
                                                                 root_delta, !check || i0 + u * View::STRIDE < P.n);
        };
        auto ragged = [&](const double (&px)[U], const double (&py)[U], int32_t i0) {
            if (general)
                any_point(px, py, i0, true, std::false_type{});
            else
                any_point(px, py, i0, true, std::integral_constant<bool, MODEL_SUMS>{});
        };
        if (!general) {
#pragma unroll
            for (int k = 0; k < 15; k++)
                acc[k] = 0.0;
            low = inf;
            high = -inf;
            // (a streamed slice: a wave that has met a point the short form is not made for stops there - the
            // sweep will be run again in the general form anyway - instead of reading the slice to its end)
            bool stop = false;
            P.full_groups([&](const double (&px)[U], const double (&py)[U], int32_t) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const double dx = px[u] - px0, dy = py[u] - py0;
                    const double d2 = fma(dx, dx, dy * dy);
                    // 1 / (sqrt(d2) + 1e-16) is taken as 1 / sqrt(d2): they differ by 1e-16 / ri relatively, below
                    // one rounding from ri = 0.5 on, and the form is left when a distance falls below 1e-6
                    const double iv = rsqrt_nr(d2);
                    const double ri = fma(d2, iv, 1e-16);
                    const double r = ri - pr;
                    high = max_no_nan(high, ri);
                    const double j0 = -dx * iv, j1 = -dy * iv;
                    acc[0] = fma(j0, j0, acc[0]);
                    acc[1] = fma(j0, j1, acc[1]);
                    acc[2] = acc[2] - j0;
                    acc[4] = acc[4] - j1;
                    acc[6] = fma(j0, r, acc[6]);
                    acc[7] = fma(j1, r, acc[7]);
                    acc[8] = acc[8] - r;
                    // (the first sweep's cost and curvature sum are never read: the loop starts from a last cost
                    // of infinity, takes the first candidate whatever it costs, and the bounds that use the
                    // curvature wait for a finite last cost)
                    if constexpr (MODEL_SUMS) {
                        acc[10] = fma(r, r, acc[10]);
                        acc[9] = fma(fabs(r), iv, acc[9]);
                        const double q = r * iv, t0 = j0 * q;
                        acc[12] = fma(t0, j0, acc[12]);
                        acc[13] = fma(t0, j1, acc[13]);
                        acc[14] = acc[14] + q;
                    }
                    low = min_no_nan(low, ri);
                }
                if constexpr (View::STREAM)
                    stop = !(max_no_nan(high - pr, pr - low) <= delta) || low < 1e-6;
            }, View::STREAM ? &stop : nullptr);
            // a point was not regular: a residual beyond the threshold, or a NaN (which a distance of
            // zero gives here as well) that the sum of squares has kept
            // (the largest |residual| sits at the largest or at the smallest distance; without points -inf)
            const double amax = max_no_nan(high - pr, pr - low);
            // (the sum of the residuals has kept any NaN as well as the sum of their squares)
            const bool odd = !(amax <= delta) || acc[8] != acc[8] || low < 1e-6;
            acc[5] = (double)P.full_points_per_thread();      // the sum of that many ones (j2 = -1)
            acc[3] = acc[5] - acc[0];                           // j0^2 + j1^2 = d2 / d2 = 1 for every point
            P.ragged_group(ragged);
            low = odd ? -inf : low;             // the smallest distance doubles as the flag
            ph.mark(2);
            block_reduce_to_lds2<15, NW, 12, false>(acc, low, high, red.scratch(), red.lohi, tot, red.owner());
            ph.mark(3);
            return;
        }
#pragma unroll
        for (int k = 0; k < 15; k++)
            acc[k] = 0.0;
        low = inf;
        high = -inf;
        P.full_groups([&](const double (&px)[U], const double (&py)[U], int32_t i0) { any_point(px, py, i0, false, std::false_type{}); });
        P.ragged_group(ragged);
        ph.mark(2);
        block_reduce_to_lds2<15, NW, 12, false>(acc, low, high, red.scratch(), red.lohi, tot, red.owner());
        ph.mark(3);
    };

    // (one call site for the sweep: its body is the bulk of the kernel's code)
    // The loop between the sweeps - 3x3 solves, the stopping rules, the model of the last step: several hundred
    // instructions on values every thread would hold alike - runs on one wave, another one after every sweep
    // (RedCtx::owner), which finds the loop's state in LDS where the last one left it.  The others wait at the
    // barrier for the next candidate (in the end: for the result), their SIMDs free for the workgroup next door.
    double *share = red.share();
    double *state = red.state();
    int32_t *ishare = reinterpret_cast<int32_t *>(share + 10);
    {
        LmState st;
        lm_begin(st, par, prm);
        if (st.done) {                              // (max_iter <= 0: the start is the result)
            lm_finish(st, par, iters);
            reach[0] = reach[1] = 0.0;
            return false;
        }
        if (threadIdx.x == 0)
            lm_state_store(st, state);              // (read after the first sweep's barrier)
    }
    double tx = par[0], ty = par[1], tr = par[2];
    int into = 0;
    bool done = false;
    bool first_sweep = true;
    while (!done) {
        const int owner = red.owner();              // the wave that forms this sweep's totals and takes the step
        if (first_sweep)
            sweep(tx, ty, tr, red.totals(into), std::false_type{});
        else
            sweep(tx, ty, tr, red.totals(into), std::true_type{});
        red.next_owner();
        if ((int)(threadIdx.x >> 6) == owner) {
            LmState st;
            lm_state_load(st, state);
            // (a short-form sweep that met a point it is not made for says so in place of the smallest distance)
            const bool again = !general && red.totals(into)[12] == -inf;
            if (!again)
                lm_step<RED_AREA, true>(st, prm, red.totals(0), into, !general && !first_sweep);
            if ((threadIdx.x & 63) == 0) {
                ishare[7] = again ? 1 : 0;
                share[0] = st.tx;
                share[1] = st.ty;
                share[2] = st.tr;
                ishare[0] = lm_into(st);
                ishare[1] = st.done ? 1 : 0;
                if (st.done) {
                    share[3] = st.cx;
                    share[4] = st.cy;
                    share[5] = st.rad;
                    share[6] = st.min_ri;
                    share[7] = st.max_ri;
                    int32_t r4[4];
                    double unused[3];
                    lm_finish(st, unused, r4);
                    ishare[2] = r4[0];
                    ishare[3] = r4[1];
                    ishare[4] = r4[2];
                    ishare[5] = r4[3];
                    ishare[6] = st.cur >= 0 ? 1 : 0;
                }
                if (!again)
                    lm_state_store(st, state);
            }
        }
        __syncthreads();
        if (ishare[7]) {
            general = true;                         // the same sweep once more, in the general form
            continue;                               // (the word is written again only after the sweep's barrier)
        }
        first_sweep = false;
        tx = share[0];
        ty = share[1];
        tr = share[2];
        into = ishare[0];
        done = ishare[1] != 0;
    }
    par[0] = share[3];
    par[1] = share[4];
    par[2] = share[5];
#pragma unroll
    for (int k = 0; k < 4; k++)
        iters[k] = ishare[2 + k];
    ph.mark(4);
    // bounds of the smallest (from above) and the largest (from below) distance (sqrt + 1e-16) of the slice's points
    // from the returned centre: known when a sweep ran at these parameters or one modelled step before them
    reach[0] = share[6];
    reach[1] = share[7];
    return ishare[6] != 0;
}

template <int PT_R, int PT_L, int FIT_BLOCK>
__global__ __launch_bounds__(FIT_BLOCK, (PT_L == 10 ? 3 : 2)) void kasa_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                        const int64_t *__restrict__ offsets, double *__restrict__ out,
                                                        int32_t *__restrict__ status)
{
    __shared__ double lds[RED_SIZE];
    RedCtx red = make_red(lds);
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT_R, PT_L, FIT_BLOCK> P;
    P.load(x, y, off, n);
    double c[3];
    const int st = kasa_slice<FIT_BLOCK / WAVE>(P, red, c);
    if (threadIdx.x == 0) {
        out[3 * s] = c[0];
        out[3 * s + 1] = c[1];
        out[3 * s + 2] = c[2];
        if (status)
            status[s] = st;
    }
}

template <int PT_R, int PT_L, int FIT_BLOCK>
__global__ __launch_bounds__(FIT_BLOCK, 2) void lm_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                      const int64_t *__restrict__ offsets,
                                                      const double *__restrict__ init, LmParams prm,
                                                      double *__restrict__ out, int32_t *__restrict__ iters)
{
    __shared__ double lds[RED_SIZE];
    __shared__ double lohi[red_lohi_doubles(FIT_BLOCK)];
    Phase ph;
    ph.init();
    RedCtx red = make_red(lds, lohi);
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT_R, PT_L, FIT_BLOCK> P;
    P.load(x, y, off, n);
    double c[3] = {init[3 * s], init[3 * s + 1], init[3 * s + 2]};
    int32_t it[4] = {0, 0, 0, 0};
    double reach[2];
    lm_slice<FIT_BLOCK / WAVE>(P, prm, red, c, it, ph, reach);
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
template <int PT_R, int PT_L, int FIT_BLOCK>
__global__ __launch_bounds__(FIT_BLOCK, 2) void fit_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                       const int64_t *__restrict__ offsets, LmParams prm,
                                                       double *__restrict__ kasa_out, double *__restrict__ out,
                                                       int32_t *__restrict__ iters, int32_t *__restrict__ status)
{
    __shared__ double lds[RED_SIZE];
    __shared__ double lohi[red_lohi_doubles(FIT_BLOCK)];
    Phase ph;
    ph.init();
    RedCtx red = make_red(lds, lohi);
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT_R, PT_L, FIT_BLOCK> P;
    P.load(x, y, off, n);
    double c[3];
    const int st = kasa_slice<FIT_BLOCK / WAVE>(P, red, c);
    if (threadIdx.x == 0 && kasa_out) {
        kasa_out[3 * s] = c[0];
        kasa_out[3 * s + 1] = c[1];
        kasa_out[3 * s + 2] = c[2];
    }
    int32_t it[4] = {0, 0, 0, 0};
    double reach[2];
    if (st != PCFIT_ST_SPARSE)
        lm_slice<FIT_BLOCK / WAVE>(P, prm, red, c, it, ph, reach);
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

// butterfly over the wave: every lane ends with the wave's (largest radius, first index) and (smallest, first index)
template <int M>
__device__ __forceinline__ void fold_extremes(Extreme2 &mx, Extreme2 &mn, int lane)
{
    if constexpr (M >= 1) {
        Extreme2 o;
        o.r = lane_xor<M>(mx.r, lane);
        o.i = lane_xor_i32<M>(mx.i, lane);
        mx = better_max(mx, o);
        o.r = lane_xor<M>(mn.r, lane);
        o.i = lane_xor_i32<M>(mn.i, lane);
        mn = better_min(mn, o);
        fold_extremes<M / 2>(mx, mn, lane);
    }
}

// Number of copies of the LDS histogram (lanes spread their updates over them by lane number, which
// cuts the serialisation of lanes that hit one counter) and the distance between copies, in counters
// (the extra 8 put the copies on different banks).
__host__ __device__ __forceinline__ int hist_copies(int32_t H) { return H <= 512 ? 4 : 1; }
__host__ __device__ __forceinline__ int hist_stride(int32_t H) { return H + 8; }

// Residual statistics of one slice about the circle (cx, cy, R): histogram of
// dr = hypot - R (numpy.histogram rules, H bins), Rmax / Rmin and the first
// index attaining each (analyzer.py:83-92, gui.py:979-983).  The H+1 edges are
// in LDS at lds_dyn[base ..], the int32 counters (hist_copies(H) copies,
// hist_stride(H) apart, zeroed by the caller) follow.
//
// Results are those of the correctly rounded hypot (cr_hypot) for every point,
// but only the points that need it pay for it.  A point's radius is first taken
// from the reciprocal square root (r1, within 4.7e-16 of the exact value,
// relatively); its bin follows from t = (r1 - R - e0) * H / (eH - e0) whenever t
// is farther than 2^-20 from an integer - the caller has checked that the edges
// are uniform to a quarter of that (edges_uniform), and the errors of r1 and of
// t stay below another quarter - and it can only be an extreme when r1 comes
// within 3e-15 of the slice's largest / smallest distance from the centre.  Those
// two are known from the last LM sweep when it ran at exactly this centre
// (reach); otherwise the largest and smallest r1 of the slice's first group of
// points stand in for them (every point beyond them is then examined).  The
// points for which either test fails (a handful of 8192) go through the exact
// code: cr_hypot, numpy's guess-and-repair against the edges, and the (radius,
// first index) comparison.  Slices whose parameters leave no such margins, and
// the ragged end of every slice, take the exact code for all points.
// `also`: what else wants to see every point of the slice on this sweep (the angular-bin boundary of
// boundary_residual_kernel): also.quick(dx, dy, r1, u) for point u of a group on the quick road, with its offsets
// from the centre and its quick radius at hand, then also.quick_done(i0, stride) once per group (point u has index
// i0 + u * stride); also.any(px, py, i, valid) for a point of the exact road.  Every point of the slice is handed
// over exactly once.
struct NoOtherUse {
    __device__ __forceinline__ void quick(double, double, double, int) const {}
    __device__ __forceinline__ void quick_done(int32_t, int32_t) const {}
    __device__ __forceinline__ void any(double, double, int32_t, bool) const {}
};

template <class View, class Also = NoOtherUse>
__device__ __forceinline__ void residual_slice(const View &P, double cx, double cy, double R, int32_t H, int32_t base,
                                               int64_t s, int64_t *__restrict__ hist, double *__restrict__ rmax_rmin,
                                               int64_t *__restrict__ argmax_argmin, Extreme2 *wmax, Extreme2 *wmin,
                                               double *wthr /* 2 * 16 doubles */, bool edges_uniform, bool have_reach,
                                               const double reach[2], Also &&also = Also{})
{
    using mask_t = unsigned long long;
    const double e0 = H > 0 ? lds_dyn[base] : 0.0, eH = H > 0 ? lds_dyn[base + H] : 0.0;
    // numpy.histogram puts d into the bin b with edges[b] <= d < edges[b+1] (last bin closed): it
    // guesses b by scaling and repairs the guess against the edges.  Any guess within one bin of the
    // truth gives the same b, so the guess here multiplies by H / span instead of dividing.
    const double scale = (double)H / (eH - e0);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int32_t cnt0 = 2 * (base + H + 1) + (lane & (hist_copies(H) - 1)) * hist_stride(H);   // this lane's copy
    Extreme2 mx{-inf, -1}, mn{inf, -1};
    constexpr int U = View::GROUP;
    // the exact code for one point
    auto exact_point = [&](double px, double py, int32_t i, bool valid, bool want_hist) {
        const double r = cr_hypot(px - cx, py - cy);
        const bool up = valid & (r > mx.r), dn = valid & (r < mn.r);       // strict: the first index stays
        mx.r = up ? r : mx.r;
        mn.r = dn ? r : mn.r;
        mx.i = up ? i : mx.i;
        mn.i = dn ? i : mn.i;
        if (want_hist) {
            // (the LDS arrays are named through lds_dyn here: a captured pointer loses its address space)
            const double d = r - R;
            const bool in = valid & (d >= e0) & (d <= eH);
            const int32_t guess = (int32_t)((d - e0) * scale);
            const int32_t g = max(0, min(guess, H - 1));
            const double below = lds_dyn[base + g], above = lds_dyn[base + g + 1];
            const bool under = d < below, over = !under & (g < H - 1) & (d >= above);
            const int32_t b = g + (over ? 1 : 0) - (under ? 1 : 0);
            if (in)
                atomicAdd(reinterpret_cast<int32_t *>(lds_dyn) + cnt0 + b, 1);
        }
    };
    auto exact_group = [&](const double (&px)[U], const double (&py)[U], int32_t i0, bool check) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t i = i0 + u * View::STRIDE;
            exact_point(px[u], py[u], i, !check || i < P.n, H > 0);
            also.any(px[u], py[u], i, !check || i < P.n);
        }
    };

    constexpr double TAU = 9.5367431640625e-07;              // 2^-20, in bins
    // the margins of the bin test: uniform edges, and the errors of r1 (4.7e-16 r) and of the fused
    // multiply-add that makes t within a quarter of TAU, in bins
    bool fast = true;
    if (H > 0) {
        const double span = fabs(R) + fmax(fabs(e0), fabs(eH));
        fast = edges_uniform && scale > 0.0 && scale < inf && span * scale * 8.9e-16 <= 0.25 * TAU;
    }
    // what an extreme has to reach (uniform over the workgroup)
    double t_hi, t_lo;
    if (have_reach) {
        t_hi = (reach[1] - 2e-16) * (1.0 - 3e-15);
        t_lo = (reach[0] + 2e-16) * (1.0 + 3e-15);
    } else {
        double m_hi = -inf, m_lo = inf;
        const bool sampled = fast && P.first_group([&](const double (&px)[U], const double (&py)[U], int32_t) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double dx = px[u] - cx, dy = py[u] - cy;
                const double d2 = fma(dx, dx, dy * dy);
                const double r1 = d2 * rsqrt_nr(d2);
                m_hi = max_no_nan(m_hi, r1);
                m_lo = min_no_nan(m_lo, r1);
            }
        });
        if (sampled) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                m_hi = max_no_nan(m_hi, __shfl_xor(m_hi, m, WAVE));
                m_lo = min_no_nan(m_lo, __shfl_xor(m_lo, m, WAVE));
            }
            if (lane == 0) {
                wthr[wave] = m_hi;
                wthr[16 + wave] = m_lo;
            }
            __syncthreads();
            for (int w = 0; w < nw; w++) {
                m_hi = max_no_nan(m_hi, wthr[w]);
                m_lo = min_no_nan(m_lo, wthr[16 + w]);
            }
        }
        fast = sampled;
        t_hi = m_hi * (1.0 - 2e-15);
        t_lo = m_lo * (1.0 + 2e-15);
    }
    if (fast) {
        const double c0 = -(R + e0) * scale;
        auto pass = [&](const double (&px)[U], const double (&py)[U], int32_t i0, auto hist_tag) {
            constexpr bool HIST = decltype(hist_tag)::value;
            // conditions live as wave masks (scalar registers): one mask per point of the group
            mask_t odd[U];
            int32_t b[U];
            mask_t any = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double dx = px[u] - cx, dy = py[u] - cy;
                const double d2 = fma(dx, dx, dy * dy);
                const double r1 = d2 * rsqrt_nr(d2);
                also.quick(dx, dy, r1, u);
                // (negated compares: a NaN radius - a point on the centre, an overflow - goes to the exact code)
                mask_t o = __builtin_amdgcn_ballot_w64(!(r1 < t_hi)) | __builtin_amdgcn_ballot_w64(!(r1 > t_lo));
                if constexpr (HIST) {
                    const double t = fma(r1, scale, c0);
                    const double ft = floor(t);
                    o |= __builtin_amdgcn_ballot_w64(fabs((t - ft) - 0.5) > 0.5 - TAU);
                    // counter H of a copy is nobody's bin: points outside the range, and the points that
                    // the exact code counts, add their one there (no branch around the update)
                    const uint32_t bin = min((uint32_t)(int32_t)ft, (uint32_t)H);
                    b[u] = __builtin_amdgcn_inverse_ballot_w64(o) ? H : (int32_t)bin;
                }
                odd[u] = o;
                any |= o;
            }
            also.quick_done(i0, View::STRIDE);
            if (any) {
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (__builtin_amdgcn_inverse_ballot_w64(odd[u]))
                        exact_point(px[u], py[u], i0 + u * View::STRIDE, true, HIST);
            }
            if constexpr (HIST) {
#pragma unroll
                for (int u = 0; u < U; u++)
                    atomicAdd(reinterpret_cast<int32_t *>(lds_dyn) + cnt0 + b[u], 1);
            }
        };
        if (H > 0)
            P.full_groups([&](const double (&px)[U], const double (&py)[U], int32_t i0) { pass(px, py, i0, std::true_type{}); });
        else
            P.full_groups([&](const double (&px)[U], const double (&py)[U], int32_t i0) { pass(px, py, i0, std::false_type{}); });
    } else {
        P.full_groups([&](const double (&px)[U], const double (&py)[U], int32_t i0) { exact_group(px, py, i0, false); });
    }
    P.ragged_group([&](const double (&px)[U], const double (&py)[U], int32_t i0) { exact_group(px, py, i0, true); });
    // The wave's extremes.  On the quick road a lane holds a candidate only when one of its points went through
    // the exact code - the slice's extremes themselves and little else - so a wave has none, or one lane has
    // one: that lane writes it and the six-step fold of (radius, index) pairs over the lanes, some 170
    // instructions a thread, is left to the waves where several lanes hold candidates.
    {
        const mask_t has_max = __builtin_amdgcn_ballot_w64(mx.i >= 0), has_min = __builtin_amdgcn_ballot_w64(mn.i >= 0);
        if (__popcll(has_max) > 1 || __popcll(has_min) > 1) {
            fold_extremes<32>(mx, mn, lane);
            if (lane == 0) {
                wmax[wave] = mx;
                wmin[wave] = mn;
            }
        } else {
            // (a lane without a candidate holds (-inf, -1) / (+inf, -1), which is what a wave without one reports)
            const int at_max = has_max ? (int)__builtin_ctzll(has_max) : 0, at_min = has_min ? (int)__builtin_ctzll(has_min) : 0;
            if (lane == at_max)
                wmax[wave] = mx;
            if (lane == at_min)
                wmin[wave] = mn;
        }
    }
    __syncthreads();
    const int32_t *lhist = reinterpret_cast<const int32_t *>(lds_dyn) + 2 * (base + H + 1);
    for (int b = threadIdx.x; b < H; b += blockDim.x) {
        int32_t c = 0;
        for (int k = 0; k < hist_copies(H); k++)
            c += lhist[k * hist_stride(H) + b];
        hist[s * H + b] = (int64_t)c;
    }
    if (threadIdx.x == 0) {
        mx = wmax[0];
        mn = wmin[0];
        for (int w = 1; w < nw; w++) {
            mx = better_max(mx, wmax[w]);
            mn = better_min(mn, wmin[w]);
        }
        rmax_rmin[2 * s] = mx.r;
        rmax_rmin[2 * s + 1] = mn.r;
        argmax_argmin[2 * s] = mx.i;
        argmax_argmin[2 * s + 1] = mn.i;
    }
}

// Kasa -> LM -> residual statistics with the slice read from HBM once.
// Dynamic LDS: the view's points, then (H+1) doubles of edges, then H int32 counters.
// WITH_LM false: the kernel for max_iter = 0 (the Kasa circle is the result).  Without the LM loop's code the
// working set beside the points is some 60 registers, so the shape with 22 points of a thread in registers and
// 10 in LDS (40 KB) runs three workgroups to a CU: the pass is bound by the latency of a workgroup's chain of
// load, reduction, circle and residual sweep, not by arithmetic, and a third workgroup hides more of it.
// GIVEN (with WITH_LM false): no fit at all - the circle of slice s is given[3 s ..] and only the residual
// statistics are taken (the streamed shape: one pass over the slice, the loads of two trips in flight).
template <int PT_R, int PT_L, int FIT_BLOCK, bool WITH_LM, bool GIVEN = false>
__global__ __launch_bounds__(FIT_BLOCK, (GIVEN ? 4 : !WITH_LM && PT_L == 10 ? 3 : 2)) void fit_residual_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets, LmParams prm,
    double *__restrict__ kasa_out, double *__restrict__ out, int32_t *__restrict__ iters,
    int32_t *__restrict__ status, int32_t H, const double *__restrict__ edges, int64_t *__restrict__ hist,
    double *__restrict__ rmax_rmin, int64_t *__restrict__ argmax_argmin, const double *__restrict__ given)
{
    static_assert(!GIVEN || !WITH_LM, "a given circle is not refined");
    __shared__ double lds[RED_SIZE];
    __shared__ double lohi[WITH_LM ? red_lohi_doubles(FIT_BLOCK) : 1];
    __shared__ Extreme2 wmax[16], wmin[16];
    __shared__ double wthr[2 * 16];
    __shared__ int edges_off;           // set when the edges are not uniformly spaced
    Phase ph;
    ph.init();
    RedCtx red = make_red(lds, lohi);
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    SliceView<PT_R, PT_L, FIT_BLOCK> P;
    constexpr int base = SliceView<PT_R, PT_L, FIT_BLOCK>::LDS_DOUBLES;
    // The edges a thread will put into LDS are asked for before the slice, so that they arrive under the slice's
    // loads instead of costing a trip to memory of their own after them (up to 512 bins here; more: below).
    constexpr int EDGE_AHEAD = 2;
    double edge_ahead[EDGE_AHEAD];
    const double e0 = H > 0 ? edges[0] : 0.0, eH = H > 0 ? edges[H] : 0.0;      // (uniform addresses: scalar loads)
#pragma unroll
    for (int k = 0; k < EDGE_AHEAD; k++) {
        const int b = threadIdx.x + k * FIT_BLOCK;
        edge_ahead[k] = H > 0 ? edges[b <= H ? b : H] : 0.0;
    }
    if (threadIdx.x == 0)
        edges_off = 0;
    if (H > 0)
        for (int b = threadIdx.x; b < hist_copies(H) * hist_stride(H); b += FIT_BLOCK)
            reinterpret_cast<int32_t *>(lds_dyn + base + H + 1)[b] = 0;
    P.load(x, y, off, n);
    if constexpr (PT_L == 0)
        __syncthreads();                // (edges_off is cleared before anyone may set it; with points in LDS load() ends on a barrier)
    ph.mark(0);
    if (H > 0) {
        // uniform to 2^-22 of a bin: what lets residual_slice take a bin number from a product
        const double width = (eH - e0) / (double)H, per_bin = (double)H / (eH - e0);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < EDGE_AHEAD; k++) {
            const int b = threadIdx.x + k * FIT_BLOCK;
            if (b <= H) {
                lds_dyn[base + b] = edge_ahead[k];
                ok &= fabs(edge_ahead[k] - fma((double)b, width, e0)) * per_bin <= 2.384185791015625e-07;
            }
        }
        for (int b = threadIdx.x + EDGE_AHEAD * FIT_BLOCK; b <= H; b += FIT_BLOCK) {
            const double e = edges[b];
            lds_dyn[base + b] = e;
            ok &= fabs(e - fma((double)b, width, e0)) * per_bin <= 2.384185791015625e-07;
        }
        if (!ok)
            edges_off = 1;
    }
    double c[3];
    int st = PCFIT_ST_OK;
    if constexpr (GIVEN) {
        c[0] = given[3 * s];
        c[1] = given[3 * s + 1];
        c[2] = given[3 * s + 2];
    } else {
        st = kasa_slice<FIT_BLOCK / WAVE>(P, red, c);
        if (threadIdx.x == 0 && kasa_out) {
            kasa_out[3 * s] = c[0];
            kasa_out[3 * s + 1] = c[1];
            kasa_out[3 * s + 2] = c[2];
        }
    }
    ph.mark(1);
    int32_t it[4] = {0, 0, 0, 0};
    double reach[2] = {0.0, 0.0};
    bool have_reach = false;
    if constexpr (WITH_LM) {
        if (st != PCFIT_ST_SPARSE)
            have_reach = lm_slice<FIT_BLOCK / WAVE>(P, prm, red, c, it, ph, reach);
    }
    __syncthreads();
    residual_slice(P, c[0], c[1], c[2], H, base, s, hist, rmax_rmin, argmax_argmin, wmax, wmin, wthr, edges_off == 0,
                   have_reach, reach);
    ph.mark(5);
    if (threadIdx.x == 0 && !GIVEN) {
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

// Launch shape for a batch whose longest slice has max_len points: threads per
// workgroup, points per thread in registers (4 VGPRs each) and in LDS (16 bytes
// each).  All zero points = stream the slice from global memory on every pass.
struct FitShape {
    int block, pr, pl;
};

static FitShape pick_shape(int64_t max_len, bool three_per_cu = false)
{
    // (three workgroups to a CU: 22 + 10 points per thread; only the kernel without the LM loop has the registers)
    static const int allow3 = [] { const char *e = getenv("PCFIT_FIT_THREE_PER_CU"); return e ? atoi(e) : 1; }();
    if (three_per_cu && allow3 && max_len > 4096 && max_len <= 8192) return {256, 22, 10};
    if (max_len <= 0) return {256, 0, 0};
    if (max_len <= 64) return {64, 1, 0};
    if (max_len <= 256) return {256, 1, 0};
    if (max_len <= 1024) return {256, 4, 0};
    if (max_len <= 2048) return {256, 8, 0};
    if (max_len <= 4096) return {256, 16, 0};
    // (512 threads with 8 + 8 points each, four waves per SIMD at 128 VGPRs, measured 20 % slower)
    if (max_len <= 8192) return {256, 16, 16};
    return {256, 0, 0};
}

template <int B, int PR, int PL>
struct ShapeTag {
    static constexpr int block = B, pr = PR, pl = PL;
    static constexpr size_t lds_points = (size_t)PL * B * 16;
};

// calls f(ShapeTag<block, pr, pl>); WITH_3: the caller has a kernel for the three-to-a-CU shape
template <bool WITH_3 = false, class F>
static void dispatch_shape(FitShape sh, F &&f)
{
    const int key = (sh.block * 100 + sh.pr) * 100 + sh.pl;
    switch (key) {
    case (64 * 100 + 1) * 100: f(ShapeTag<64, 1, 0>{}); break;
    case (256 * 100 + 1) * 100: f(ShapeTag<256, 1, 0>{}); break;
    case (256 * 100 + 4) * 100: f(ShapeTag<256, 4, 0>{}); break;
    case (256 * 100 + 8) * 100: f(ShapeTag<256, 8, 0>{}); break;
    case (256 * 100 + 16) * 100: f(ShapeTag<256, 16, 0>{}); break;
    case (256 * 100 + 16) * 100 + 16: f(ShapeTag<256, 16, 16>{}); break;
    case (256 * 100 + 22) * 100 + 10:
        if constexpr (WITH_3) {
            f(ShapeTag<256, 22, 10>{});
            break;
        }
        [[fallthrough]];
    default: f(ShapeTag<256, 0, 0>{}); break;
    }
}

// dynamic LDS beyond 64 KB has to be asked for, per kernel
template <class K>
static int allow_lds(K kern, size_t bytes)
{
    if (bytes > 48 * 1024)
        PCFIT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)bytes));
    return PCFIT_OK;
}

int launch_kasa(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len, double *out,
                int32_t *status, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    int rc = PCFIT_OK;
    dispatch_shape<true>(pick_shape(max_len, true), [&](auto t) {
        using T = decltype(t);
        auto kern = kasa_kernel<T::pr, T::pl, T::block>;
        if ((rc = allow_lds(kern, T::lds_points)) != PCFIT_OK) return;
        kern<<<dim3((unsigned)S), dim3(T::block), T::lds_points, st>>>(x, y, offsets, out, status);
    });
    PCFIT_TRY(rc);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_lm(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
              const double *init, const LmParams &prm, double *out, int32_t *iters, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    int rc = PCFIT_OK;
    dispatch_shape(pick_shape(max_len), [&](auto t) {
        using T = decltype(t);
        auto kern = lm_kernel<T::pr, T::pl, T::block>;
        if ((rc = allow_lds(kern, T::lds_points)) != PCFIT_OK) return;
        kern<<<dim3((unsigned)S), dim3(T::block), T::lds_points, st>>>(x, y, offsets, init, prm, out, iters);
    });
    PCFIT_TRY(rc);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_fit(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
               const LmParams &prm, double *kasa_out, double *out, int32_t *iters, int32_t *status, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    int rc = PCFIT_OK;
    dispatch_shape(pick_shape(max_len), [&](auto t) {
        using T = decltype(t);
        auto kern = fit_kernel<T::pr, T::pl, T::block>;
        if ((rc = allow_lds(kern, T::lds_points)) != PCFIT_OK) return;
        kern<<<dim3((unsigned)S), dim3(T::block), T::lds_points, st>>>(x, y, offsets, prm, kasa_out, out, iters, status);
    });
    PCFIT_TRY(rc);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

int launch_fit_residual(const double *x, const double *y, const int64_t *offsets, int64_t S, int64_t max_len,
                        const LmParams &prm, double *kasa_out, double *out, int32_t *iters, int32_t *status, int32_t H,
                        const double *edges, int64_t *hist, double *rmm, int64_t *amm, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    const size_t lds_hist = H > 0 ? (size_t)(H + 1) * 8 + (size_t)hist_copies(H) * hist_stride(H) * 4 : 0;
    int rc = PCFIT_OK;
    if (prm.max_iter == 0) {
        dispatch_shape<true>(pick_shape(max_len, true), [&](auto t) {
            using T = decltype(t);
            auto kern = fit_residual_kernel<T::pr, T::pl, T::block, false>;
            if ((rc = allow_lds(kern, T::lds_points + lds_hist)) != PCFIT_OK) return;
            kern<<<dim3((unsigned)S), dim3(T::block), T::lds_points + lds_hist, st>>>(
                x, y, offsets, prm, kasa_out, out, iters, status, H, edges, hist, rmm, amm, nullptr);
        });
    } else {
        dispatch_shape(pick_shape(max_len), [&](auto t) {
            using T = decltype(t);
            auto kern = fit_residual_kernel<T::pr, T::pl, T::block, true>;
            if ((rc = allow_lds(kern, T::lds_points + lds_hist)) != PCFIT_OK) return;
            kern<<<dim3((unsigned)S), dim3(T::block), T::lds_points + lds_hist, st>>>(
                x, y, offsets, prm, kasa_out, out, iters, status, H, edges, hist, rmm, amm, nullptr);
        });
    }
    PCFIT_TRY(rc);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

// residual statistics about given circles, no per-point output: the streamed shape of the fused kernel
int launch_residual_stats(const double *x, const double *y, const int64_t *offsets, int64_t S, const double *cxcyr,
                          int32_t H, const double *edges, int64_t *hist, double *rmm, int64_t *amm, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    const size_t lds_hist = H > 0 ? (size_t)(H + 1) * 8 + (size_t)hist_copies(H) * hist_stride(H) * 4 : 0;
    auto kern = fit_residual_kernel<0, 0, 256, false, true>;
    PCFIT_TRY(allow_lds(kern, lds_hist));
    const LmParams prm{0.0, 0.0, 0.0, 0.0, 0};
    kern<<<dim3((unsigned)S), dim3(256), lds_hist, st>>>(x, y, offsets, prm, nullptr, nullptr, nullptr, nullptr, H, edges,
                                                         hist, rmm, amm, cxcyr);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

// boundary_by_angle_max (boundary.py:7-42) and the residual statistics (gui.py:972-987, analyzer.py:83-92) about
// the same circle in ONE sweep over the slice: both start from a point's offsets from the centre and its quick
// radius, and a slice of a scan-sized window is far too long to stay on chip between two kernels.  The streamed
// shape of the residual pass (loads of two trips in flight) with the angular bins of angle_math.h riding along.
// Dynamic LDS: (H+1) doubles of edges, the histogram counters, then angle_bins_lds_bytes(bins).
struct BoundaryAlso {
    AngleBins *ab;
    AngleBins::Probe pr[4];
    __device__ __forceinline__ void quick(double dx, double dy, double r1, int u) { pr[u] = ab->probe(dx, dy, r1, true); }
    __device__ __forceinline__ void quick_done(int32_t i0, int32_t stride)
    {
#pragma unroll
        for (int u = 0; u < 4; u++)
            ab->settle(pr[u], (uint32_t)(i0 + u * stride), true);
    }
    __device__ __forceinline__ void any(double px, double py, int32_t i, bool valid) const
    {
        const double dx = px - ab->cx, dy = py - ab->cy;
        const double d2 = fma(dx, dx, dy * dy);
        ab->point(dx, dy, d2 * rsqrt_nr(d2), (uint32_t)i, valid);
    }
};

__host__ __device__ inline size_t hist_lds_bytes(int32_t H)
{
    return H > 0 ? (size_t)(H + 1) * 8 + (((size_t)hist_copies(H) * hist_stride(H) * 4 + 7) & ~(size_t)7) : 0;
}

__global__ __launch_bounds__(256, 3) void boundary_residual_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const int64_t *__restrict__ offsets,
    const double *__restrict__ cxcyr, int32_t bins, int64_t *__restrict__ sel, int32_t H,
    const double *__restrict__ edges, int64_t *__restrict__ hist, double *__restrict__ rmax_rmin,
    int64_t *__restrict__ argmax_argmin)
{
    __shared__ Extreme2 wmax[16], wmin[16];
    __shared__ double wthr[2 * 16];
    __shared__ int edges_off;
    constexpr int BLOCK = 256;
    const int64_t s = blockIdx.x;
    const int64_t off = offsets[s], n = offsets[s + 1] - off;
    const double cx = cxcyr[3 * s], cy = cxcyr[3 * s + 1], R = cxcyr[3 * s + 2];
    if (threadIdx.x == 0)
        edges_off = 0;
    AngleBins ab;
    ab.init(reinterpret_cast<char *>(lds_dyn) + hist_lds_bytes(H), bins, cx, cy);
    if (H > 0)
        for (int b = threadIdx.x; b < hist_copies(H) * hist_stride(H); b += BLOCK)
            reinterpret_cast<int32_t *>(lds_dyn + H + 1)[b] = 0;
    SliceView<0, 0, BLOCK> P;
    P.load(x, y, off, n);
    __syncthreads();
    if (H > 0) {
        const double e0 = edges[0], eH = edges[H];
        const double width = (eH - e0) / (double)H, per_bin = (double)H / (eH - e0);
        bool ok = true;
        for (int b = threadIdx.x; b <= H; b += BLOCK) {
            const double e = edges[b];
            lds_dyn[b] = e;
            ok &= fabs(e - fma((double)b, width, e0)) * per_bin <= 2.384185791015625e-07;
        }
        if (!ok)
            edges_off = 1;
    }
    // the first stretch of the slice raises the angular bins' bounds only
    {
        const int32_t n_seed = P.n < ANGLE_SEED_POINTS ? P.n : ANGLE_SEED_POINTS;
        for (int32_t i = threadIdx.x; i < n_seed; i += BLOCK) {
            const double dx = P.xs[i] - cx, dy = P.ys[i] - cy;
            const double d2 = fma(dx, dx, dy * dy);
            ab.seed(dx, dy, d2 * rsqrt_nr(d2));
        }
    }
    __syncthreads();
    const double reach[2] = {0.0, 0.0};
    residual_slice(P, cx, cy, R, H, 0, s, hist, rmax_rmin, argmax_argmin, wmax, wmin, wthr, edges_off == 0, false, reach,
                   BoundaryAlso{&ab, {}});
    ab.finish(P.xs, P.ys, P.n, sel + s * bins);
}

int launch_boundary_residual(const double *x, const double *y, const int64_t *offsets, int64_t S, const double *cxcyr,
                             int32_t bins, int64_t *sel, int32_t H, const double *edges, int64_t *hist, double *rmm,
                             int64_t *amm, hipStream_t st)
{
    if (S == 0) return PCFIT_OK;
    const size_t lds = hist_lds_bytes(H) + angle_bins_lds_bytes(bins);
    PCFIT_TRY(allow_lds(boundary_residual_kernel, lds));
    boundary_residual_kernel<<<dim3((unsigned)S), dim3(256), lds, st>>>(x, y, offsets, cxcyr, bins, sel, H, edges, hist,
                                                                     rmm, amm);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API int pcfit_kasa_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                   int64_t max_slice_len, double *cxcyr, int32_t *status, void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "kasa: n_slices out of range");
    PCFIT_REQUIRE(n_slices == 0 || (x && y && offsets && cxcyr), "kasa: null pointer");
    return launch_kasa(x, y, offsets, n_slices, max_slice_len, cxcyr, status, as_stream(stream));
}

PCFIT_API int pcfit_lm_batch_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                 int64_t max_slice_len, const double *init, double huber_delta, int32_t max_iter,
                                 double lm_lambda0, double tol_update, double tol_cost, double *out, int32_t *iters,
                                 void *stream)
{
    PCFIT_ENTER();
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
    PCFIT_ENTER();
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
    PCFIT_ENTER();
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

PCFIT_API int pcfit_boundary_residuals_batch_dev(const double *x, const double *y, const int64_t *offsets,
                                                 int64_t n_slices, const double *cxcyr, int32_t bins, int64_t *sel,
                                                 int32_t n_bins, const double *edges, int64_t *hist,
                                                 double *rmax_rmin, int64_t *argmax_argmin, void *stream)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_slices >= 0 && n_slices < (1ll << 31), "boundary_residuals: n_slices out of range");
    PCFIT_REQUIRE(bins >= 1 && bins <= ANGLE_FAST_BINS, "boundary_residuals: bins must be 1..4096");
    PCFIT_REQUIRE(n_bins >= 0 && n_bins <= 4096, "boundary_residuals: n_bins must be 0..4096");
    if (n_slices == 0)
        return PCFIT_OK;
    PCFIT_REQUIRE(x && y && offsets && cxcyr && sel && rmax_rmin && argmax_argmin, "boundary_residuals: null pointer");
    PCFIT_REQUIRE(n_bins == 0 || (edges && hist), "boundary_residuals: histogram asked for without edges/hist");
    return launch_boundary_residual(x, y, offsets, n_slices, cxcyr, bins, sel, n_bins, edges, hist, rmax_rmin,
                                    argmax_argmin, as_stream(stream));
}

#ifdef PCFIT_PHASE_CLOCK
// tools only: clocks accumulated per phase by thread 0 of every workgroup (16 counters)
PCFIT_API int pcfit_debug_phase_clocks(int64_t *out16, int reset)
{
    PCFIT_ENTER();
    PCFIT_HIP(hipDeviceSynchronize());
    PCFIT_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(pcfit::g_phase), 16 * sizeof(int64_t)));
    if (reset) {
        int64_t zero[16] = {0};
        PCFIT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pcfit::g_phase), zero, sizeof(zero)));
    }
    return PCFIT_OK;
}
#endif
