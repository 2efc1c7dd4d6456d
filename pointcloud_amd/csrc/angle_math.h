// boundary_by_angle_max (boundary.py:7-42) on one workgroup per slice: the pieces shared by the boundary kernels of
// residual.hip and the kernel of fit.hip that takes the boundary and the residual statistics in one sweep.
#pragma once

#include "pcfit_internal.h"

namespace pcfit {

// the exact bin of a point: numpy's floor(theta / width), clipped (boundary.py:21-25)
__device__ __forceinline__ int32_t exact_bin(double dx, double dy, double width, int32_t bins)
{
    const double fb = floor(wrap_angle(dy, dx) / width);
    return fb < 0.0 ? 0 : (fb > (double)(bins - 1) ? bins - 1 : (int32_t)fb);
}

// the two exact passes of the angular-bin boundary over the points of one slice: largest radius per bin (r >= 0,
// so its bit pattern orders like r; +1 keeps an all-zero radius apart from "empty"), then the first index that
// attains it
__device__ inline void angle_max_exact_passes(const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                                              double cx, double cy, double width, int32_t bins,
                                              unsigned long long *best, unsigned int *who)
{
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double dx = x[i] - cx, dy = y[i] - cy;
        const double r = cr_hypot(dx, dy);
        if (r == r && dx == dx && dy == dy)
            atomicMax(&best[exact_bin(dx, dy, width, bins)], (unsigned long long)__double_as_longlong(r) + 1ull);
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double dx = x[i] - cx, dy = y[i] - cy;
        const double r = cr_hypot(dx, dy);
        if (r == r && dx == dx && dy == dy) {
            const int32_t b = exact_bin(dx, dy, width, bins);
            if ((unsigned long long)__double_as_longlong(r) + 1ull == best[b])
                atomicMin(&who[b], (unsigned int)i);
        }
    }
    __syncthreads();
}

// The angle of (dx, dy) in [0, 2 pi) to within ANGLE_EPS of what wrap_angle gives: atan(a) = a P(a^2) on [0, 1]
// (weighted least-squares fit of degree 6 in a^2, reweighted towards equal ripple; largest error 2.5e-7 over two
// million samples - the points that land within that of a bin's edge, a handful per slice, are judged by the exact
// code anyway), the octant by selects.  Not a number when both offsets are zero.
constexpr double ANGLE_EPS = 6e-7;
// one Horner step as a three-operand multiply-add (left to itself the compiler copies each coefficient into the
// accumulator of a two-operand form: a 64-bit copy costs what the multiply-add does)
__device__ __forceinline__ double horner_step(double p, double t, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(t), "v"(c));
    return r;
}
__device__ __forceinline__ double quick_angle(double dx, double dy)
{
    double big, small;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(big) : "v"(dx), "v"(dy));       // (a NaN offset still ends in a NaN ratio)
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(small) : "v"(dx), "v"(dy));
    double inv = __builtin_amdgcn_rcp(big);
    inv = fma(fma(-big, inv, 1.0), inv, inv);
    const double a = small * inv, t = a * a;
    double p = 0.006811761117765259;
    p = horner_step(p, t, -0.03360411730997864);
    p = horner_step(p, t, 0.07962354525197074);
    p = horner_step(p, t, -0.13233334611559186);
    p = horner_step(p, t, 0.1980781343401504);
    p = horner_step(p, t, -0.33317367796696673);
    p = horner_step(p, t, 0.9999961114613874);
    double th = a * p;
    th = fabs(dy) > fabs(dx) ? 0.25 * TWO_PI - th : th;
    th = dx < 0.0 ? 0.5 * TWO_PI - th : th;
    th = dy < 0.0 ? TWO_PI - th : th;
    return th;
}

// boundary_by_angle_max for up to ANGLE_FAST_BINS bins in ONE sweep over the slice, without a hypot and an atan2 per
// point.  A point's radius is taken from the reciprocal square root (r1, within 4.7e-16) and its bin from quick_angle;
// a point whose quick angle comes within ANGLE_EPS of a bin's edge is "not clear" and goes on the candidate list
// whatever its radius.  Every clear point raises its bin's bound - the largest quick radius seen so far, kept as a
// float rounded down - by an LDS atomic that returns the bound it found, and goes on the list when its radius reaches
// what it found: a bound never exceeds the largest radius of the bin, so a point below it cannot be the bin's
// largest, and the point that is the largest is listed whenever it arrives.  A short first stretch of the slice only
// raises bounds (and is swept again with the rest), which keeps the list to a few entries per bin: its own largest
// and the record breakers after it.  The exact code (correctly rounded hypot, atan2, numpy's bin rule) then runs over
// the list twice: largest radius per bin, first index that attains it - the result of the exact passes over all
// points.  A list that overflows (points that arrive in ascending order of radius) sends the slice through those.
// Every wave appends to a list of its own, at a count it keeps itself: no atomic counter, nothing to wait for.
constexpr int ANGLE_FAST_BINS = 4096;
constexpr int ANGLE_SEED_POINTS = 4096;
// every wave of the workgroup keeps a candidate list of its own (no atomic counter, no wait for one): entries per wave
// (a slice in random order leaves about bins x (1 + ln(points per bin / seeds per bin)) / waves entries in each)
__host__ __device__ constexpr unsigned int angle_list_cap(int32_t bins)
{
    return (bins <= 1024 ? 2u : 1u) * (unsigned int)bins + 64u;
}
constexpr int ANGLE_MAX_WAVES = 4;
// LDS of an AngleBins, in bytes: bins x (uint64 best, uint32 who) + (bins + 1) x uint32 bound + the lists + their counts
__host__ __device__ constexpr size_t angle_bins_lds_bytes(int32_t bins)
{
    return (size_t)bins * 12 + (((size_t)bins + 1 + 1) & ~(size_t)1) * 4 +
           (size_t)ANGLE_MAX_WAVES * angle_list_cap(bins) * 4 + ANGLE_MAX_WAVES * 4;
}

struct AngleBins {
    unsigned long long *best;
    unsigned int *who, *bound, *list, *n_list;
    unsigned int mine_at, count;            // this wave's list, and how many entries it has (uniform over the wave)
    int32_t bins;
    double cx, cy, width, inv_width, margin;

    // mem: angle_bins_lds_bytes(bins) bytes of LDS, 8-byte aligned; workgroups of at most ANGLE_MAX_WAVES waves.
    // Ends without a barrier: the caller has one before the first point.
    __device__ __forceinline__ void init(void *mem, int32_t nbins, double centre_x, double centre_y)
    {
        bins = nbins;
        cx = centre_x;
        cy = centre_y;
        best = reinterpret_cast<unsigned long long *>(mem);
        who = reinterpret_cast<unsigned int *>(best + bins);
        bound = who + bins;
        list = bound + ((bins + 2) & ~1);
        n_list = list + ANGLE_MAX_WAVES * angle_list_cap(bins);
        mine_at = (threadIdx.x >> 6) * angle_list_cap(bins);
        count = 0u;
        width = TWO_PI / (double)bins;
        inv_width = (double)bins / TWO_PI;
        margin = ANGLE_EPS * inv_width + 1e-9;                   // in bins
        for (int b = threadIdx.x; b < bins; b += blockDim.x) {
            best[b] = 0ull;
            who[b] = 0xffffffffu;
            bound[b] = 0u;
        }
        if (threadIdx.x == 0)
            bound[bins] = 0u;                                   // the slot of the points that are not clear
    }

    // bin of a point with offsets (dx, dy) and quick radius r1 when it is certain, else `bins`.
    // (A NaN anywhere - a point on the centre, an overflowing distance - is not clear.)
    __device__ __forceinline__ int32_t clear_bin(double dx, double dy, double r1) const
    {
        const double tb = quick_angle(dx, dy) * inv_width;
        const double fb = floor(tb);
        const bool clear = fabs((tb - fb) - 0.5) < 0.5 - margin && (uint32_t)(int32_t)fb < (uint32_t)bins &&
                           r1 <= 1.7976931348623157e308;
        return clear ? (int32_t)fb : bins;
    }

    // What a bin's bound holds for a point of quick radius r1: the bit pattern of r1 as a float rounded to
    // nearest, which orders like the radius; the float just below it (pattern - 1) is below r1, so it is a lower
    // bound of the radius of the point that set it.  0 = nobody yet.
    __device__ __forceinline__ static unsigned int bound_bits(double r1) { return __float_as_uint((float)r1); }

    // a point of the first stretch: raises its bin's bound, nothing else
    __device__ __forceinline__ void seed(double dx, double dy, double r1)
    {
        const int32_t b = clear_bin(dx, dy, r1);
        atomicMax(&bound[b], b < bins ? bound_bits(r1) : 0u);
    }

    // Every point of the slice (those of the first stretch again) in two steps, so that a caller with several points
    // in hand can have all their atomics in flight before it looks at what the first one found:
    // probe() raises the bin's bound and notes what it found there; settle() lists the point when it reaches that.
    // `i`: the point's index in the slice; invalid: a slot past the slice's end.
    struct Probe {
        unsigned int found;
        bool clear;
        double r1;
    };
    __device__ __forceinline__ Probe probe(double dx, double dy, double r1, bool valid)
    {
        const int32_t b = clear_bin(dx, dy, r1);
        const bool clear = b < bins;
        // (no branch around the atomic: a point that is not clear, or not there, raises nothing in the spare slot)
        return Probe{atomicMax(&bound[b], clear && valid ? bound_bits(r1) : 0u), clear, r1};
    }
    __device__ __forceinline__ void settle(const Probe &pr, uint32_t i, bool valid)
    {
        // (the float below the one found; for an empty bin the pattern of a NaN, below which nothing lies)
        const double thr = (double)__uint_as_float(pr.found - 1u) * (1.0 - 2e-15);
        const bool cand = valid && (!pr.clear || !(pr.r1 < thr));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
        if (m != 0ull) {                                        // (uniform over the wave)
            const unsigned int at = count + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            if (cand && at < angle_list_cap(bins))
                list[mine_at + at] = i;
            count += (unsigned int)__popcll(m);
        }
    }
    __device__ __forceinline__ void point(double dx, double dy, double r1, uint32_t i, bool valid)
    {
        settle(probe(dx, dy, r1, valid), i, valid);
    }

    // behind the last point: the waves' counts out, a barrier, the exact code over the lists, then the indices
    __device__ __forceinline__ void finish(const double *__restrict__ xs, const double *__restrict__ ys, int64_t n,
                                           int64_t *__restrict__ sel_row)
    {
        const int nw = (int)(blockDim.x >> 6);
        if ((threadIdx.x & 63) == 0)
            n_list[threadIdx.x >> 6] = count;
        __syncthreads();
        bool over = false;
        for (int w = 0; w < nw; w++)
            over = over || n_list[w] > angle_list_cap(bins);
        if (over) {
            angle_max_exact_passes(xs, ys, n, cx, cy, width, bins, best, who);
        } else {
            for (int pass = 0; pass < 2; pass++) {
                for (int w = 0; w < nw; w++) {
                    const unsigned int *lw = list + w * angle_list_cap(bins);
                    const unsigned int listed = n_list[w];
                    for (unsigned int k = threadIdx.x; k < listed; k += blockDim.x) {
                        const unsigned int i = lw[k];
                        const double dx = xs[i] - cx, dy = ys[i] - cy;
                        const double r = cr_hypot(dx, dy);
                        if (r == r && dx == dx && dy == dy) {
                            const int32_t b = exact_bin(dx, dy, width, bins);
                            const unsigned long long key = (unsigned long long)__double_as_longlong(r) + 1ull;
                            if (pass == 0)
                                atomicMax(&best[b], key);
                            else if (key == best[b])
                                atomicMin(&who[b], i);
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (int b = threadIdx.x; b < bins; b += blockDim.x)
            sel_row[b] = who[b] == 0xffffffffu ? -1 : (int64_t)who[b];
    }
};

}  // namespace pcfit
