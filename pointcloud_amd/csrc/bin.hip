// Slice binning: scatter the (x, y) of every point into slice-contiguous arrays,
// keeping the original order inside a slice (analyzer.py:49-52 takes
// points[mask] per window).  A stable counting sort in three steps, gfx950:
//
//   count    one wave per contiguous chunk of the input builds the chunk's
//            per-window histogram in LDS and stores it as a row of a
//            chunks x windows matrix;
//   scan     column-wise exclusive prefix over the chunk rows (segmented in
//            three small kernels) plus the prefix over windows = CSR offsets;
//   scatter  the same wave walks its chunk again, 64 points a step.  Lanes
//            holding the same window find each other with ballots over the key
//            bits, so a lane's rank among its peers and the peer count are bit
//            counts; one LDS cursor per window (private to the wave, seeded from
//            the scan) gives the output position.  No atomics, no inter-wave
//            ordering: stable and deterministic by construction.  For up to
//            256 windows the points go through LDS first (see
//            bin_scatter_staged_kernel) so that the output is written in runs;
//            more windows than that are binned in two levels when the scan is
//            large (bin_two_level for the host-level calls, pcfit_bin2_*_dev
//            for callers on the device): first into groups of 64 or 128
//            windows with a 16-bit key per entry, then every group into its
//            windows by the keys, both through the staged kernel.
//
// Windows are closed intervals that may share an edge, so a point can belong
// to two consecutive windows (both get a copy) or to none.  Membership is
// decided by the caller's z_lo / z_hi arrays, compared exactly; the uniform
// spacing is used only to find the candidate window and to skip the array
// look-ups for points that are clearly interior.
// HBM traffic: 24 B/point read in count, 24 B read + 16 B (+8 B perm) written
// in scatter; in two levels 24 B (count) + 24 B read + 18 B written (level 1)
// + 2 B (count) + 18 B read + 16 B written (level 2), +16 B with perm.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "pcfit_internal.h"

namespace pcfit {

constexpr int BIN_SEG = 64;            // chunk rows per scan segment
constexpr int BIN_UNROLL = 4;          // 64-point trips whose loads are issued together
constexpr int64_t BIN_MAX_SLICES = 16384;
constexpr int BIN_GROUP_MAX = 128;     // windows per group of the two-level form, at most
constexpr double BIN_EDGE_EPS = 1e-6;  // in units of the window step
// Points from which more than BIN_STG_MAX_SLICES windows are binned in two levels.  Measured on resident scans in
// random z order, one analysis with one level / with two: 100 M points / 2000 windows 10.7 / 7.3 ms, 20 M / 1000
// windows 2.25 / 1.94 ms, 10 M / 400 windows 1.58 / 1.48 ms, 5 M / 400 windows 1.31 / 1.33 ms, 5 M / 2000 windows
// 3.84 / 3.99 ms (the second count pass and the extra launches cost more than they save).
constexpr int64_t BIN_TWO_LEVEL_MIN = 8'000'000;

struct BinGeom {
    double lo0, inv_step, rel_width;   // rel_width = (z_hi[0]-z_lo[0]) / step
    int64_t S;
};

// where the points are read from: coordinate i of point k at x[k * stride] etc. (a scan of xyz rows: x = xyz,
// y = xyz + 1, z = xyz + 2, stride 3; three separate arrays: stride 1)
struct BinSrc {
    const double *x, *y, *z;
    int64_t stride;
    const double *zcol;            // z of point k at zcol[k] as well (nullable): the count pass reads this
};
// (x, y) of a row of a scan: 16 bytes at an address that is a multiple of 8, taken with one load
struct __attribute__((packed, aligned(8))) BinRowXY {
    double x, y;
};
__device__ __forceinline__ bool is_rows(const BinSrc &src) { return src.stride == 3 && src.y == src.x + 1 && src.z == src.x + 2; }
static BinSrc rows_of(const double *xyz, const double *zcol = nullptr) { return BinSrc{xyz, xyz + 1, xyz + 2, 3, zcol}; }

// Workgroups are handed to the 8 XCDs in turn (workgroup b runs on XCD b % 8) and every XCD has its own L2.
// Consecutive chunks write neighbouring parts of every window's output, often parts of the same cache line; with
// this mapping an XCD works through one contiguous range of chunks, so those parts meet in one L2.
constexpr int N_XCD = 8;
__device__ __forceinline__ int64_t chunk_of_block(int64_t b, int64_t chunks)
{
    const int64_t xcd = b % N_XCD, idx = b / N_XCD;
    const int64_t q = chunks / N_XCD, r = chunks % N_XCD;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

// windows containing z: k1 (or -1) and, if a second one, k1 + 1 (has2)
__device__ __forceinline__ void windows_of(double z, const BinGeom &g, const double *__restrict__ z_lo,
                                           const double *__restrict__ z_hi, int32_t &k1, bool &has2)
{
    k1 = -1;
    has2 = false;
    if (!(z == z))
        return;
    const double t = (z - g.lo0) * g.inv_step;
    if (!(t > -2.0) || !(t < (double)g.S + 2.0))
        return;
    const double ft = floor(t);
    const int64_t k0 = (int64_t)ft;
    const double frac = t - ft;
    if (frac > BIN_EDGE_EPS && frac < g.rel_width - BIN_EDGE_EPS) {
        if (k0 >= 0 && k0 < g.S)
            k1 = (int32_t)k0;
        return;
    }
    if (frac > g.rel_width + BIN_EDGE_EPS && frac < 1.0 - BIN_EDGE_EPS)
        return;                                             // in the gap between two windows
    // near an edge: the arrays decide, exactly as z >= lo && z <= hi
#pragma unroll
    for (int d = -1; d <= 1; d++) {
        const int64_t k = k0 + d;
        if (k < 0 || k >= g.S)
            continue;
        if (z >= z_lo[k] && z <= z_hi[k]) {
            if (k1 < 0)
                k1 = (int32_t)k;
            else
                has2 = true;                                // windows are consecutive: this is k1 + 1
        }
    }
}

// Two levels (bin_two_level): level 1 sends every member of a window to the group of 1 << wshift consecutive
// windows that holds the window, with a 16-bit key - the window's number within the group, bit 15 when the point
// is in the next window of the same group as well.  A point in the last window of a group and in the first of the
// next becomes one entry in each.  Level 2 reads the keys in place of z.
struct BinGroups {
    int32_t G, wshift;              // number of groups, log2 of the windows per group
    const uint16_t *key_in;         // level 2
    uint16_t *key_out;              // level 1
    const int64_t *goff, *cbase;    // level 2: first entry and first chunk of every group (G + 1 each)
    int64_t cap1;                   // level 1: entries the level-1 arrays hold (more found: nothing is written)
};

// group (and key) of a member of window k1 / of windows k1 and k1 + 1; `two`: an entry in group b + 1 as well
__device__ __forceinline__ void group_of(int32_t k1, bool has2, int32_t wshift, int32_t &b, bool &two, uint32_t &key16)
{
    b = k1 >> wshift;
    two = has2 && ((k1 + 1) >> wshift) != b;
    key16 = (uint32_t)(k1 - (b << wshift)) | ((has2 && !two) ? 0x8000u : 0u);
}

// GROUPS: count per group instead of per window (gr.G groups; counts has gr.G columns)
template <bool GROUPS>
__global__ __launch_bounds__(64) void bin_count_kernel(BinSrc src, int64_t n, int64_t chunk_len,
                                                      const double *__restrict__ z_lo,
                                                      const double *__restrict__ z_hi, BinGeom g, BinGroups gr,
                                                      uint32_t *__restrict__ counts)
{
    extern __shared__ uint32_t lhist[];
    const int64_t c = blockIdx.x;
    const int64_t cols = GROUPS ? (int64_t)gr.G : g.S;
    for (int64_t s = threadIdx.x; s < cols; s += 64)
        lhist[s] = 0;
    __syncthreads();
    const int64_t begin = c * chunk_len, end = min(n, begin + chunk_len);
    // (the z column by itself when the scan has one: 8 bytes a point instead of the 24 of a row)
    const double *zsrc = src.zcol ? src.zcol : src.z;
    const int64_t zstride = src.zcol ? 1 : src.stride;
    // BIN_UNROLL loads of z in flight per lane: one wave per chunk has nothing else to hide the latency with
    for (int64_t base = begin + threadIdx.x; base < end; base += 64 * BIN_UNROLL) {
        double z[BIN_UNROLL];
#pragma unroll
        for (int u = 0; u < BIN_UNROLL; u++) {
            const int64_t i = base + 64 * u;
            z[u] = i < end ? zsrc[i * zstride] : __longlong_as_double(0x7ff8000000000000ll);   // NaN: no window
        }
#pragma unroll
        for (int u = 0; u < BIN_UNROLL; u++) {
            int32_t k1;
            bool has2;
            windows_of(z[u], g, z_lo, z_hi, k1, has2);
            if (k1 >= 0) {
                if constexpr (GROUPS) {
                    int32_t b;
                    uint32_t key16;
                    group_of(k1, has2, gr.wshift, b, has2, key16);
                    k1 = b;
                }
                atomicAdd(&lhist[k1], 1u);
                if (has2)
                    atomicAdd(&lhist[k1 + 1], 1u);
            }
        }
    }
    __syncthreads();
    for (int64_t s = threadIdx.x; s < cols; s += 64)
        counts[c * cols + s] = lhist[s];
}

// the group of level-2 chunk c: the last one whose first chunk is not after c
__device__ __forceinline__ int32_t group_of_chunk(const int64_t *__restrict__ cbase, int32_t G, int64_t c)
{
    int32_t lo = 0, hi = G - 1;
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (cbase[mid] <= c)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// level 2, count: one wave per chunk of a group's entries; counts[c][k] for the 1 << wshift windows of the group
__global__ __launch_bounds__(64) void bin2_count_kernel(BinGroups gr, int64_t chunk_len, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t lhist[BIN_GROUP_MAX + 1];
    const int W = 1 << gr.wshift;
    const int64_t c = blockIdx.x;
    if (c >= gr.cbase[gr.G])
        return;                     // (a launch sized for the most chunks there can be)
    for (int s = threadIdx.x; s <= W; s += 64)
        lhist[s] = 0;
    __syncthreads();
    const int32_t grp = group_of_chunk(gr.cbase, gr.G, c);
    const int64_t begin = gr.goff[grp] + (c - gr.cbase[grp]) * chunk_len, end = min(gr.goff[grp + 1], begin + chunk_len);
    constexpr int UN = 8;
    for (int64_t base = begin + threadIdx.x; base < end; base += 64 * UN) {
        uint32_t k[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int64_t i = base + 64 * u;
            k[u] = i < end ? (uint32_t)gr.key_in[i] : 0xffffu;
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            if (k[u] == 0xffffu)
                continue;
            const uint32_t k1 = k[u] & 0x7fffu;
            atomicAdd(&lhist[k1], 1u);
            if (k[u] >> 15)
                atomicAdd(&lhist[k1 + 1], 1u);
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < W; s += 64)
        counts[c * W + s] = lhist[s];
}

// level 2, scan: one wave per window; down the chunk rows of the window's group, counts[c][k] <- members in the
// group's chunks before c, totals[window] = all of them
__global__ __launch_bounds__(64) void bin2_colscan_kernel(BinGroups gr, int64_t S, uint32_t *__restrict__ counts,
                                                         int64_t *__restrict__ totals)
{
    const int64_t s = blockIdx.x;
    const int W = 1 << gr.wshift, lane = threadIdx.x;
    const int32_t grp = (int32_t)(s >> gr.wshift), k = (int32_t)(s & (W - 1));
    const int64_t c0 = gr.cbase[grp], c1 = gr.cbase[grp + 1];
    int64_t run = 0;
    for (int64_t base = c0; base < c1; base += 64) {
        const int64_t c = base + lane;
        const uint32_t v = c < c1 ? counts[c * W + k] : 0u;
        int64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(incl, d);
            if (lane >= d)
                incl += t;
        }
        if (c < c1)
            counts[c * W + k] = (uint32_t)(run + incl - v);
        run += __shfl(incl, 63);
    }
    if (lane == 0 && s < S)
        totals[s] = run;
}

// level 2, chunks: first chunk of every group from the groups' first entries; status 1 (and no chunks at all) when
// the entries exceed what the level-1 arrays hold, 2 when a group has 2^32 entries or more
__global__ void bin2_plan_kernel(const int64_t *__restrict__ goff, int32_t G, int64_t cap1, int64_t chunk_len,
                                 int64_t *__restrict__ cbase, int32_t *__restrict__ status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    int32_t st = goff[G] > cap1 ? 1 : 0;
    for (int32_t k = 0; k < G && st == 0; k++)
        if (goff[k + 1] - goff[k] >= (1ll << 32))
            st = 2;
    int64_t c = 0;
    for (int32_t k = 0; k < G; k++) {
        cbase[k] = c;
        if (st == 0)
            c += (goff[k + 1] - goff[k] + chunk_len - 1) / chunk_len;
    }
    cbase[G] = c;
    *status = st;
}

// segment sums: segsum[g][s] = sum of counts rows of segment g
__global__ void bin_segsum_kernel(const uint32_t *__restrict__ counts, int64_t chunks, int64_t S,
                                  int64_t *__restrict__ segsum)
{
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t seg = blockIdx.y;
    if (s >= S)
        return;
    const int64_t c0 = seg * BIN_SEG, c1 = min(chunks, c0 + BIN_SEG);
    int64_t acc = 0;
    for (int64_t c = c0; c < c1; c++)
        acc += counts[c * S + s];
    segsum[seg * S + s] = acc;
}

// per window: exclusive prefix of the segment sums (in place) and the total; one wave per window, 64 segments a step
__global__ __launch_bounds__(64) void bin_segscan_kernel(int64_t *__restrict__ segsum, int64_t segs, int64_t S,
                                                        int64_t *__restrict__ totals)
{
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;
    int64_t run = 0;
    for (int64_t base = 0; base < segs; base += 64) {
        const int64_t sg = base + lane;
        const int64_t v = sg < segs ? segsum[sg * S + s] : 0;
        int64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(incl, d);
            if (lane >= d)
                incl += t;
        }
        if (sg < segs)
            segsum[sg * S + s] = run + incl - v;
        run += __shfl(incl, 63);
    }
    if (lane == 0)
        totals[s] = run;
}

// offsets[0..S] = exclusive prefix of totals; one workgroup, serial over tiles of 1024
__global__ __launch_bounds__(1024) void bin_offsets_kernel(const int64_t *__restrict__ totals, int64_t S,
                                                          int64_t *__restrict__ offsets)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < S; base += 1024) {
        const int64_t s = base + threadIdx.x;
        const int64_t v = s < S ? totals[s] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int64_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const int64_t incl = part[threadIdx.x];
        if (s < S)
            offsets[s] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023)
            carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        offsets[S] = carry;
}

// counts[c][s] <- number of members of window s in the chunks of c's segment before c
__global__ void bin_rowscan_kernel(uint32_t *__restrict__ counts, int64_t chunks, int64_t S)
{
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t seg = blockIdx.y;
    if (s >= S)
        return;
    const int64_t c0 = seg * BIN_SEG, c1 = min(chunks, c0 + BIN_SEG);
    uint32_t run = 0;
    for (int64_t c = c0; c < c1; c++) {
        const uint32_t t = counts[c * S + s];
        counts[c * S + s] = run;
        run += t;
    }
}

// lanes whose key equals this lane's query: ballots over the key bits
// (NBITS is a compile-time constant so that the ballots stay in scalar registers and the loops unroll)
template <int NBITS>
__device__ __forceinline__ uint64_t match_lanes(const uint64_t (&bit_ballot)[NBITS], uint32_t query)
{
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < NBITS; b++)
        m &= ((query >> b) & 1u) ? bit_ballot[b] : ~bit_ballot[b];
    return m;
}

template <int NBITS>
__device__ __forceinline__ void key_ballots(uint32_t key, uint64_t (&bit_ballot)[NBITS])
{
#pragma unroll
    for (int b = 0; b < NBITS; b++)
        bit_ballot[b] = __ballot((key >> b) & 1u);
}

// The x, y, z of BIN_UNROLL trips of 64 points, all loads issued before any is used.  Lanes past the end
// read the chunk's last point (a load without a branch) and get NaN for z: no window.
__device__ __forceinline__ void load_trips(const BinSrc &src, int64_t sbase, int64_t end, int lane,
                                           double (&lx)[BIN_UNROLL], double (&ly)[BIN_UNROLL],
                                           double (&lz)[BIN_UNROLL])
{
#pragma unroll
    for (int u = 0; u < BIN_UNROLL; u++) {
        const int64_t i = sbase + 64 * u + lane;
        const int64_t ic = i < end ? i : end - 1;
        lx[u] = src.x[ic * src.stride];
        ly[u] = src.y[ic * src.stride];
        const double z = src.z[ic * src.stride];
        lz[u] = i < end ? z : __longlong_as_double(0x7ff8000000000000ll);
    }
}

template <int NBITS, bool PERM>
__global__ __launch_bounds__(64) void bin_scatter_kernel(BinSrc src, int64_t n, int64_t chunk_len,
                                                        const double *__restrict__ z_lo,
                                                        const double *__restrict__ z_hi, BinGeom g,
                                                        const uint32_t *__restrict__ counts,
                                                        const int64_t *__restrict__ segbase,
                                                        const int64_t *__restrict__ offsets,
                                                        double *__restrict__ x_out, double *__restrict__ y_out,
                                                        int64_t *__restrict__ perm)
{
    // cursor[s]: where this wave's next member of window s goes, relative to the window's first slot
    extern __shared__ uint32_t cursor[];
    const int64_t c = chunk_of_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int64_t s = lane; s < g.S; s += 64)
        cursor[s] = (uint32_t)(segbase[(c / BIN_SEG) * g.S + s] + (int64_t)counts[c * g.S + s]);
    __syncthreads();
    const int64_t begin = c * chunk_len, end = min(n, begin + chunk_len);
    for (int64_t sbase = begin; sbase < end; sbase += 64 * BIN_UNROLL) {
        // the loads of BIN_UNROLL trips are issued together; the trips are then placed in order
        double lx[BIN_UNROLL], ly[BIN_UNROLL], lz[BIN_UNROLL];
        load_trips(src, sbase, end, lane, lx, ly, lz);
#pragma unroll
        for (int u = 0; u < BIN_UNROLL; u++) {
            if (sbase + 64 * u >= end)
                break;
            const int64_t i = sbase + 64 * u + lane;
            int32_t k1;
            bool has2;
            const double px = lx[u], py = ly[u];
            windows_of(lz[u], g, z_lo, z_hi, k1, has2);
            // key 0 = no window, key k+1 = window k
            const uint32_t key = (uint32_t)(k1 + 1);
            uint64_t bit_ballot[NBITS];
            key_ballots<NBITS>(key, bit_ballot);
            const uint64_t twice = __ballot(has2);
            uint64_t same = match_lanes<NBITS>(bit_ballot, key);
            if (twice == 0ull) {
                if (k1 >= 0) {
                    const int rank = __popcll(same & below), cnt = __popcll(same);
                    const uint32_t rel = cursor[k1];
                    if (rank == cnt - 1)
                        cursor[k1] = rel + cnt;
                    const int64_t at = offsets[k1] + rel;
                    x_out[at + rank] = px;
                    y_out[at + rank] = py;
                    if (PERM) perm[at + rank] = i;
                }
            } else {
                // window k receives lanes with key1 == k and lanes with key1 == k-1 that lie in two windows
                const uint64_t from_prev = match_lanes<NBITS>(bit_ballot, key - 1u) & twice;   // key-1 = window k1-1
                const uint64_t to_next = match_lanes<NBITS>(bit_ballot, key + 1u);            // lanes in window k1+1
                int64_t at1 = 0, at2 = 0;
                int rank1 = 0, cnt1 = 0, rank2 = 0, cnt2 = 0;
                if (k1 >= 0) {
                    const uint64_t grp1 = same | (key >= 2u ? from_prev : 0ull);
                    rank1 = __popcll(grp1 & below);
                    cnt1 = __popcll(grp1);
                    at1 = cursor[k1];               // (relative; the window's first slot is added below)
                    if (has2) {
                        const uint64_t grp2 = to_next | (same & twice);
                        rank2 = __popcll(grp2 & below);
                        cnt2 = __popcll(grp2);
                        at2 = cursor[k1 + 1];
                    }
                }
                // all reads of the cursors precede the writes (one wave, LDS ops in order)
                if (k1 >= 0) {
                    if (rank1 == cnt1 - 1)
                        cursor[k1] = (uint32_t)(at1 + cnt1);
                    if (has2 && rank2 == cnt2 - 1)
                        cursor[k1 + 1] = (uint32_t)(at2 + cnt2);
                    at1 += offsets[k1];
                    if (has2)
                        at2 += offsets[k1 + 1];
                    x_out[at1 + rank1] = px;
                    y_out[at1 + rank1] = py;
                    if (PERM) perm[at1 + rank1] = i;
                    if (has2) {
                        x_out[at2 + rank2] = px;
                        y_out[at2 + rank2] = py;
                        if (PERM) perm[at2 + rank2] = i;
                    }
                }
            }
        }
    }
}

// Scatter through LDS for up to BIN_STG_MAX_SLICES windows.  The direct scatter above stores 8 bytes at a
// time at up to 64 unrelated addresses per trip; when consecutive points fall in unrelated windows the
// memory system sees partial lines (measured: 2.9 bytes written to HBM per byte of output).  Here a
// workgroup of four waves takes its chunk in tiles of TILE points, orders each tile by window inside LDS
// (same order as the input within a window) and copies the ordered tile out, so that neighbouring lanes
// store neighbouring addresses: a tile's members of one window are one contiguous run of the output.
//
//   pass 1   wave w walks its quarter of the tile, 64 points a trip: window of every point, its rank
//            among the wave's members of that window (ballots, as above), count per (wave, window);
//   scan     per window: members in the tile, first LDS slot of the window's run, first slot of every
//            wave's share of the run, and where the run goes in the output;
//   pass 2   every point into its slot; pass 3: the slots, in order, to the output.
//
// A tile whose members (points on a shared edge count twice) exceed the staging area is stored directly
// from pass 2 with the same positions.
constexpr int BIN_STG_MAX_SLICES = 256;      // window index fits the byte kept per staged point

template <int TILE> constexpr int bin_stage_cap() { return TILE + TILE / 16; }

template <int TILE, int WAVES, bool PERM, int LEVEL = 0> static size_t bin_stage_lds(int64_t S)
{
    const size_t cap = (size_t)bin_stage_cap<TILE>();
    return cap * 16 + (size_t)S * 16 + ((size_t)WAVES * S + (size_t)S + 1 + WAVES) * 4 +
           (PERM ? cap * 4 : 0) + (LEVEL == 1 ? cap * 2 : 0) + cap;
}

// LEVEL 0: the windows themselves are the buckets.  LEVEL 1 / 2: the two levels of bin_two_level - at level 1 the
// buckets are the gr.G groups (windows found as at level 0, over all windows; every entry's key goes to gr.key_out),
// at level 2 the 1 << gr.wshift windows of the chunk's group (keys from gr.key_in, x and y from src, a group's
// entries in chunks of chunk_len from gr.goff[group] on; counts has one row per chunk, scanned by
// bin2_colscan_kernel; g.S = all windows).
template <int NBITS, bool PERM, int TILE, int WAVES, int LEVEL = 0>
__global__ __launch_bounds__(64 * WAVES) void bin_scatter_staged_kernel(
    BinSrc src, int64_t n, int64_t chunk_len, const double *__restrict__ z_lo,
    const double *__restrict__ z_hi, BinGeom g, const uint32_t *__restrict__ counts,
    const int64_t *__restrict__ segbase, const int64_t *__restrict__ offsets, double *__restrict__ x_out,
    double *__restrict__ y_out, int64_t *__restrict__ perm, double *__restrict__ z_out,
    const int64_t *__restrict__ idx_in, BinGroups gr)
{
    constexpr int TRIPS = TILE / (64 * WAVES);
    constexpr int CAP = bin_stage_cap<TILE>();
    extern __shared__ __align__(16) unsigned char stg_lds[];
    const int S = LEVEL == 0 ? (int)g.S : LEVEL == 1 ? gr.G : (1 << gr.wshift);
    double2 *stage = reinterpret_cast<double2 *>(stg_lds);
    int64_t *gpos = reinterpret_cast<int64_t *>(stg_lds + (size_t)CAP * 16);   // next output slot per window
    int64_t *gdelta = gpos + S;                                                  // output slot - LDS slot
    uint32_t *cnt = reinterpret_cast<uint32_t *>(gdelta + S);                    // [wave][window]
    uint32_t *lstart = cnt + WAVES * S;                                  // S + 1
    uint32_t *wave_sum = lstart + S + 1;                                         // WAVES
    uint32_t *sidx = wave_sum + WAVES;                                   // CAP, with PERM
    uint16_t *sk16 = reinterpret_cast<uint16_t *>(sidx + (PERM ? CAP : 0));      // CAP, at level 1
    uint8_t *skey = reinterpret_cast<uint8_t *>(sk16 + (LEVEL == 1 ? CAP : 0));  // CAP

    // (level 2: a launch may be sized for the most chunks there can be; the chunks there are share the XCDs evenly)
    const int64_t n_chunks = LEVEL == 2 ? gr.cbase[gr.G] : (int64_t)gridDim.x;
    if (LEVEL == 2 && (int64_t)blockIdx.x >= n_chunks)
        return;
    const int64_t c = chunk_of_block(blockIdx.x, n_chunks);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t *cw = cnt + w * S;
    int64_t begin, end;
    if constexpr (LEVEL == 1) {
        if (offsets[S] > gr.cap1)
            return;                 // more entries than the level-1 arrays hold: the plan kernel has reported it
    }
    if constexpr (LEVEL == 2) {
        const int32_t grp = group_of_chunk(gr.cbase, gr.G, c);
        const int64_t s0 = (int64_t)grp << gr.wshift;
        for (int s = tid; s < S; s += 64 * WAVES)
            gpos[s] = s0 + s < g.S ? offsets[s0 + s] + (int64_t)counts[c * S + s] : 0;
        begin = gr.goff[grp] + (c - gr.cbase[grp]) * chunk_len;
        end = min(gr.goff[grp + 1], begin + chunk_len);
    } else {
        for (int s = tid; s < S; s += 64 * WAVES)
            gpos[s] = offsets[s] + segbase[(c / BIN_SEG) * S + s] + (int64_t)counts[c * (int64_t)S + s];
        begin = c * chunk_len;
        end = min(n, begin + chunk_len);
    }
    for (int64_t tbase = begin; tbase < end; tbase += TILE) {
        for (int s = tid; s < WAVES * S; s += 64 * WAVES)
            cnt[s] = 0;
        __syncthreads();
        // pass 1
        double px[TRIPS], py[TRIPS];
        uint32_t key[TRIPS], r1[TRIPS], r2[TRIPS];      // key: bucket + 1, bit 31 = also in the next bucket
        uint32_t key16[LEVEL == 1 ? TRIPS : 1];         // level 1: what level 2 will read
        const int64_t wbase = tbase + (int64_t)w * (TRIPS * 64);
        if constexpr (LEVEL == 2) {
            uint32_t kk[TRIPS];
#pragma unroll
            for (int u = 0; u < TRIPS; u++) {
                const int64_t i = wbase + 64 * u + lane;
                const int64_t ic = i < end ? i : end - 1;
                px[u] = src.x[ic];
                py[u] = src.y[ic];
                kk[u] = gr.key_in[ic];
            }
#pragma unroll
            for (int u = 0; u < TRIPS; u++) {
                const int64_t i = wbase + 64 * u + lane;
                key[u] = i < end ? ((kk[u] & 0x7fffu) + 1u) | ((kk[u] >> 15) ? 0x80000000u : 0u) : 0u;
            }
        } else {
            // every load of the tile is issued before the first is used; lanes past the end read the
            // chunk's last point and get NaN for z: no window
            double pz[TRIPS];
            if (is_rows(src)) {             // rows of a scan: 16 + 8 bytes a lane
#pragma unroll
                for (int u = 0; u < TRIPS; u++) {
                    const int64_t i = wbase + 64 * u + lane;
                    const int64_t ic = i < end ? i : end - 1;
                    const BinRowXY xy = *reinterpret_cast<const BinRowXY *>(src.x + ic * 3);
                    px[u] = xy.x;
                    py[u] = xy.y;
                    pz[u] = src.x[ic * 3 + 2];
                }
            } else {
#pragma unroll
                for (int u = 0; u < TRIPS; u++) {
                    const int64_t i = wbase + 64 * u + lane;
                    const int64_t ic = i < end ? i : end - 1;
                    px[u] = src.x[ic * src.stride];
                    py[u] = src.y[ic * src.stride];
                    pz[u] = src.z[ic * src.stride];
                }
            }
#pragma unroll
            for (int u = 0; u < TRIPS; u++) {
                const int64_t i = wbase + 64 * u + lane;
                int32_t k1;
                bool has2;
                windows_of(i < end ? pz[u] : __longlong_as_double(0x7ff8000000000000ll), g, z_lo, z_hi, k1, has2);
                if constexpr (LEVEL == 1) {
                    key16[u] = 0u;
                    if (k1 >= 0) {
                        int32_t b;
                        group_of(k1, has2, gr.wshift, b, has2, key16[u]);
                        k1 = b;
                    }
                }
                key[u] = (uint32_t)(k1 + 1) | (has2 ? 0x80000000u : 0u);
            }
        }
#pragma unroll
        for (int u = 0; u < TRIPS; u++) {
            const bool has2 = (key[u] >> 31) != 0u;
            const int32_t k1 = (int32_t)(key[u] & 0x7fffffffu) - 1;
            const uint32_t k = (uint32_t)(k1 + 1);
            uint64_t bit_ballot[NBITS];
            key_ballots<NBITS>(k, bit_ballot);
            const uint64_t twice = __ballot(has2);
            const uint64_t same = match_lanes<NBITS>(bit_ballot, k);
            uint32_t a1 = 0, a2 = 0;
            if (twice == 0ull) {
                if (k1 >= 0) {
                    const int rank = __popcll(same & below), m = __popcll(same);
                    const uint32_t rel = cw[k1];
                    if (rank == m - 1)
                        cw[k1] = rel + m;
                    a1 = rel + rank;
                }
            } else {
                const uint64_t from_prev = match_lanes<NBITS>(bit_ballot, k - 1u) & twice;
                const uint64_t to_next = match_lanes<NBITS>(bit_ballot, k + 1u);
                int rank1 = 0, m1 = 0, rank2 = 0, m2 = 0;
                uint32_t rel1 = 0, rel2 = 0;
                if (k1 >= 0) {
                    const uint64_t grp1 = same | (k >= 2u ? from_prev : 0ull);
                    rank1 = __popcll(grp1 & below);
                    m1 = __popcll(grp1);
                    rel1 = cw[k1];
                    if (has2) {
                        const uint64_t grp2 = to_next | (same & twice);
                        rank2 = __popcll(grp2 & below);
                        m2 = __popcll(grp2);
                        rel2 = cw[k1 + 1];
                    }
                }
                // all reads of the counts precede the writes (one wave, LDS ops in order)
                if (k1 >= 0) {
                    if (rank1 == m1 - 1)
                        cw[k1] = rel1 + m1;
                    if (has2 && rank2 == m2 - 1)
                        cw[k1 + 1] = rel2 + m2;
                    a1 = rel1 + rank1;
                    a2 = rel2 + rank2;
                }
            }
            r1[u] = a1;
            r2[u] = a2;
        }
        __syncthreads();
        // scan: thread s owns window s
        uint32_t part[WAVES], tot = 0;
        if (tid < S) {
#pragma unroll
            for (int j = 0; j < WAVES; j++) {
                part[j] = cnt[j * S + tid];
                tot += part[j];
            }
        }
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d)
                incl += t;
        }
        if (lane == 63)
            wave_sum[w] = incl;
        __syncthreads();
        uint32_t before = 0, tile_total = 0;
#pragma unroll
        for (int j = 0; j < WAVES; j++) {
            const uint32_t t = wave_sum[j];
            before += j < w ? t : 0u;
            tile_total += t;
        }
        if (tid < S) {
            uint32_t first = before + incl - tot;
            lstart[tid] = first;
            const int64_t gp = gpos[tid];
            gdelta[tid] = gp - (int64_t)first;
            gpos[tid] = gp + (int64_t)tot;
#pragma unroll
            for (int j = 0; j < WAVES; j++) {
                cnt[j * S + tid] = first;
                first += part[j];
            }
        }
        __syncthreads();
        const bool direct = tile_total > (uint32_t)CAP;         // (uniform over the workgroup)
        // pass 2
#pragma unroll
        for (int u = 0; u < TRIPS; u++) {
            const uint32_t k = key[u] & 0x7fffffffu;
            if (k == 0u)
                continue;
            const int k1 = (int)k - 1;
            const bool has2 = (key[u] >> 31) != 0u;
            const uint32_t local = (uint32_t)(w * (TRIPS * 64) + 64 * u + lane);
            const uint32_t p1 = cw[k1] + r1[u];
            const uint32_t p2 = has2 ? cw[k1 + 1] + r2[u] : 0u;
            if (!direct) {
                stage[p1] = make_double2(px[u], py[u]);
                skey[p1] = (uint8_t)k1;
                if (PERM) sidx[p1] = local;
                if (LEVEL == 1) sk16[p1] = (uint16_t)key16[LEVEL == 1 ? u : 0];
                if (has2) {
                    stage[p2] = make_double2(px[u], py[u]);
                    skey[p2] = (uint8_t)(k1 + 1);
                    if (PERM) sidx[p2] = local;
                    if (LEVEL == 1) sk16[p2] = 0;           // the first window of the next group, and no other
                }
            } else {
                const int64_t d1 = gdelta[k1] + (int64_t)p1;
                const int64_t from = tbase + local;
                x_out[d1] = px[u];
                y_out[d1] = py[u];
                if (PERM && perm) perm[d1] = idx_in ? idx_in[from] : from;
                if (PERM && z_out) z_out[d1] = src.z[from * src.stride];
                if (LEVEL == 1) gr.key_out[d1] = (uint16_t)key16[LEVEL == 1 ? u : 0];
                if (has2) {
                    const int64_t d2 = gdelta[k1 + 1] + (int64_t)p2;
                    x_out[d2] = px[u];
                    y_out[d2] = py[u];
                    if (PERM && perm) perm[d2] = idx_in ? idx_in[from] : from;
                    if (PERM && z_out) z_out[d2] = src.z[from * src.stride];
                    if (LEVEL == 1) gr.key_out[d2] = 0;
                }
            }
        }
        __syncthreads();
        // pass 3
        if (!direct) {
            for (uint32_t e = tid; e < tile_total; e += 64 * WAVES) {
                const double2 v = stage[e];
                const int64_t d = gdelta[skey[e]] + (int64_t)e;
                x_out[d] = v.x;
                y_out[d] = v.y;
                if (LEVEL == 1) gr.key_out[d] = sk16[e];
                if (PERM) {
                    // (with PERM the source index of every staged point is kept: it serves perm, and z of the
                    // point when the caller wants z scattered as well - a re-read of what this tile just loaded)
                    const int64_t from = tbase + sidx[e];
                    if (perm) perm[d] = idx_in ? idx_in[from] : from;
                    if (z_out) z_out[d] = src.z[from * src.stride];
                }
            }
        }
        // (the next tile's first barrier comes before anything read in pass 3 is written again)
    }
}

// the kernels take the number of key bits as a template argument: the next size up of a few
template <int N> struct KeyBits { static constexpr int value = N; };
template <class F> static void with_key_bits(int nbits, F &&f)
{
    if (nbits <= 3) f(KeyBits<3>{});
    else if (nbits <= 6) f(KeyBits<6>{});
    else if (nbits <= 8) f(KeyBits<8>{});
    else if (nbits <= 10) f(KeyBits<10>{});
    else if (nbits <= 12) f(KeyBits<12>{});
    else f(KeyBits<15>{});
}

struct BinPlan {
    int64_t chunks, chunk_len, segs;
    size_t counts_off, segsum_off, totals_off, bytes;
};

static BinPlan bin_plan(int64_t n, int64_t S)
{
    BinPlan p;
    const int64_t Sg = std::max<int64_t>(S, 1);
    int64_t chunks = std::max<int64_t>(1, (n + 4095) / 4096);
    static const int64_t cap = [] { const char *e = getenv("PCFIT_BIN_CHUNKS"); return e ? atoll(e) : 16384ll; }();
    chunks = std::min<int64_t>(chunks, std::min<int64_t>(cap, std::max<int64_t>(1, (32ll << 20) / Sg)));
    int64_t len = (n + chunks - 1) / std::max<int64_t>(chunks, 1);
    len = std::max<int64_t>(64, (len + 63) / 64 * 64);
    chunks = std::max<int64_t>(1, (n + len - 1) / len);
    p.chunks = chunks;
    p.chunk_len = len;
    p.segs = (chunks + BIN_SEG - 1) / BIN_SEG;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    p.counts_off = 0;
    p.segsum_off = up((size_t)chunks * Sg * 4);
    p.totals_off = p.segsum_off + up((size_t)p.segs * Sg * 8);
    p.bytes = p.totals_off + up((size_t)Sg * 8);
    return p;
}

static int bin_geom(const double *z_lo_host3, int64_t S, BinGeom &g)
{
    // z_lo_host3 = {z_lo[0], z_hi[0], z_lo[S-1]}
    g.S = S;
    g.lo0 = z_lo_host3[0];
    const double width = z_lo_host3[1] - z_lo_host3[0];
    const double step = S > 1 ? (z_lo_host3[2] - z_lo_host3[0]) / (double)(S - 1) : (width > 0 ? width : 1.0);
    PCFIT_REQUIRE(step > 0 && width >= 0 && std::isfinite(step) && std::isfinite(width),
                  "bin: windows must be ascending with finite bounds");
    PCFIT_REQUIRE(width <= step * (1.0 + 1e-7), "bin: windows overlap by more than a shared edge");
    g.inv_step = 1.0 / step;
    g.rel_width = width / step;
    return PCFIT_OK;
}

// the windows must be what analyzer.py:240-243 builds: ascending, near-uniform; geom = what the _dev calls take
int bin_check_windows(const double *z_lo, const double *z_hi, int64_t n_slices, double geom[3])
{
    const double step = n_slices > 1 ? (z_lo[n_slices - 1] - z_lo[0]) / (double)(n_slices - 1) : 1.0;
    const double width = z_hi[0] - z_lo[0];
    for (int64_t s = 0; s < n_slices; s++) {
        const double want = z_lo[0] + (double)s * step;
        PCFIT_REQUIRE(fabs(z_lo[s] - want) <= 1e-7 * step && fabs((z_hi[s] - z_lo[s]) - width) <= 1e-7 * step,
                      "bin: windows are not uniformly spaced (window %lld)", (long long)s);
    }
    geom[0] = z_lo[0];
    geom[1] = z_hi[0];
    geom[2] = z_lo[n_slices - 1];
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API size_t pcfit_bin_workspace_bytes(int64_t n_points, int64_t n_slices)
{
    return bin_plan(std::max<int64_t>(n_points, 0), n_slices).bytes;
}

namespace pcfit {

// the count pass and its scans: offsets (device, S + 1) and the workspace the scatter pass reads.
// wshift >= 0: level 1 of the two-level form - counts, scans and offsets (G + 1) are those of the G groups of
// 1 << wshift windows.
static int count_launch(const BinSrc &src, int64_t n_points, const double *z_lo, const double *z_hi, int64_t n_slices,
                        const double *window_geom, int64_t *offsets, void *workspace, size_t workspace_bytes,
                        hipStream_t st, int wshift = -1)
{
    PCFIT_REQUIRE(n_points >= 0 && n_slices >= 1, "bin: need n_points >= 0 and n_slices >= 1");
    if (n_slices > BIN_MAX_SLICES) {
        set_error("bin: at most %lld windows per call (got %lld)", (long long)BIN_MAX_SLICES, (long long)n_slices);
        return PCFIT_ERR_UNSUPPORTED;
    }
    PCFIT_REQUIRE(z_lo && z_hi && window_geom && offsets && (src.x || n_points == 0), "bin: null pointer");
    const int64_t S = wshift >= 0 ? ((n_slices - 1) >> wshift) + 1 : n_slices;     // columns: windows or groups
    const BinPlan p = bin_plan(n_points, S);
    if (!workspace || workspace_bytes < p.bytes) {
        set_error("bin: workspace of %zu bytes needed", p.bytes);
        return PCFIT_ERR_WORKSPACE;
    }
    BinGeom g;
    int rc = bin_geom(window_geom, n_slices, g);
    if (rc) return rc;
    char *ws = static_cast<char *>(workspace);
    uint32_t *counts = reinterpret_cast<uint32_t *>(ws + p.counts_off);
    int64_t *segsum = reinterpret_cast<int64_t *>(ws + p.segsum_off);
    int64_t *totals = reinterpret_cast<int64_t *>(ws + p.totals_off);
    const BinGroups gr{(int32_t)S, wshift, nullptr, nullptr, nullptr, nullptr, 0};
    if (wshift >= 0)
        bin_count_kernel<true><<<dim3((unsigned)p.chunks), dim3(64), (size_t)S * 4, st>>>(src, n_points, p.chunk_len,
                                                                                        z_lo, z_hi, g, gr, counts);
    else
        bin_count_kernel<false><<<dim3((unsigned)p.chunks), dim3(64), (size_t)S * 4, st>>>(src, n_points, p.chunk_len,
                                                                                         z_lo, z_hi, g, gr, counts);
    PCFIT_LAUNCH_CHECK();
    const dim3 cols((unsigned)((S + 255) / 256), (unsigned)p.segs);
    bin_segsum_kernel<<<cols, dim3(256), 0, st>>>(counts, p.chunks, S, segsum);
    PCFIT_LAUNCH_CHECK();
    bin_segscan_kernel<<<dim3((unsigned)S), dim3(64), 0, st>>>(segsum, p.segs, S, totals);
    PCFIT_LAUNCH_CHECK();
    bin_offsets_kernel<<<dim3(1), dim3(1024), 0, st>>>(totals, S, offsets);
    PCFIT_LAUNCH_CHECK();
    bin_rowscan_kernel<<<cols, dim3(256), 0, st>>>(counts, p.chunks, S);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

// the scatter pass.  z_out (z scattered like x and y) and idx_in (perm receives idx_in[k] instead of k) are
// for the staged kernel only, i.e. for at most BIN_STG_MAX_SLICES windows.
static int scatter_launch(const BinSrc &src, int64_t n_points, const double *z_lo, const double *z_hi,
                          int64_t n_slices, const double *window_geom, const int64_t *offsets, double *x_out,
                          double *y_out, int64_t *perm, double *z_out, const int64_t *idx_in, void *workspace,
                          size_t workspace_bytes, hipStream_t st)
{
    PCFIT_REQUIRE(n_points >= 0 && n_slices >= 1 && n_slices <= BIN_MAX_SLICES, "bin: bad sizes");
    PCFIT_REQUIRE(z_lo && z_hi && window_geom && offsets && x_out && y_out && (src.x || n_points == 0),
                  "bin: null pointer");
    const BinPlan p = bin_plan(n_points, n_slices);
    if (!workspace || workspace_bytes < p.bytes) {
        set_error("bin: workspace of %zu bytes needed (the one pcfit_bin_count_dev filled)", p.bytes);
        return PCFIT_ERR_WORKSPACE;
    }
    BinGeom g;
    int rc = bin_geom(window_geom, n_slices, g);
    if (rc) return rc;
    char *ws = static_cast<char *>(workspace);
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(ws + p.counts_off);
    const int64_t *segsum = reinterpret_cast<const int64_t *>(ws + p.segsum_off);
    int nbits = 1;
    while ((1ll << nbits) < n_slices + 2)
        nbits++;
    static const int staged = [] { const char *e = getenv("PCFIT_BIN_STAGED"); return e ? atoi(e) : 1; }();
    const bool extras = z_out != nullptr || idx_in != nullptr;
    PCFIT_REQUIRE(!extras || n_slices <= BIN_STG_MAX_SLICES, "bin: z_out / idx_in need at most %d windows",
                  BIN_STG_MAX_SLICES);
    if ((staged || extras) && n_slices <= BIN_STG_MAX_SLICES) {
        int rc2 = PCFIT_OK;
        with_key_bits(nbits, [&](auto tag) {
            constexpr int NB = decltype(tag)::value < 10 ? decltype(tag)::value : 10;    // 258 keys at most
            auto launch = [&](auto kern, size_t lds, int waves) {
                // more than 64 KB of dynamic LDS has to be granted explicitly (per device; idempotent)
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                    (void)hipGetLastError();
                    set_error("bin: %zu bytes of LDS refused", lds);
                    rc2 = PCFIT_ERR_HIP;
                    return;
                }
                kern<<<dim3((unsigned)p.chunks), dim3(64 * waves), lds, st>>>(src, n_points, p.chunk_len, z_lo, z_hi, g,
                                                                              counts, segsum, offsets, x_out, y_out,
                                                                              perm, z_out, idx_in, BinGroups{});
            };
            // Tiles of 2048 points on four waves. Measured against 4096 x 4 and 4096 x 8: longer tiles make
            // longer runs (50 M points in random order, 57 windows, no perm: 0.47 vs 0.56 ms) but leave one or two
            // workgroups per CU, which loses more wherever the input is partly ordered, small, or perm is wanted.
            if (perm || z_out)
                launch(&bin_scatter_staged_kernel<NB, true, 2048, 4>, bin_stage_lds<2048, 4, true>(n_slices), 4);
            else
                launch(&bin_scatter_staged_kernel<NB, false, 2048, 4>, bin_stage_lds<2048, 4, false>(n_slices), 4);
        });
        if (rc2) return rc2;
        PCFIT_LAUNCH_CHECK();
        return PCFIT_OK;
    }
    with_key_bits(nbits, [&](auto tag) {
        constexpr int NB = decltype(tag)::value;
        if (perm)
            bin_scatter_kernel<NB, true><<<dim3((unsigned)p.chunks), dim3(64), (size_t)n_slices * 4, st>>>(
                src, n_points, p.chunk_len, z_lo, z_hi, g, counts, segsum, offsets, x_out, y_out, perm);
        else
            bin_scatter_kernel<NB, false><<<dim3((unsigned)p.chunks), dim3(64), (size_t)n_slices * 4, st>>>(
                src, n_points, p.chunk_len, z_lo, z_hi, g, counts, segsum, offsets, x_out, y_out, perm);
    });
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

// the staged kernel at one of the two levels (chunks workgroups)
template <int LEVEL>
static int staged_level_launch(const BinSrc &src, int64_t n, int64_t chunks, int64_t chunk_len, const double *z_lo,
                               const double *z_hi, const BinGeom &g, const uint32_t *counts, const int64_t *segsum,
                               const int64_t *offsets, double *x_out, double *y_out, int64_t *perm,
                               const int64_t *idx_in, const BinGroups &gr, hipStream_t st)
{
    const int buckets = LEVEL == 1 ? gr.G : 1 << gr.wshift;
    int nbits = 1;
    while ((1 << nbits) < buckets + 2)
        nbits++;
    int rc = PCFIT_OK;
    with_key_bits(nbits, [&](auto tag) {
        constexpr int NB = decltype(tag)::value < 8 ? decltype(tag)::value : 8;     // 130 keys at most
        auto launch = [&](auto kern, size_t lds) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                set_error("bin: %zu bytes of LDS refused", lds);
                rc = PCFIT_ERR_HIP;
                return;
            }
            kern<<<dim3((unsigned)chunks), dim3(256), lds, st>>>(src, n, chunk_len, z_lo, z_hi, g, counts, segsum,
                                                                 offsets, x_out, y_out, perm, nullptr, idx_in, gr);
        };
        // Tiles of 2048 points on four waves, 4096 at level 2 without perm.  Measured on 100 M points in random z
        // order, 2000 windows: level 2 0.97 ms with tiles of 2048, 0.73 ms with 4096 (runs of 64 points instead of
        // 32); with perm 1.51 against 1.88 ms (one workgroup per CU); level 1 1.09 against 1.82 ms.
        if (perm)
            launch(&bin_scatter_staged_kernel<NB, true, 2048, 4, LEVEL>, bin_stage_lds<2048, 4, true, LEVEL>(buckets));
        else if (LEVEL == 2)
            launch(&bin_scatter_staged_kernel<NB, false, 4096, 4, LEVEL>, bin_stage_lds<4096, 4, false, LEVEL>(buckets));
        else
            launch(&bin_scatter_staged_kernel<NB, false, 2048, 4, LEVEL>, bin_stage_lds<2048, 4, false, LEVEL>(buckets));
    });
    if (rc) return rc;
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

constexpr int64_t BIN_L2_CHUNK = 4096;       // entries per level-2 workgroup

// Two levels for many windows.  With more than BIN_STG_MAX_SLICES windows the scatter cannot order its tiles in
// LDS, and storing every point on its own costs several times the output in HBM writes when the points come in no
// z order.  Level 1 finds every point's window(s) as the single pass does and scatters x, y and a 16-bit key (and
// the source index when perm is wanted) into groups of 64 or 128 consecutive windows; level 2 bins every group's
// entries into the group's windows by their keys.  Both levels run the staged kernel, level 2 for all groups in
// one launch.  Windows are decided once, by the same code on the same z, and both levels are stable, so the
// result - members, order, bytes - is that of the single pass.
static int bin_two_level(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t S,
                         bool want_perm, int64_t capacity, int64_t *offsets, DevBuf &doff, DevBuf &dx, DevBuf &dy,
                         DevBuf &dperm, const DevBuf &dlo, const DevBuf &dhi, hipStream_t st, const double *dz)
{
    const int wshift = S <= 4096 ? 6 : 7;
    const int64_t W = 1ll << wshift, G = ((S - 1) >> wshift) + 1;
    double geom[3];
    PCFIT_TRY(bin_check_windows(z_lo, z_hi, S, geom));
    BinGeom g;
    PCFIT_TRY(bin_geom(geom, S, g));
    const BinSrc rows = rows_of(dxyz, dz);
    // level 1
    const BinPlan p1 = bin_plan(n_points, G);
    DevBuf dgoff, dws1, x1, y1, k1, i1;
    PCFIT_TRY(dgoff.alloc((size_t)(G + 1) * 8));
    PCFIT_TRY(dws1.alloc(p1.bytes));
    PCFIT_TRY(count_launch(rows, n_points, dlo.as<double>(), dhi.as<double>(), S, geom, dgoff.as<int64_t>(), dws1.p,
                           dws1.bytes, st, wshift));
    std::vector<int64_t> goff((size_t)G + 1), cbase((size_t)G + 1, 0);
    PCFIT_TRY(dgoff.download(goff.data(), goff.size() * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    const int64_t n1 = goff[(size_t)G];
    for (int64_t k = 0; k < G; k++) {
        const int64_t ng = goff[(size_t)k + 1] - goff[(size_t)k];
        PCFIT_REQUIRE(ng < (1ll << 32), "bin: more than 2^32 points in %lld consecutive windows", (long long)W);
        cbase[(size_t)k + 1] = cbase[(size_t)k] + (ng + BIN_L2_CHUNK - 1) / BIN_L2_CHUNK;
    }
    const int64_t C2 = cbase[(size_t)G];
    PCFIT_TRY(x1.alloc((size_t)n1 * 8));
    PCFIT_TRY(y1.alloc((size_t)n1 * 8));
    PCFIT_TRY(k1.alloc((size_t)n1 * 2));
    if (want_perm) PCFIT_TRY(i1.alloc((size_t)n1 * 8));
    if (n1 > 0) {
        const BinGroups gr1{(int32_t)G, wshift, nullptr, k1.as<uint16_t>(), nullptr, nullptr, n1};
        const char *ws = static_cast<const char *>(dws1.p);
        PCFIT_TRY(staged_level_launch<1>(rows, n_points, p1.chunks, p1.chunk_len, dlo.as<double>(), dhi.as<double>(), g,
                                         reinterpret_cast<const uint32_t *>(ws + p1.counts_off),
                                         reinterpret_cast<const int64_t *>(ws + p1.segsum_off), dgoff.as<int64_t>(),
                                         x1.as<double>(), y1.as<double>(), want_perm ? i1.as<int64_t>() : nullptr,
                                         nullptr, gr1, st));
    }
    // level 2: counts per chunk and window, scanned down every group's chunks; the windows' totals give the offsets
    DevBuf dcbase, counts2, totals;
    PCFIT_TRY(dcbase.upload(cbase.data(), cbase.size() * 8, st));
    PCFIT_TRY(counts2.alloc(std::max<size_t>((size_t)C2 * (size_t)W * 4, 8)));
    PCFIT_TRY(totals.alloc((size_t)S * 8));
    const BinGroups gr2{(int32_t)G, wshift, k1.as<uint16_t>(), nullptr, dgoff.as<int64_t>(), dcbase.as<int64_t>(), n1};
    if (C2 > 0) {
        bin2_count_kernel<<<dim3((unsigned)C2), dim3(64), 0, st>>>(gr2, BIN_L2_CHUNK, counts2.as<uint32_t>());
        PCFIT_LAUNCH_CHECK();
    }
    bin2_colscan_kernel<<<dim3((unsigned)(G * W)), dim3(64), 0, st>>>(gr2, S, counts2.as<uint32_t>(),
                                                                      totals.as<int64_t>());
    PCFIT_LAUNCH_CHECK();
    bin_offsets_kernel<<<dim3(1), dim3(1024), 0, st>>>(totals.as<int64_t>(), S, doff.as<int64_t>());
    PCFIT_LAUNCH_CHECK();
    PCFIT_TRY(doff.download(offsets, (size_t)(S + 1) * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    const int64_t total = offsets[S];
    if (capacity >= 0 && total > capacity)
        return PCFIT_OK;                       // (the caller reports it; nothing is scattered)
    PCFIT_TRY(dx.alloc((size_t)total * 8));
    PCFIT_TRY(dy.alloc((size_t)total * 8));
    if (want_perm) PCFIT_TRY(dperm.alloc((size_t)total * 8));
    if (total > 0) {
        const BinSrc cols{x1.as<double>(), y1.as<double>(), nullptr, 1};
        PCFIT_TRY(staged_level_launch<2>(cols, n1, C2, BIN_L2_CHUNK, nullptr, nullptr, g, counts2.as<uint32_t>(),
                                         nullptr, doff.as<int64_t>(), dx.as<double>(), dy.as<double>(),
                                         want_perm ? dperm.as<int64_t>() : nullptr,
                                         want_perm ? i1.as<int64_t>() : nullptr, gr2, st));
    }
    // (the level-1 arrays are released when this returns: the stream runs the scatter first, as every other
    // block handed back to the cache relies on)
    return PCFIT_OK;
}

// More windows than one count / scatter pass takes (a long scan cut very finely): the windows go in runs of
// BIN_MAX_SLICES consecutive ones, every run a count and a scatter pass of its own over the scan (a point outside
// a run's z range finds no window there), the runs' outputs one after the other.  Same members, order and bytes
// as one pass would give.
static int bin_in_runs(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t S,
                       bool want_perm, int64_t capacity, int64_t *offsets, DevBuf &doff, DevBuf &dx, DevBuf &dy,
                       DevBuf &dperm, const DevBuf &dlo, const DevBuf &dhi, hipStream_t st, const double *dz)
{
    const int64_t W = BIN_MAX_SLICES, G = (S + W - 1) / W;
    const BinSrc rows = rows_of(dxyz, dz);
    std::vector<DevBuf> ws((size_t)G), off((size_t)G);
    std::vector<double> geom((size_t)G * 3);
    std::vector<int64_t> hoff;
    std::vector<int64_t> base((size_t)G + 1, 0);
    for (int64_t g = 0; g < G; g++) {
        const int64_t s0 = g * W, Sg = std::min<int64_t>(W, S - s0);
        PCFIT_TRY(bin_check_windows(z_lo + s0, z_hi + s0, Sg, &geom[(size_t)g * 3]));
        PCFIT_TRY(ws[(size_t)g].alloc(bin_plan(n_points, Sg).bytes));
        PCFIT_TRY(off[(size_t)g].alloc((size_t)(Sg + 1) * 8));
        PCFIT_TRY(count_launch(rows, n_points, dlo.as<double>() + s0, dhi.as<double>() + s0, Sg, &geom[(size_t)g * 3],
                               off[(size_t)g].as<int64_t>(), ws[(size_t)g].p, ws[(size_t)g].bytes, st));
        hoff.resize((size_t)Sg + 1);
        PCFIT_TRY(off[(size_t)g].download(hoff.data(), hoff.size() * 8, st));
        PCFIT_HIP(hipStreamSynchronize(st));
        for (int64_t j = 0; j < Sg; j++)
            offsets[s0 + j] = base[(size_t)g] + hoff[(size_t)j];
        base[(size_t)g + 1] = base[(size_t)g] + hoff[(size_t)Sg];
    }
    const int64_t total = base[(size_t)G];
    offsets[S] = total;
    if (capacity >= 0 && total > capacity)
        return PCFIT_OK;
    PCFIT_TRY(doff.upload(offsets, (size_t)(S + 1) * 8, st));
    PCFIT_TRY(dx.alloc((size_t)total * 8));
    PCFIT_TRY(dy.alloc((size_t)total * 8));
    if (want_perm) PCFIT_TRY(dperm.alloc((size_t)total * 8));
    for (int64_t g = 0; g < G; g++) {
        const int64_t s0 = g * W, Sg = std::min<int64_t>(W, S - s0);
        if (base[(size_t)g + 1] == base[(size_t)g])
            continue;
        PCFIT_TRY(scatter_launch(rows, n_points, dlo.as<double>() + s0, dhi.as<double>() + s0, Sg,
                                 &geom[(size_t)g * 3], off[(size_t)g].as<int64_t>(),
                                 dx.as<double>() + base[(size_t)g], dy.as<double>() + base[(size_t)g],
                                 want_perm ? dperm.as<int64_t>() + base[(size_t)g] : nullptr, nullptr, nullptr,
                                 ws[(size_t)g].p, ws[(size_t)g].bytes, st));
    }
    return PCFIT_OK;
}

// Bin the scan on the device: offsets to the host (and to doff), x, y (and perm) left in dx, dy (dperm).
// With capacity >= 0 and more members than that, only the offsets are produced.
int bin_on_device(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t S,
                  bool want_perm, int64_t capacity, int64_t *offsets, DevBuf &doff, DevBuf &dx, DevBuf &dy,
                  DevBuf &dperm, hipStream_t st, const double *dz)
{
    PCFIT_REQUIRE(n_points >= 0 && S >= 1, "bin: need n_points >= 0 and n_slices >= 1");
    PCFIT_REQUIRE(z_lo && z_hi && offsets, "bin: null pointer");
    PCFIT_REQUIRE(S < (1ll << 31), "bin: too many windows");
    double geom[3];
    PCFIT_TRY(bin_check_windows(z_lo, z_hi, S, geom));
    DevBuf dlo, dhi, dws;
    PCFIT_TRY(dlo.upload(z_lo, (size_t)S * 8, st));
    PCFIT_TRY(dhi.upload(z_hi, (size_t)S * 8, st));
    PCFIT_TRY(doff.alloc((size_t)(S + 1) * 8));
    if (S > BIN_MAX_SLICES)
        return bin_in_runs(dxyz, n_points, z_lo, z_hi, S, want_perm, capacity, offsets, doff, dx, dy, dperm, dlo, dhi,
                           st, dz);
    // (read at every call so that a test can switch between the two forms; negative = never)
    const char *env = getenv("PCFIT_BIN_TWO_LEVEL_MIN");
    const int64_t two_level_min = env ? atoll(env) : BIN_TWO_LEVEL_MIN;
    if (S > BIN_STG_MAX_SLICES && two_level_min >= 0 && n_points >= two_level_min)
        return bin_two_level(dxyz, n_points, z_lo, z_hi, S, want_perm, capacity, offsets, doff, dx, dy, dperm, dlo,
                             dhi, st, dz);
    const BinSrc rows = rows_of(dxyz, dz);
    PCFIT_TRY(dws.alloc(bin_plan(n_points, S).bytes));
    PCFIT_TRY(count_launch(rows, n_points, dlo.as<double>(), dhi.as<double>(), S, geom, doff.as<int64_t>(), dws.p,
                           dws.bytes, st));
    PCFIT_TRY(doff.download(offsets, (size_t)(S + 1) * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    const int64_t total = offsets[S];
    if (capacity >= 0 && total > capacity)
        return PCFIT_OK;
    PCFIT_TRY(dx.alloc((size_t)total * 8));
    PCFIT_TRY(dy.alloc((size_t)total * 8));
    if (want_perm) PCFIT_TRY(dperm.alloc((size_t)total * 8));
    if (total > 0)
        PCFIT_TRY(scatter_launch(rows, n_points, dlo.as<double>(), dhi.as<double>(), S, geom, doff.as<int64_t>(),
                                 dx.as<double>(), dy.as<double>(), want_perm ? dperm.as<int64_t>() : nullptr, nullptr,
                                 nullptr, dws.p, dws.bytes, st));
    return PCFIT_OK;
}

}  // namespace pcfit

PCFIT_API int pcfit_bin_count_dev(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                                  int64_t n_slices, const double *window_geom, int64_t *offsets, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    return count_launch(rows_of(xyz), n_points, z_lo, z_hi, n_slices, window_geom, offsets, workspace, workspace_bytes,
                        as_stream(stream));
}

PCFIT_API int pcfit_bin_scatter_dev(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                                    int64_t n_slices, const double *window_geom, const int64_t *offsets,
                                    double *x_out, double *y_out, int64_t *perm, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    return scatter_launch(rows_of(xyz), n_points, z_lo, z_hi, n_slices, window_geom, offsets, x_out, y_out, perm,
                          nullptr, nullptr, workspace, workspace_bytes, as_stream(stream));
}

// ---- the two-level form for callers that hold everything on the device (no read-back in between) ----
namespace pcfit {

struct Bin2Plan {
    int wshift;
    int64_t W, G, cap1, max_chunks;
    BinPlan p1;
    size_t ws1_off, goff_off, cbase_off, totals_off, counts2_off, x1_off, y1_off, i1_off, k1_off, bytes;
};

static Bin2Plan bin2_plan(int64_t n, int64_t S, bool want_perm, int64_t max_entries)
{
    Bin2Plan q;
    q.wshift = S <= 4096 ? 6 : 7;
    q.W = 1ll << q.wshift;
    q.G = ((std::max<int64_t>(S, 1) - 1) >> q.wshift) + 1;
    q.cap1 = max_entries > 0 ? max_entries : 2 * std::max<int64_t>(n, 0);      // a point makes two entries at most
    q.max_chunks = q.cap1 / BIN_L2_CHUNK + q.G;
    q.p1 = bin_plan(n, q.G);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += up(bytes); return o; };
    q.ws1_off = take(q.p1.bytes);
    q.goff_off = take((size_t)(q.G + 1) * 8);
    q.cbase_off = take((size_t)(q.G + 1) * 8);
    q.totals_off = take((size_t)std::max<int64_t>(S, 1) * 8);
    q.counts2_off = take((size_t)q.max_chunks * (size_t)q.W * 4);
    q.x1_off = take((size_t)q.cap1 * 8);
    q.y1_off = take((size_t)q.cap1 * 8);
    q.i1_off = take(want_perm ? (size_t)q.cap1 * 8 : 0);
    q.k1_off = take((size_t)q.cap1 * 2);
    q.bytes = at;
    return q;
}

static int bin2_check(int64_t n_points, int64_t n_slices, const void *workspace, size_t workspace_bytes,
                      const Bin2Plan &q)
{
    PCFIT_REQUIRE(n_points >= 0 && n_slices >= 1, "bin: need n_points >= 0 and n_slices >= 1");
    if (n_slices > BIN_MAX_SLICES) {
        set_error("bin: at most %lld windows per call (got %lld)", (long long)BIN_MAX_SLICES, (long long)n_slices);
        return PCFIT_ERR_UNSUPPORTED;
    }
    if (!workspace || workspace_bytes < q.bytes) {
        set_error("bin: workspace of %zu bytes needed", q.bytes);
        return PCFIT_ERR_WORKSPACE;
    }
    return PCFIT_OK;
}

}  // namespace pcfit

PCFIT_API size_t pcfit_bin2_workspace_bytes(int64_t n_points, int64_t n_slices, int want_perm, int64_t max_entries)
{
    return bin2_plan(std::max<int64_t>(n_points, 0), n_slices, want_perm != 0, max_entries).bytes;
}

PCFIT_API int pcfit_bin2_count_dev(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                                   int64_t n_slices, const double *window_geom, int want_perm, int64_t max_entries,
                                   int64_t *offsets, int32_t *status, void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    PCFIT_ENTER();
    const Bin2Plan q = bin2_plan(std::max<int64_t>(n_points, 0), n_slices, want_perm != 0, max_entries);
    PCFIT_TRY(bin2_check(n_points, n_slices, workspace, workspace_bytes, q));
    PCFIT_REQUIRE(z_lo && z_hi && window_geom && offsets && status && (xyz || n_points == 0), "bin: null pointer");
    hipStream_t st = as_stream(stream);
    BinGeom g;
    PCFIT_TRY(bin_geom(window_geom, n_slices, g));
    char *ws = static_cast<char *>(workspace);
    int64_t *goff = reinterpret_cast<int64_t *>(ws + q.goff_off), *cbase = reinterpret_cast<int64_t *>(ws + q.cbase_off);
    int64_t *totals = reinterpret_cast<int64_t *>(ws + q.totals_off);
    uint32_t *counts2 = reinterpret_cast<uint32_t *>(ws + q.counts2_off);
    double *x1 = reinterpret_cast<double *>(ws + q.x1_off), *y1 = reinterpret_cast<double *>(ws + q.y1_off);
    int64_t *i1 = want_perm ? reinterpret_cast<int64_t *>(ws + q.i1_off) : nullptr;
    uint16_t *k1 = reinterpret_cast<uint16_t *>(ws + q.k1_off);
    const BinSrc rows = rows_of(xyz);
    PCFIT_TRY(count_launch(rows, n_points, z_lo, z_hi, n_slices, window_geom, goff, ws + q.ws1_off, q.p1.bytes, st,
                           q.wshift));
    bin2_plan_kernel<<<dim3(1), dim3(64), 0, st>>>(goff, (int32_t)q.G, q.cap1, BIN_L2_CHUNK, cbase, status);
    PCFIT_LAUNCH_CHECK();
    const BinGroups gr1{(int32_t)q.G, q.wshift, nullptr, k1, nullptr, nullptr, q.cap1};
    if (n_points > 0)
        PCFIT_TRY(staged_level_launch<1>(rows, n_points, q.p1.chunks, q.p1.chunk_len, z_lo, z_hi, g,
                                         reinterpret_cast<const uint32_t *>(ws + q.ws1_off + q.p1.counts_off),
                                         reinterpret_cast<const int64_t *>(ws + q.ws1_off + q.p1.segsum_off), goff, x1, y1,
                                         i1, nullptr, gr1, st));
    const BinGroups gr2{(int32_t)q.G, q.wshift, k1, nullptr, goff, cbase, q.cap1};
    bin2_count_kernel<<<dim3((unsigned)q.max_chunks), dim3(64), 0, st>>>(gr2, BIN_L2_CHUNK, counts2);
    PCFIT_LAUNCH_CHECK();
    bin2_colscan_kernel<<<dim3((unsigned)(q.G * q.W)), dim3(64), 0, st>>>(gr2, n_slices, counts2, totals);
    PCFIT_LAUNCH_CHECK();
    bin_offsets_kernel<<<dim3(1), dim3(1024), 0, st>>>(totals, n_slices, offsets);
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

PCFIT_API int pcfit_bin2_scatter_dev(int64_t n_points, int64_t n_slices, const double *window_geom, int want_perm,
                                     int64_t max_entries, const int64_t *offsets, double *x_out, double *y_out,
                                     int64_t *perm, void *workspace, size_t workspace_bytes, void *stream)
{
    PCFIT_ENTER();
    const Bin2Plan q = bin2_plan(std::max<int64_t>(n_points, 0), n_slices, want_perm != 0, max_entries);
    PCFIT_TRY(bin2_check(n_points, n_slices, workspace, workspace_bytes, q));
    PCFIT_REQUIRE(window_geom && offsets && x_out && y_out, "bin: null pointer");
    PCFIT_REQUIRE(!perm || want_perm, "bin: perm needs the count step run with want_perm");
    BinGeom g;
    PCFIT_TRY(bin_geom(window_geom, n_slices, g));
    char *ws = static_cast<char *>(workspace);
    const BinGroups gr2{(int32_t)q.G, q.wshift, reinterpret_cast<const uint16_t *>(ws + q.k1_off), nullptr,
                        reinterpret_cast<const int64_t *>(ws + q.goff_off),
                        reinterpret_cast<const int64_t *>(ws + q.cbase_off), q.cap1};
    const BinSrc cols{reinterpret_cast<const double *>(ws + q.x1_off), reinterpret_cast<const double *>(ws + q.y1_off),
                      nullptr, 1};
    return staged_level_launch<2>(cols, q.cap1, q.max_chunks, BIN_L2_CHUNK, nullptr, nullptr, g,
                                  reinterpret_cast<const uint32_t *>(ws + q.counts2_off), nullptr, offsets, x_out, y_out,
                                  perm, perm ? reinterpret_cast<const int64_t *>(ws + q.i1_off) : nullptr, gr2,
                                  as_stream(stream));
}

namespace pcfit {

__global__ __launch_bounds__(256) void z_column_kernel(const double *__restrict__ xyz, int64_t n, double *__restrict__ z)
{
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        z[i] = xyz[3 * i + 2];
}

int scan_z_column(pcfit_scan &scan, hipStream_t st)
{
    const char *env = getenv("PCFIT_SCAN_Z_COLUMN");
    if ((env && env[0] == '0') || scan.n_points <= 0) return PCFIT_OK;
    if (scan.z.alloc((size_t)scan.n_points * 8) != PCFIT_OK) {
        (void)hipGetLastError();               // (memory is short: the count passes read the rows)
        return PCFIT_OK;
    }
    const int64_t blocks = std::min<int64_t>(8192, (scan.n_points + 255) / 256);
    z_column_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(scan.xyz.as<double>(), scan.n_points, scan.z.as<double>());
    PCFIT_LAUNCH_CHECK();
    return PCFIT_OK;
}

}  // namespace pcfit

// bin on the device, slices to the host: the scan is on the device already
static int bin_resident(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t n_slices,
                        int64_t *offsets, double *x_out, double *y_out, int64_t *perm, int64_t capacity,
                        int64_t *total, const double *dz = nullptr)
{
    PCFIT_REQUIRE(total, "bin: null pointer");
    hipStream_t st = nullptr;
    DevBuf doff, dx, dy, dperm;
    const int64_t room = (x_out == nullptr || y_out == nullptr) ? 0 : std::max<int64_t>(capacity, 0);
    PCFIT_TRY(bin_on_device(dxyz, n_points, z_lo, z_hi, n_slices, perm != nullptr, room, offsets, doff, dx, dy, dperm,
                            st, dz));
    *total = offsets[n_slices];
    if (*total > room) {
        set_error("bin: outputs hold %lld points, %lld needed", (long long)capacity, (long long)*total);
        return PCFIT_ERR_INVALID;
    }
    PCFIT_TRY(dx.download(x_out, (size_t)*total * 8, st));
    PCFIT_TRY(dy.download(y_out, (size_t)*total * 8, st));
    if (perm) PCFIT_TRY(dperm.download(perm, (size_t)*total * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_bin_slices(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                               int64_t n_slices, int64_t *offsets, double *x_out, double *y_out, int64_t *perm,
                               int64_t capacity, int64_t *total)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(n_points >= 0 && (xyz || n_points == 0), "bin: null pointer");
    DevBuf dxyz;
    PCFIT_TRY(dxyz.upload(xyz, (size_t)n_points * 24, nullptr));
    return bin_resident(dxyz.as<double>(), n_points, z_lo, z_hi, n_slices, offsets, x_out, y_out, perm, capacity,
                        total);
}

PCFIT_API int pcfit_scan_bin_slices(const pcfit_scan *scan, const double *z_lo, const double *z_hi, int64_t n_slices,
                                    int64_t *offsets, double *x_out, double *y_out, int64_t *perm, int64_t capacity,
                                    int64_t *total)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(scan, "bin: scan is null");
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    PCFIT_REQUIRE(dev == scan->device, "bin: the scan lives on device %d, the current device is %d", scan->device, dev);
    return bin_resident(scan->xyz.as<double>(), scan->n_points, z_lo, z_hi, n_slices, offsets, x_out, y_out, perm,
                        capacity, total, scan->z.as<double>());
}

// members per window of the resident scan: the count pass alone
PCFIT_API int pcfit_scan_window_counts(const pcfit_scan *scan, const double *z_lo, const double *z_hi,
                                       int64_t n_slices, int64_t *counts)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(scan, "window_counts: scan is null");
    PCFIT_REQUIRE(n_slices >= 1 && z_lo && z_hi && counts, "window_counts: need n_slices >= 1 and non-null arrays");
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    PCFIT_REQUIRE(dev == scan->device, "window_counts: the scan lives on device %d, the current device is %d",
                  scan->device, dev);
    hipStream_t st = nullptr;
    DevBuf dlo, dhi, doff, dws;
    PCFIT_TRY(dlo.upload(z_lo, (size_t)n_slices * 8, st));
    PCFIT_TRY(dhi.upload(z_hi, (size_t)n_slices * 8, st));
    std::vector<int64_t> off;
    for (int64_t s0 = 0; s0 < n_slices; s0 += BIN_MAX_SLICES) {          // (runs of windows, as bin_in_runs)
        const int64_t Sg = std::min<int64_t>(BIN_MAX_SLICES, n_slices - s0);
        double geom[3];
        PCFIT_TRY(bin_check_windows(z_lo + s0, z_hi + s0, Sg, geom));
        PCFIT_TRY(doff.alloc((size_t)(Sg + 1) * 8));
        PCFIT_TRY(dws.alloc(pcfit_bin_workspace_bytes(scan->n_points, Sg)));
        PCFIT_TRY(count_launch(rows_of(scan->xyz.as<double>(), scan->z.as<double>()), scan->n_points,
                               dlo.as<double>() + s0, dhi.as<double>() + s0, Sg, geom, doff.as<int64_t>(), dws.p, dws.bytes,
                               st));
        off.resize((size_t)Sg + 1);
        PCFIT_TRY(doff.download(off.data(), off.size() * 8, st));
        PCFIT_HIP(hipStreamSynchronize(st));
        for (int64_t s = 0; s < Sg; s++)
            counts[s0 + s] = off[(size_t)s + 1] - off[(size_t)s];
    }
    return PCFIT_OK;
}
