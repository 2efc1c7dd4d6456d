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
//            the host-level calls bin more windows than that in two levels
//            when the scan is large (bin_two_level).
//
// Windows are closed intervals that may share an edge, so a point can belong
// to two consecutive windows (both get a copy) or to none.  Membership is
// decided by the caller's z_lo / z_hi arrays, compared exactly; the uniform
// spacing is used only to find the candidate window and to skip the array
// look-ups for points that are clearly interior.
// HBM traffic: 24 B/point read in count, 24 B read + 16 B (+8 B perm) written
// in scatter.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "pcfit_internal.h"

namespace pcfit {

constexpr int BIN_SEG = 64;            // chunk rows per scan segment
constexpr int BIN_UNROLL = 4;          // 64-point trips whose loads are issued together
constexpr int64_t BIN_MAX_SLICES = 16384;
constexpr double BIN_EDGE_EPS = 1e-6;  // in units of the window step
// Points from which more than BIN_STG_MAX_SLICES windows are binned in two levels. Measured on resident scans in
// random z order: 100 M points / 2000 windows 10.7 -> 9.3 ms per analysis, 50 M / 400 windows 3.53 -> 3.36 ms,
// 10 M / 400 windows 1.60 -> 1.70 ms (the second count pass and the extra launches cost more than they save).
constexpr int64_t BIN_TWO_LEVEL_MIN = 32'000'000;

struct BinGeom {
    double lo0, inv_step, rel_width;   // rel_width = (z_hi[0]-z_lo[0]) / step
    int64_t S;
};

// where the points are read from: coordinate i of point k at x[k * stride] etc. (a scan of xyz rows: x = xyz,
// y = xyz + 1, z = xyz + 2, stride 3; three separate arrays: stride 1)
struct BinSrc {
    const double *x, *y, *z;
    int64_t stride;
};
static BinSrc rows_of(const double *xyz) { return BinSrc{xyz, xyz + 1, xyz + 2, 3}; }

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

__global__ __launch_bounds__(64) void bin_count_kernel(BinSrc src, int64_t n, int64_t chunk_len,
                                                      const double *__restrict__ z_lo,
                                                      const double *__restrict__ z_hi, BinGeom g,
                                                      uint32_t *__restrict__ counts)
{
    extern __shared__ uint32_t lhist[];
    const int64_t c = blockIdx.x;
    for (int64_t s = threadIdx.x; s < g.S; s += 64)
        lhist[s] = 0;
    __syncthreads();
    const int64_t begin = c * chunk_len, end = min(n, begin + chunk_len);
    // BIN_UNROLL loads of z in flight per lane: one wave per chunk has nothing else to hide the latency with
    for (int64_t base = begin + threadIdx.x; base < end; base += 64 * BIN_UNROLL) {
        double z[BIN_UNROLL];
#pragma unroll
        for (int u = 0; u < BIN_UNROLL; u++) {
            const int64_t i = base + 64 * u;
            z[u] = i < end ? src.z[i * src.stride] : __longlong_as_double(0x7ff8000000000000ll);   // NaN: no window
        }
#pragma unroll
        for (int u = 0; u < BIN_UNROLL; u++) {
            int32_t k1;
            bool has2;
            windows_of(z[u], g, z_lo, z_hi, k1, has2);
            if (k1 >= 0) {
                atomicAdd(&lhist[k1], 1u);
                if (has2)
                    atomicAdd(&lhist[k1 + 1], 1u);
            }
        }
    }
    __syncthreads();
    for (int64_t s = threadIdx.x; s < g.S; s += 64)
        counts[c * g.S + s] = lhist[s];
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

template <int TILE, int WAVES, bool PERM> static size_t bin_stage_lds(int64_t S)
{
    const size_t cap = (size_t)bin_stage_cap<TILE>();
    return cap * 16 + (size_t)S * 16 + ((size_t)WAVES * S + (size_t)S + 1 + WAVES) * 4 +
           (PERM ? cap * 4 : 0) + cap;
}

template <int NBITS, bool PERM, int TILE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bin_scatter_staged_kernel(
    BinSrc src, int64_t n, int64_t chunk_len, const double *__restrict__ z_lo,
    const double *__restrict__ z_hi, BinGeom g, const uint32_t *__restrict__ counts,
    const int64_t *__restrict__ segbase, const int64_t *__restrict__ offsets, double *__restrict__ x_out,
    double *__restrict__ y_out, int64_t *__restrict__ perm, double *__restrict__ z_out,
    const int64_t *__restrict__ idx_in)
{
    constexpr int TRIPS = TILE / (64 * WAVES);
    constexpr int CAP = bin_stage_cap<TILE>();
    extern __shared__ __align__(16) unsigned char stg_lds[];
    const int S = (int)g.S;
    double2 *stage = reinterpret_cast<double2 *>(stg_lds);
    int64_t *gpos = reinterpret_cast<int64_t *>(stg_lds + (size_t)CAP * 16);   // next output slot per window
    int64_t *gdelta = gpos + S;                                                  // output slot - LDS slot
    uint32_t *cnt = reinterpret_cast<uint32_t *>(gdelta + S);                    // [wave][window]
    uint32_t *lstart = cnt + WAVES * S;                                  // S + 1
    uint32_t *wave_sum = lstart + S + 1;                                         // WAVES
    uint32_t *sidx = wave_sum + WAVES;                                   // CAP, with PERM
    uint8_t *skey = reinterpret_cast<uint8_t *>(sidx + (PERM ? CAP : 0));        // CAP

    const int64_t c = chunk_of_block(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t *cw = cnt + w * S;
    for (int s = tid; s < S; s += 64 * WAVES)
        gpos[s] = offsets[s] + segbase[(c / BIN_SEG) * g.S + s] + (int64_t)counts[c * g.S + s];
    const int64_t begin = c * chunk_len, end = min(n, begin + chunk_len);
    for (int64_t tbase = begin; tbase < end; tbase += TILE) {
        for (int s = tid; s < WAVES * S; s += 64 * WAVES)
            cnt[s] = 0;
        __syncthreads();
        // pass 1
        double px[TRIPS], py[TRIPS];
        uint32_t key[TRIPS], r1[TRIPS], r2[TRIPS];      // key: window + 1, bit 31 = also in the next window
        const int64_t wbase = tbase + (int64_t)w * (TRIPS * 64);
        {
            // every load of the tile is issued before the first is used; lanes past the end read the
            // chunk's last point and get NaN for z: no window
            double pz[TRIPS];
#pragma unroll
            for (int u = 0; u < TRIPS; u++) {
                const int64_t i = wbase + 64 * u + lane;
                const int64_t ic = i < end ? i : end - 1;
                px[u] = src.x[ic * src.stride];
                py[u] = src.y[ic * src.stride];
                pz[u] = src.z[ic * src.stride];
            }
#pragma unroll
            for (int u = 0; u < TRIPS; u++) {
                const int64_t i = wbase + 64 * u + lane;
                int32_t k1;
                bool has2;
                windows_of(i < end ? pz[u] : __longlong_as_double(0x7ff8000000000000ll), g, z_lo, z_hi, k1, has2);
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
                if (has2) {
                    stage[p2] = make_double2(px[u], py[u]);
                    skey[p2] = (uint8_t)(k1 + 1);
                    if (PERM) sidx[p2] = local;
                }
            } else {
                const int64_t d1 = gdelta[k1] + (int64_t)p1;
                const int64_t from = tbase + local;
                x_out[d1] = px[u];
                y_out[d1] = py[u];
                if (PERM && perm) perm[d1] = idx_in ? idx_in[from] : from;
                if (PERM && z_out) z_out[d1] = src.z[from * src.stride];
                if (has2) {
                    const int64_t d2 = gdelta[k1 + 1] + (int64_t)p2;
                    x_out[d2] = px[u];
                    y_out[d2] = py[u];
                    if (PERM && perm) perm[d2] = idx_in ? idx_in[from] : from;
                    if (PERM && z_out) z_out[d2] = src.z[from * src.stride];
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

// the count pass and its scans: offsets (device, S + 1) and the workspace the scatter pass reads
static int count_launch(const BinSrc &src, int64_t n_points, const double *z_lo, const double *z_hi, int64_t n_slices,
                        const double *window_geom, int64_t *offsets, void *workspace, size_t workspace_bytes,
                        hipStream_t st)
{
    PCFIT_REQUIRE(n_points >= 0 && n_slices >= 1, "bin: need n_points >= 0 and n_slices >= 1");
    if (n_slices > BIN_MAX_SLICES) {
        set_error("bin: at most %lld windows per call (got %lld)", (long long)BIN_MAX_SLICES, (long long)n_slices);
        return PCFIT_ERR_UNSUPPORTED;
    }
    PCFIT_REQUIRE(z_lo && z_hi && window_geom && offsets && (src.x || n_points == 0), "bin: null pointer");
    const BinPlan p = bin_plan(n_points, n_slices);
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
    const int64_t S = n_slices;
    bin_count_kernel<<<dim3((unsigned)p.chunks), dim3(64), (size_t)S * 4, st>>>(src, n_points, p.chunk_len, z_lo, z_hi,
                                                                              g, counts);
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
                                                                              perm, z_out, idx_in);
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

// Two levels for many windows.  With more than BIN_STG_MAX_SLICES windows the scatter cannot order its tiles in
// LDS, and storing every point on its own costs several times the output in HBM writes when the points come in no
// z order.  Level 1 scatters x, y, z (and the source index when perm is wanted) into groups of BIN_STG_MAX_SLICES
// consecutive windows, each group one closed window from the first window's lower to the last one's upper edge;
// level 2 bins every group into its own windows.  Both levels run the staged kernel.  A point on the edge shared
// by two groups goes into both and each group keeps it for its own window only, so the result - members, order,
// bytes - is that of the single pass.
static int bin_two_level(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t S,
                         bool want_perm, int64_t capacity, int64_t *offsets, DevBuf &doff, DevBuf &dx, DevBuf &dy,
                         DevBuf &dperm, const DevBuf &dlo, const DevBuf &dhi, hipStream_t st)
{
    const int64_t W = BIN_STG_MAX_SLICES, G = (S + W - 1) / W;
    // level 1: the groups as windows (the last one padded to the common width; what it takes in beyond the last
    // real window finds no window at level 2)
    std::vector<double> glo((size_t)G), ghi((size_t)G);
    for (int64_t g = 0; g < G; g++) {
        glo[(size_t)g] = z_lo[g * W];
        ghi[(size_t)g] = (g + 1) * W <= S ? z_hi[(g + 1) * W - 1] : glo[(size_t)g] + (ghi[0] - glo[0]);
    }
    double geom1[3];
    PCFIT_TRY(bin_check_windows(glo.data(), ghi.data(), G, geom1));
    DevBuf dglo, dghi, dgoff, dws1, x1, y1, z1, i1;
    PCFIT_TRY(dglo.upload(glo.data(), (size_t)G * 8, st));
    PCFIT_TRY(dghi.upload(ghi.data(), (size_t)G * 8, st));
    PCFIT_TRY(dgoff.alloc((size_t)(G + 1) * 8));
    PCFIT_TRY(dws1.alloc(bin_plan(n_points, G).bytes));
    const BinSrc rows = rows_of(dxyz);
    PCFIT_TRY(count_launch(rows, n_points, dglo.as<double>(), dghi.as<double>(), G, geom1, dgoff.as<int64_t>(), dws1.p,
                           dws1.bytes, st));
    std::vector<int64_t> goff((size_t)G + 1);
    PCFIT_TRY(dgoff.download(goff.data(), goff.size() * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    const int64_t n1 = goff[(size_t)G];
    PCFIT_TRY(x1.alloc((size_t)n1 * 8));
    PCFIT_TRY(y1.alloc((size_t)n1 * 8));
    PCFIT_TRY(z1.alloc((size_t)n1 * 8));
    if (want_perm) PCFIT_TRY(i1.alloc((size_t)n1 * 8));
    if (n1 > 0)
        PCFIT_TRY(scatter_launch(rows, n_points, dglo.as<double>(), dghi.as<double>(), G, geom1, dgoff.as<int64_t>(),
                                 x1.as<double>(), y1.as<double>(), want_perm ? i1.as<int64_t>() : nullptr,
                                 z1.as<double>(), nullptr, dws1.p, dws1.bytes, st));
    // level 2, counts: every group against its own windows (one workspace and one offsets array, a part each)
    std::vector<size_t> ws_at((size_t)G + 1, 0);
    for (int64_t g = 0; g < G; g++) {
        const int64_t Sg = std::min<int64_t>(W, S - g * W), ng = goff[(size_t)g + 1] - goff[(size_t)g];
        ws_at[(size_t)g + 1] = ws_at[(size_t)g] + (ng > 0 ? bin_plan(ng, Sg).bytes : 0);
    }
    DevBuf ws2, off2;
    PCFIT_TRY(ws2.alloc(std::max<size_t>(ws_at[(size_t)G], 8)));
    PCFIT_TRY(off2.alloc((size_t)(S + G) * 8));                      // group g's Sg + 1 offsets start at g * W + g
    PCFIT_HIP(hipMemsetAsync(off2.p, 0, (size_t)(S + G) * 8, st));   // (groups without points are not counted)
    std::vector<double> geom2((size_t)G * 3);
    for (int64_t g = 0; g < G; g++) {
        const int64_t s0 = g * W, Sg = std::min<int64_t>(W, S - s0), ng = goff[(size_t)g + 1] - goff[(size_t)g];
        double *gm = &geom2[(size_t)g * 3];
        gm[0] = z_lo[s0];
        gm[1] = z_hi[s0];
        gm[2] = z_lo[s0 + Sg - 1];
        if (ng == 0)
            continue;
        const BinSrc cols{x1.as<double>() + goff[(size_t)g], y1.as<double>() + goff[(size_t)g],
                          z1.as<double>() + goff[(size_t)g], 1};
        PCFIT_TRY(count_launch(cols, ng, dlo.as<double>() + s0, dhi.as<double>() + s0, Sg, gm,
                               off2.as<int64_t>() + s0 + g, static_cast<char *>(ws2.p) + ws_at[(size_t)g],
                               ws_at[(size_t)g + 1] - ws_at[(size_t)g], st));
    }
    std::vector<int64_t> hoff((size_t)(S + G));
    PCFIT_TRY(off2.download(hoff.data(), hoff.size() * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> base((size_t)G + 1, 0);
    for (int64_t g = 0; g < G; g++) {
        const int64_t s0 = g * W, Sg = std::min<int64_t>(W, S - s0);
        const int64_t *og = &hoff[(size_t)(s0 + g)];
        for (int64_t j = 0; j < Sg; j++)
            offsets[s0 + j] = base[(size_t)g] + og[j];
        base[(size_t)g + 1] = base[(size_t)g] + og[Sg];
    }
    const int64_t total = base[(size_t)G];
    offsets[S] = total;
    if (capacity >= 0 && total > capacity)
        return PCFIT_OK;                       // (the caller reports it; nothing is scattered)
    PCFIT_TRY(doff.upload(offsets, (size_t)(S + 1) * 8, st));
    PCFIT_TRY(dx.alloc((size_t)total * 8));
    PCFIT_TRY(dy.alloc((size_t)total * 8));
    if (want_perm) PCFIT_TRY(dperm.alloc((size_t)total * 8));
    // level 2, scatter: group g writes at base[g] with its own offsets
    for (int64_t g = 0; g < G; g++) {
        const int64_t s0 = g * W, Sg = std::min<int64_t>(W, S - s0), ng = goff[(size_t)g + 1] - goff[(size_t)g];
        if (ng == 0 || base[(size_t)g + 1] == base[(size_t)g])
            continue;
        const BinSrc cols{x1.as<double>() + goff[(size_t)g], y1.as<double>() + goff[(size_t)g],
                          z1.as<double>() + goff[(size_t)g], 1};
        PCFIT_TRY(scatter_launch(cols, ng, dlo.as<double>() + s0, dhi.as<double>() + s0, Sg, &geom2[(size_t)g * 3],
                                 off2.as<int64_t>() + s0 + g, dx.as<double>() + base[(size_t)g],
                                 dy.as<double>() + base[(size_t)g],
                                 want_perm ? dperm.as<int64_t>() + base[(size_t)g] : nullptr, nullptr,
                                 want_perm ? i1.as<int64_t>() + goff[(size_t)g] : nullptr,
                                 static_cast<char *>(ws2.p) + ws_at[(size_t)g],
                                 ws_at[(size_t)g + 1] - ws_at[(size_t)g], st));
    }
    // (the level-1 arrays are released when this returns: the stream runs the scatters first, as every other
    // block handed back to the cache relies on)
    return PCFIT_OK;
}

// More windows than one count / scatter pass takes (a long scan cut very finely): the windows go in runs of
// BIN_MAX_SLICES consecutive ones, every run a count and a scatter pass of its own over the scan (a point outside
// a run's z range finds no window there), the runs' outputs one after the other.  Same members, order and bytes
// as one pass would give.
static int bin_in_runs(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t S,
                       bool want_perm, int64_t capacity, int64_t *offsets, DevBuf &doff, DevBuf &dx, DevBuf &dy,
                       DevBuf &dperm, const DevBuf &dlo, const DevBuf &dhi, hipStream_t st)
{
    const int64_t W = BIN_MAX_SLICES, G = (S + W - 1) / W;
    const BinSrc rows = rows_of(dxyz);
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
                  DevBuf &dperm, hipStream_t st)
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
                           st);
    // (read at every call so that a test can switch between the two forms; negative = never)
    const char *env = getenv("PCFIT_BIN_TWO_LEVEL_MIN");
    const int64_t two_level_min = env ? atoll(env) : BIN_TWO_LEVEL_MIN;
    if (S > BIN_STG_MAX_SLICES && two_level_min >= 0 && n_points >= two_level_min)
        return bin_two_level(dxyz, n_points, z_lo, z_hi, S, want_perm, capacity, offsets, doff, dx, dy, dperm, dlo,
                             dhi, st);
    const BinSrc rows = rows_of(dxyz);
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

// bin on the device, slices to the host: the scan is on the device already
static int bin_resident(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t n_slices,
                        int64_t *offsets, double *x_out, double *y_out, int64_t *perm, int64_t capacity,
                        int64_t *total)
{
    PCFIT_REQUIRE(total, "bin: null pointer");
    hipStream_t st = nullptr;
    DevBuf doff, dx, dy, dperm;
    const int64_t room = (x_out == nullptr || y_out == nullptr) ? 0 : std::max<int64_t>(capacity, 0);
    PCFIT_TRY(bin_on_device(dxyz, n_points, z_lo, z_hi, n_slices, perm != nullptr, room, offsets, doff, dx, dy, dperm,
                            st));
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
                        capacity, total);
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
        PCFIT_TRY(pcfit_bin_count_dev(scan->xyz.as<double>(), scan->n_points, dlo.as<double>() + s0,
                                      dhi.as<double>() + s0, Sg, geom, doff.as<int64_t>(), dws.p, dws.bytes, st));
        off.resize((size_t)Sg + 1);
        PCFIT_TRY(doff.download(off.data(), off.size() * 8, st));
        PCFIT_HIP(hipStreamSynchronize(st));
        for (int64_t s = 0; s < Sg; s++)
            counts[s0 + s] = off[(size_t)s + 1] - off[(size_t)s];
    }
    return PCFIT_OK;
}
