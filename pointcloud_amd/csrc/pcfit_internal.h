// Internal helpers shared by the kernel files of libpcfit_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <type_traits>

#include "pcfit.h"

#define PCFIT_API extern "C" __attribute__((visibility("default")))

namespace pcfit {

void set_error(const char *fmt, ...);
int bin_check_windows(const double *z_lo, const double *z_hi, int64_t n_slices, double geom[3]);   // bin.hip
// hull.hip: pcfit_unique_hull_batch_dev with the option to drop interior points of medium slices too
int unique_hull_dev(const double *x, const double *y, const int64_t *offsets, int64_t n_slices, int64_t max_slice_len,
                    double *ux, double *uy, int64_t *n_unique, int64_t *hull_idx, int64_t *n_hull, int32_t *status,
                    const int64_t *hull_start, void *workspace, size_t workspace_bytes, void *stream, int hull_only);
size_t unique_hull_workspace_bytes(int64_t total_points, int64_t n_slices, int hull_only);
// qhull_order.hip: the vertex Qhull's facet list starts with, per slice of sorted distinct points (host cores; only
// where select is null or select[s] == PCFIT_HULL_START_QHULL, -1 elsewhere), and the turn of device walks
// to the vertices with given coordinates and their reduction to Qhull's vertex set (where found[s] >= 0; n_hull is
// updated; tmp: as many entries as hull_idx)
int qhull_start_host(const double *ux, const double *uy, const int64_t *offsets, const int64_t *n_unique,
                     int64_t n_slices, int32_t n_threads, const int64_t *select, int64_t *start, uint8_t *sure,
                     int64_t *walk, int64_t *n_walk);
int hull_rotate_dev(const int64_t *offsets, int64_t *n_hull, const double *ux, const double *uy,
                    const int64_t *found, const double *start_xy, int64_t n_slices, int64_t *hull_idx, int64_t *tmp,
                    hipStream_t st, const void *verts = nullptr, const int32_t *vcount = nullptr, int vstride = 0);
// qhull_dev.hip: the same replay on the device, one workgroup per slice, of slices that are sorted and distinct
// (n_unique points each) or as they come (n_unique null); start[s] = QHULL_START_DEFER where the host has to
// decide (anything but the plain course of the construction); start_xy (nullable): the vertex's coordinates.
// scratch: 3 * total (x, y) pairs, 48 bytes a point (the outside sets are kept as points, not as indices: every pass
// over them streams).  want_index = false: start[s] = 0 wherever there is an answer, which saves the pass that
// finds the vertex's place in its slice.  max_hull: the largest n_hull of the batch when the host knows it (the kernel
// then runs in the form with the fewest facet slots that suffices), else -1.  hull_x, hull_y, hull_idx (nullable,
// all three or none): the slices' strictly convex hulls as unique_hull_dev left them - vertex i of slice s is
// (hull_x, hull_y)[offsets[s] + hull_idx[offsets[s] + i]] - among whose vertices the kernel then finds the
// coordinate extremes instead of by a pass over the slice.
// verts, vcount, vstride (nullable): the kernel also lists the vertices Qhull ends with, in Qhull's order, as (x, y)
// pairs at verts[s * vstride ...], vcount[s] of them (vcount is left alone where there is no answer; a slice with
// more than vstride vertices is handed back).
constexpr int64_t QHULL_START_DEFER = -3;
int qhull_start_dev(const double *ux, const double *uy, const int64_t *offsets, const int64_t *n_unique,
                    const int64_t *n_hull, const int64_t *select, int64_t n_slices, int64_t total, void *scratch,
                    int64_t *start, double *start_xy, hipStream_t st, int64_t max_hull, const double *hull_x,
                    const double *hull_y, const int64_t *hull_idx, bool want_index, void *verts = nullptr,
                    int32_t *vcount = nullptr, int vstride = 0);

// longsort.hip: lexicographic sort of (a, b) pairs inside every slice longer than min_len, and the distinct
// pairs of the sorted slices (see there)
size_t long_sort_workspace_bytes(int64_t total, int64_t n_slices);
int long_sort_dev(const double *in_a, const double *in_b, double *a0, double *b0, double *a1, double *b1,
                  const int64_t *offsets, int64_t n_slices, int64_t total, int64_t max_len, int64_t min_len,
                  void *small_ws, int *in_second, hipStream_t st);
int long_unique_dev(const double *a, const double *b, const int64_t *offsets, int64_t n_slices, int64_t total,
                    int64_t max_len, int64_t min_len, void *small_ws, double *ua, double *ub, int64_t *n_unique,
                    hipStream_t st);

#define PCFIT_HIP(call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            pcfit::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PCFIT_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)

#define PCFIT_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            pcfit::set_error(__VA_ARGS__);       \
            return PCFIT_ERR_INVALID;            \
        }                                        \
    } while (0)

// kernel launches are asynchronous: this catches bad launch configurations only
#define PCFIT_LAUNCH_CHECK() PCFIT_HIP(hipGetLastError())
// First statement of every entry point that touches the device: HIP keeps the last error of the thread
// until it is read, so a failed call of the application's own would otherwise fail the next launch check.
#define PCFIT_ENTER() (void)hipGetLastError()

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Device memory of the host-pointer entry points comes from a per-device cache of blocks (api.hip):
// fresh device memory costs far more to map and touch for the first time than the kernels that use it
// run, so blocks are kept between calls (bounded, see pcfit_release_cache).  A block comes back with
// whatever the last user left in it.
int pool_get(size_t bytes, void **ptr, size_t *got);
void pool_put(void *ptr, size_t bytes);
int pool_release();
// Large host-to-device copies from pageable memory: several host threads copy chunks into pinned staging
// buffers, each chunk followed by its own asynchronous transfer (api.hip).  Returns when the data is in place.
int upload_staged(void *dst_dev, const void *src_host, size_t bytes);
int download_staged(void *dst_host, const void *src_dev, size_t bytes);      // the same, the other way
// Pinned host memory of the calling thread, at least `bytes` long (grown on demand, kept for the thread's life; null
// when it cannot be had): small results copied into it arrive at the pinned rate and touch no fresh pages.  What
// it held is gone after the next call from the same thread; the caller synchronises before it lets go of it.
void *pinned_scratch(size_t bytes);
constexpr size_t STAGED_UPLOAD_MIN = (size_t)32 << 20;

// Device buffer that returns itself: the host-pointer entry points own all the
// device memory they touch for the duration of one call.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) pool_put(p, bytes); }
    int alloc(size_t n)
    {
        if (p && n <= bytes) return PCFIT_OK;          // reuse
        if (p) { pool_put(p, bytes); p = nullptr; bytes = 0; }
        return pool_get(n, &p, &bytes);
    }
    int upload(const void *src, size_t n, hipStream_t st)
    {
        int rc = alloc(n);
        if (rc) return rc;
        if (n >= STAGED_UPLOAD_MIN && st == nullptr)
            return upload_staged(p, src, n);
        if (n) PCFIT_HIP(hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, st));
        return PCFIT_OK;
    }
    int download(void *dst, size_t n, hipStream_t st) const
    {
        if (n >= STAGED_UPLOAD_MIN && dst && st == nullptr)
            return download_staged(dst, p, n);
        if (n && dst) PCFIT_HIP(hipMemcpyAsync(dst, p, n, hipMemcpyDeviceToHost, st));
        return PCFIT_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// bin.hip: the scan (rows of xyz on the device) binned into the windows - offsets to the host and to doff,
// x, y (and perm) left on the device; in two levels when the windows are many.  With capacity >= 0 and more
// members than that only the offsets are produced.
// dz (nullable): the scan's z column by itself; the count passes then read it instead of the rows.
int bin_on_device(const double *dxyz, int64_t n_points, const double *z_lo, const double *z_hi, int64_t n_slices,
                  bool want_perm, int64_t capacity, int64_t *offsets, DevBuf &doff, DevBuf &dx, DevBuf &dy,
                  DevBuf &dperm, hipStream_t st, const double *dz = nullptr);
// the z column of a scan of rows (xyz -> z), enqueued; scan_z_column: for a scan just uploaded (an empty column and
// PCFIT_OK when it is switched off by PCFIT_SCAN_Z_COLUMN=0 or memory is short)
int scan_z_column(pcfit_scan &scan, hipStream_t st);

}  // namespace pcfit

// A scan kept on the device between calls (the opaque pcfit_scan of the header).
struct pcfit_scan {
    pcfit::DevBuf xyz;
    pcfit::DevBuf z;               // the z column by itself (8 bytes a point; empty when it was not made): what the
                                   // count passes of the binning read instead of the rows
    int64_t n_points = 0;
    int device = 0;
};

namespace pcfit {

#define PCFIT_TRY(expr)          \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != PCFIT_OK)     \
            return rc_;          \
    } while (0)

inline int check_offsets(const int64_t *offsets, int64_t S, int64_t *total, int64_t *max_len)
{
    PCFIT_REQUIRE(S >= 0, "n_slices negative");
    PCFIT_REQUIRE(offsets != nullptr, "offsets is null");
    int64_t mx = 0;
    for (int64_t s = 0; s < S; s++) {
        PCFIT_REQUIRE(offsets[s + 1] >= offsets[s], "offsets must be non-decreasing (slice %lld)", (long long)s);
        mx = std::max(mx, offsets[s + 1] - offsets[s]);
    }
    PCFIT_REQUIRE(offsets[0] >= 0, "offsets[0] negative");
    *total = offsets[S];
    *max_len = mx;
    return PCFIT_OK;
}


constexpr int WAVE = 64;
constexpr double TWO_PI = 6.283185307179586476925286766559;

// ---------------------------------------------------------------- device side
// Butterfly sum over the 64 lanes of a wave: every lane ends with the same
// value, the order of additions is fixed (bitwise reproducible).
__device__ __forceinline__ double wave_sum(double v);       // (below, once the lane exchanges are declared)

// Sum K values per thread over a whole workgroup (blockDim.x a multiple of 64,
// at most 1024).  All threads receive the K totals.  scratch: K * 16 doubles
// of LDS.  Fixed order: lanes by butterfly, then waves 0,1,2,...
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++)
        v[k] = wave_sum(v[k]);
    if (nw == 1)
        return;
    __syncthreads();                       // scratch may still be read from a previous call
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++)
            scratch[k * 16 + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = scratch[k * 16];
        for (int w = 1; w < nw; w++)
            s += scratch[k * 16 + w];
        v[k] = s;
    }
}

// Same sum, but the K totals are left in LDS (dst, K doubles) instead of in
// every thread's registers; callers read what they need after the call.
template <int K>
__device__ __forceinline__ void block_sum_to_lds(double (&v)[K], double *scratch, double *dst)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++)
        v[k] = wave_sum(v[k]);
    __syncthreads();                       // scratch / dst may still be read from a previous call
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++)
            scratch[k * 16 + wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double s = scratch[threadIdx.x * 16];
        for (int w = 1; w < nw; w++)
            s += scratch[threadIdx.x * 16 + w];
        dst[threadIdx.x] = s;
    }
    __syncthreads();
}

// ---- lane exchanges without the LDS crossbar
// __shfl_xor travels through ds_bpermute (two of them per double, some hundred cycles each, one after the other
// along a butterfly).  Inside a row of 16 lanes the data-parallel-primitive moves do the same in a few cycles,
// and gfx950 swaps rows of 16 and halves of 32 lanes between two registers in one instruction.
//   dpp_mov<CTRL, BANKS>(old, src): lanes of the banks (groups of four lanes of a row) in BANKS receive src from the
//   lane CTRL names, the others keep old.
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_SAME = 0xE4, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHR4 = 0x114,
              DPP_ROW_ROR8 = 0x128;

template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_mov(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}

// Rows of 16 (HALF = 16) or halves of 32 lanes (HALF = 32): the lanes whose HALF bit is set receive b of their
// partner (lane ^ HALF) in a, the others receive a of their partner in b; what is not exchanged stays.
template <int HALF>
__device__ __forceinline__ void swap_halves(double &a, double &b)
{
    int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    if constexpr (HALF == 32) {
        const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
    }
    a = __hiloint2double(ahi, alo);
    b = __hiloint2double(bhi, blo);
}

// One step of a reduce-scatter over the lane bit M: lanes with the bit clear end with x (own) + x (partner's),
// lanes with the bit set with y (own) + y (partner's).  Same sums, same order of operands, as
//     keep = up ? y : x;  send = up ? x : y;  keep + __shfl_xor(send, M).
template <int M>
__device__ __forceinline__ double rs_pair(double x, double y, int lane)
{
    if constexpr (M >= 16) {
        swap_halves<M>(x, y);           // clear: x own, y = partner's x; set: y own, x = partner's y
        return x + y;
    } else if constexpr (M == 8) {
        const double keep = dpp_mov<DPP_SAME, 0x3>(y, x);                 // banks 0, 1 = lanes with the bit clear
        double other = dpp_mov<DPP_ROW_ROR8, 0xC>(y, y);
        other = dpp_mov<DPP_ROW_ROR8, 0x3>(other, x);
        return keep + other;
    } else if constexpr (M == 4) {
        const double keep = dpp_mov<DPP_SAME, 0x5>(y, x);                 // banks 0, 2
        double other = dpp_mov<DPP_ROW_SHL4, 0x5>(y, x);                  // from lane + 4
        other = dpp_mov<DPP_ROW_SHR4, 0xA>(other, y);                     // from lane - 4
        return keep + other;
    } else {
        const bool up = (lane & M) != 0;
        const double keep = up ? y : x, send = up ? x : y;
        return keep + dpp_mov<M == 2 ? DPP_QUAD_XOR2 : DPP_QUAD_XOR1, 0xf>(send, send);
    }
}

// the value of lane ^ M (M a power of two below 64)
template <int M>
__device__ __forceinline__ double lane_xor(double v, int lane)
{
    if constexpr (M >= 16) {
        double a = v, b = v;
        swap_halves<M>(a, b);           // clear: b = partner's; set: a = partner's
        return (lane & M) ? a : b;
    } else if constexpr (M == 8) {
        return dpp_mov<DPP_ROW_ROR8, 0xf>(v, v);
    } else if constexpr (M == 4) {
        const double t = dpp_mov<DPP_ROW_SHL4, 0x5>(v, v);
        return dpp_mov<DPP_ROW_SHR4, 0xA>(t, v);
    } else {
        return dpp_mov<M == 2 ? DPP_QUAD_XOR2 : DPP_QUAD_XOR1, 0xf>(v, v);
    }
}

template <int CTRL, int BANKS>
__device__ __forceinline__ int dpp_mov_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, BANKS, false);
}

template <int M>
__device__ __forceinline__ int lane_xor_i32(int v, int lane)
{
    if constexpr (M == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & M) ? r[0] : r[1];
    } else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & M) ? r[0] : r[1];
    } else if constexpr (M == 8) {
        return dpp_mov_i32<DPP_ROW_ROR8, 0xf>(v, v);
    } else if constexpr (M == 4) {
        const int t = dpp_mov_i32<DPP_ROW_SHL4, 0x5>(v, v);
        return dpp_mov_i32<DPP_ROW_SHR4, 0xA>(t, v);
    } else {
        return dpp_mov_i32<M == 2 ? DPP_QUAD_XOR2 : DPP_QUAD_XOR1, 0xf>(v, v);
    }
}

// v + (value of lane ^ M), without the select of lane_xor for the wide steps
template <int M>
__device__ __forceinline__ double add_lane_xor(double v, int lane)
{
    if constexpr (M >= 16) {
        double a = v, b = v;
        swap_halves<M>(a, b);
        return a + b;                   // own + partner's on either side
    } else {
        return v + lane_xor<M>(v, lane);
    }
}

// Reduce-scatter butterfly over a wave: CNT values per lane go in; each exchange
// halves the number of values a lane carries (a lane keeps one half and sends the
// other to its partner), so summing K values over 64 lanes moves about K + 5
// doubles per lane instead of 6 K.  Once one value is left the remaining
// exchanges are plain butterflies.  At the end v[0] of lane l is the total of the
// value with index rs_index<K, 32>(l) (lanes that differ only in the bits of
// rs_lowmask<K, 32>() hold the same total).  The order of the additions is fixed.
template <int CNT, int M, int K>
__device__ __forceinline__ void rs_steps(double (&v)[K], int lane)
{
    if constexpr (M >= 1) {
        if constexpr (CNT > 1) {
            constexpr int H = (CNT + 1) / 2;
#pragma unroll
            for (int j = 0; j < H; j++) {
                const double hi = (j + H < CNT) ? v[j + H] : 0.0;
                v[j] = rs_pair<M>(v[j], hi, lane);
            }
            rs_steps<H, M / 2>(v, lane);
        } else {
            v[0] = add_lane_xor<M>(v[0], lane);
            rs_steps<1, M / 2>(v, lane);
        }
    }
}

template <int CNT, int M>
__device__ __forceinline__ int rs_index(int lane, bool &valid)
{
    if constexpr (M >= 1 && CNT > 1) {
        constexpr int H = (CNT + 1) / 2;
        const int loc = rs_index<H, M / 2>(lane, valid) + ((lane & M) ? H : 0);
        if (loc >= CNT)
            valid = false;              // a padding slot of an odd split
        return loc;
    } else {
        return 0;
    }
}

template <int CNT, int M>
constexpr int rs_lowmask()
{
    if constexpr (M >= 1 && CNT > 1)
        return rs_lowmask<(CNT + 1) / 2, M / 2>();
    else
        return M >= 1 ? 2 * M - 1 : 0;
}

// the smaller / larger of two numbers, neither a signalling NaN: one instruction (fmin() and fmax() first
// quiet their operands); a quiet NaN operand yields the other one
__device__ __forceinline__ double min_no_nan(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double max_no_nan(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max(a, |b|), same remarks
__device__ __forceinline__ double max_abs_no_nan(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int M, class OP>
__device__ __forceinline__ double wave_fold(double v, int lane, OP op)
{
    if constexpr (M >= 1) {
        if constexpr (M >= 16) {
            double a = v, b = v;
            swap_halves<M>(a, b);
            v = op(a, b);               // own and partner's, in one order or the other
        } else {
            v = op(v, lane_xor<M>(v, lane));
        }
        return wave_fold<M / 2>(v, lane, op);
    } else {
        return v;
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
    return wave_fold<32>(v, (int)(threadIdx.x & 63), [](double a, double b) { return a + b; });
}

__device__ __forceinline__ double wave_min(double v)
{
    return wave_fold<32>(v, (int)(threadIdx.x & 63), [](double a, double b) { return fmin(a, b); });
}

// Sum K values per thread over a workgroup of NW waves and, when WITH_MIN, take
// the smallest `low` as well; the K totals (and the minimum, at dst[K]) are left
// in LDS for every thread to read after the call.  scratch: (K + 1) * 16 doubles.
// Order: reduce-scatter butterfly inside a wave, then waves 0, 1, 2, ...
// Two barriers; the caller alternates between two scratch areas (and makes sure
// nobody still reads dst), so that none is needed on entry.  Without END_BARRIER the
// second one is left out: the totals are written by lanes of the first wave, which
// may read them at once (LDS operations of one wave complete in order); any other
// wave has to pass a barrier of the caller's first.
// `owner`: the wave whose lanes form the totals (and may read them at once when END_BARRIER is left out).
template <int K, bool WITH_MIN, int NW, bool END_BARRIER = true>
__device__ __forceinline__ void block_reduce_to_lds(double (&v)[K], double low, double *scratch, double *dst,
                                                    int owner = 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tid = (int)threadIdx.x - owner * 64;          // lanes 0, 1, ... of the owning wave
    rs_steps<K, 32>(v, lane);
    bool valid = true;
    const int k = rs_index<K, 32>(lane, valid);
    if constexpr (WITH_MIN)
        low = wave_min(low);
    if (valid && (lane & rs_lowmask<K, 32>()) == 0)
        scratch[k * 16 + wave] = v[0];
    if (WITH_MIN && lane == 0)
        scratch[K * 16 + wave] = low;
    __syncthreads();
    if (tid >= 0 && tid < K) {
        double part[NW];
#pragma unroll
        for (int w = 0; w < NW; w++)
            part[w] = scratch[tid * 16 + w];
        double s = part[0];
#pragma unroll
        for (int w = 1; w < NW; w++)
            s += part[w];
        dst[tid] = s;
    } else if (WITH_MIN && tid == K) {
        double s = scratch[K * 16];
#pragma unroll
        for (int w = 1; w < NW; w++)
            s = fmin(s, scratch[K * 16 + w]);
        dst[K] = s;
    }
    if constexpr (END_BARRIER)
        __syncthreads();
}

__device__ __forceinline__ double wave_max(double v)
{
    return wave_fold<32>(v, (int)(threadIdx.x & 63), [](double a, double b) { return fmax(a, b); });
}

// min over the four lanes of a quad (every lane of the quad ends with it): two quad permutes, no copy first
// (every lane receives a value, so what the destination held before does not matter)
__device__ __forceinline__ double quad_min(double v)
{
    auto from = [](double x, auto ctrl) {
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), decltype(ctrl)::value, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), decltype(ctrl)::value, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    };
    v = min_no_nan(v, from(v, std::integral_constant<int, DPP_QUAD_XOR1>{}));
    return min_no_nan(v, from(v, std::integral_constant<int, DPP_QUAD_XOR2>{}));
}

// The same with two extremes: the smallest `low` at dst[AT] and the largest `high` at dst[AT + 1]; the sums
// from the AT-th on follow them (sum k at dst[k + 2]).  scratch: K * 16 doubles; lohi: 2 * NW * 16 doubles
// (one area, not two in turn: between the owner's reads and the next call's writes lies the barrier
// through which the owner hands its step to the others).
// The two extremes take the cheap road: a fold over the lanes of every wave would cost some 80 instructions
// on each of them (two values, six exchanges of two moves each, the operation); here a quad folds by two
// permutes, one lane in four leaves the quad's pair in LDS, and after the barrier the owning wave alone
// folds them - its lower half the smallest distances, its upper half the largest, negated, so that one chain
// of minima serves both.
template <int K, int NW, int AT = K, bool END_BARRIER = true>
__device__ __forceinline__ void block_reduce_to_lds2(double (&v)[K], double low, double high, double *scratch,
                                                     double *lohi, double *dst, int owner = 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tid = (int)threadIdx.x - owner * 64;
    rs_steps<K, 32>(v, lane);
    bool valid = true;
    const int k = rs_index<K, 32>(lane, valid);
    low = quad_min(low);
    high = quad_min(-high);
    if (valid && (lane & rs_lowmask<K, 32>()) == 0)
        scratch[k * 16 + wave] = v[0];
    if ((lane & 3) == 0) {
        lohi[wave * 16 + (lane >> 2)] = low;
        lohi[NW * 16 + wave * 16 + (lane >> 2)] = high;
    }
    __syncthreads();
    if (wave == owner) {                        // (uniform over the wave: all its lanes take part in the exchanges)
        // NW * 16 values of either kind: lane l of a half takes those at l, l + 32, ...
        constexpr int PER = NW * 16, READS = (PER + 31) / 32;
        const double *src = lohi + (lane >> 5) * PER;
        const int l5 = lane & 31;
        double m = src[l5 < PER ? l5 : PER - 1];
#pragma unroll
        for (int j = 1; j < READS; j++)
            m = min_no_nan(m, src[l5 + 32 * j < PER ? l5 + 32 * j : PER - 1]);
        m = wave_fold<16>(m, lane, [](double a, double b) { return min_no_nan(a, b); });
        if (lane == 0)
            dst[AT] = m;
        if (lane == 32)
            dst[AT + 1] = -m;
    }
    if (tid >= 0 && tid < K) {
        double part[NW];
#pragma unroll
        for (int w = 0; w < NW; w++)
            part[w] = scratch[tid * 16 + w];
        double s = part[0];
#pragma unroll
        for (int w = 1; w < NW; w++)
            s += part[w];
        dst[tid < AT ? tid : tid + 2] = s;
    }
    if constexpr (END_BARRIER)
        __syncthreads();
}

// (theta, r) as the reference forms them: theta = (atan2(dy,dx) + 2pi) mod 2pi
__device__ __forceinline__ double wrap_angle(double dy, double dx)
{
    return fmod(atan2(dy, dx) + TWO_PI, TWO_PI);
}

// 1/sqrt(d) to about one ulp (d > 0, normal range): the hardware estimate is good to 2^-24 (measured,
// tools/ubench/rsq_acc.hip), so one third-order step, y' = y + y e (1/2 + 3/8 e) with e = 1 - d y^2, leaves
// an error of 1.4e-16 at most - below that of two Newton steps (2.4e-16) at five operations instead of seven.
__device__ __forceinline__ double rsqrt_nr(double d)
{
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// hypot(x, y) rounded to nearest.  The sum of squares is formed exactly as a
// double-double (s + tail); r0 ~ sqrt(s) comes from the reciprocal square root
// and one Newton step with the residual s + tail - r0^2 taken exactly lands on
// the correctly rounded value (the step's own error is ~1e-32 relative, so only
// a result within that distance of a rounding tie could differ).  No range
// scaling: meant for |x|, |y| within [1e-140, 1e140].
__device__ __forceinline__ double cr_hypot(double x, double y)
{
    const double p1 = x * x, e1 = fma(x, x, -p1);
    const double p2 = y * y, e2 = fma(y, y, -p2);
    const double s = p1 + p2;
    const double bv = s - p1;
    const double es = (p1 - (s - bv)) + (p2 - bv);
    const double tail = es + (e1 + e2);
    const double inv = rsqrt_nr(s);
    const double r = s * inv;
    const double rr = r * r, er = fma(r, r, -rr);
    const double resid = ((s - rr) - er) + tail;
    const double h = fma(resid, 0.5 * inv, r);
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    return (s > 0.0 && s < inf) ? h : s;        // 0, inf or NaN pass through (a select, not a branch)
}

__device__ __forceinline__ double next_up(double v) { return __longlong_as_double(__double_as_longlong(v) + 1); }
__device__ __forceinline__ double next_down(double v) { return __longlong_as_double(__double_as_longlong(v) - 1); }

// 2x2 solve exactly as LAPACK dgesv evaluates it on the build host (partial
// pivoting, reciprocal-scaled multiplier, fused multiply-adds in the two
// triangular solves).  Rows (a b | c), (d e | f); returns the pivot product.
__device__ __forceinline__ double solve2x2(double a, double b, double c, double d, double e, double f, double &x1,
                                           double &x2)
{
    if (fabs(d) > fabs(a)) {
        double t;
        t = a; a = d; d = t;
        t = b; b = e; e = t;
        t = c; c = f; f = t;
    }
    const double l = d * (1.0 / a);
    const double u22 = e - l * b;
    const double y2 = fma(-l, c, f);
    x2 = y2 / u22;
    x1 = fma(-b, x2, c) / a;
    return a * u22;
}

}  // namespace pcfit
