// Library plumbing, RANSAC sampling tables (host, integer work) and the
// host-pointer convenience entry points of libpcfit_hip.so.
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "pcfit_internal.h"

namespace pcfit {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---- numpy SeedSequence -> PCG64, Generator.choice(n, 3, replace=False) ----
typedef unsigned __int128 u128;
static const u128 PCG_MULT = ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;

struct Pcg {
    u128 state, inc;
    bool has32 = false;
    uint32_t hold = 0;
    uint64_t next64()
    {
        state = state * PCG_MULT + inc;
        const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
        const uint64_t v = hi ^ lo;
        const unsigned rot = (unsigned)(hi >> 58);
        return (v >> rot) | (v << ((64 - rot) & 63));
    }
    uint32_t next32()
    {
        if (has32) {
            has32 = false;
            return hold;
        }
        const uint64_t r = next64();
        has32 = true;
        hold = (uint32_t)(r >> 32);
        return (uint32_t)r;
    }
    // uniform on [0, top], Lemire's method on 32-bit draws; top == 0 draws nothing
    uint32_t bounded(uint32_t top)
    {
        if (top == 0) return 0;
        if (top == 0xFFFFFFFFu) return next32();
        const uint32_t span = top + 1;
        uint64_t m = (uint64_t)next32() * span;
        uint32_t low = (uint32_t)m;
        if (low < span) {
            const uint32_t thr = (uint32_t)((0xFFFFFFFFull - top) % span);
            while (low < thr) {
                m = (uint64_t)next32() * span;
                low = (uint32_t)m;
            }
        }
        return (uint32_t)(m >> 32);
    }
};

static void choice3_table(const uint64_t st[4], int64_t n, int32_t trials, int32_t *out)
{
    Pcg g;
    g.state = ((u128)st[0] << 64) | st[1];
    g.inc = ((u128)st[2] << 64) | st[3];
    for (int32_t t = 0; t < trials; t++, out += 3) {
        // Floyd's sampling of 3 out of n through a 4-slot open-addressed set ...
        int64_t slot[4] = {-1, -1, -1, -1};
        for (int k = 0; k < 3; k++) {
            const int64_t j = n - 3 + k;
            const int64_t v = g.bounded((uint32_t)j);
            int at = (int)(v & 3);
            while (slot[at] != -1 && slot[at] != v)
                at = (at + 1) & 3;
            if (slot[at] == -1) {
                slot[at] = v;
                out[k] = (int32_t)v;
            } else {
                at = (int)(j & 3);
                while (slot[at] != -1)
                    at = (at + 1) & 3;
                slot[at] = j;
                out[k] = (int32_t)j;
            }
        }
        // ... followed by a shuffle of the three
        for (int i = 2; i >= 1; i--)
            std::swap(out[i], out[g.bounded((uint32_t)i)]);
    }
}


// ---- cache of device blocks for the host-pointer entry points ----
// All of them work on the null stream and synchronise it before they return, and work queued there
// by a later call runs after whatever an earlier call left in flight, so a block can be handed out again
// as soon as its DevBuf is gone.
namespace {
struct PoolBlock {
    void *p;
    size_t bytes;
};
constexpr int POOL_MAX_DEV = 64;
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool[POOL_MAX_DEV];
size_t g_pool_bytes[POOL_MAX_DEV];

size_t pool_cap()
{
    static const size_t cap = [] {
        const char *e = getenv("PCFIT_POOL_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)16 << 30);
    }();
    return cap;
}
bool pool_poison()
{
    static const bool on = [] { const char *e = getenv("PCFIT_POOL_POISON"); return e && *e == '1'; }();
    return on;
}
void pool_drop(int dev)
{
    for (const PoolBlock &b : g_pool[dev])
        (void)hipFree(b.p);
    g_pool[dev].clear();
    g_pool_bytes[dev] = 0;
}
}  // namespace

int pool_get(size_t bytes, void **ptr, size_t *got)
{
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    const size_t need = std::max<size_t>(256, (bytes + 255) & ~(size_t)255);
    if (dev >= 0 && dev < POOL_MAX_DEV) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        std::vector<PoolBlock> &v = g_pool[dev];
        int best = -1;
        for (int i = 0; i < (int)v.size(); i++)
            if (v[i].bytes >= need && v[i].bytes <= 2 * need + (1u << 20) && (best < 0 || v[i].bytes < v[best].bytes))
                best = i;
        if (best >= 0) {
            *ptr = v[best].p;
            *got = v[best].bytes;
            g_pool_bytes[dev] -= v[best].bytes;
            v.erase(v.begin() + best);
            if (pool_poison())
                PCFIT_HIP(hipMemsetAsync(*ptr, 0xff, *got, nullptr));
            return PCFIT_OK;
        }
    }
    hipError_t e = hipMalloc(ptr, need);
    if (e == hipErrorOutOfMemory && dev >= 0 && dev < POOL_MAX_DEV) {
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(g_pool_mu);
            pool_drop(dev);                      // give the cached blocks back and try once more
        }
        e = hipMalloc(ptr, need);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        set_error("hipMalloc of %zu bytes failed: %s", need, hipGetErrorString(e));
        return PCFIT_ERR_HIP;
    }
    *got = need;
    if (pool_poison())
        PCFIT_HIP(hipMemsetAsync(*ptr, 0xff, need, nullptr));
    return PCFIT_OK;
}

void pool_put(void *ptr, size_t bytes)
{
    if (!ptr)
        return;
    int dev = -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) == hipSuccess)
        dev = attr.device;
    else
        (void)hipGetLastError();
    if (dev >= 0 && dev < POOL_MAX_DEV) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (g_pool_bytes[dev] + bytes <= pool_cap()) {
            g_pool[dev].push_back({ptr, bytes});
            g_pool_bytes[dev] += bytes;
            return;
        }
    }
    (void)hipFree(ptr);
}

// ---- staged upload ----
void *pinned_scratch(size_t bytes)
{
    static thread_local void *buf = nullptr;
    static thread_local size_t cap = 0;
    static const bool off = [] { const char *e = getenv("PCFIT_PINNED_SCRATCH"); return e && e[0] == '0'; }();
    if (off) return nullptr;
    if (bytes <= cap) return buf;
    if (buf) {
        (void)hipHostFree(buf);
        buf = nullptr;
        cap = 0;
    }
    size_t want = (size_t)1 << 20;
    while (want < bytes) want <<= 1;
    if (hipHostMalloc(&buf, want, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        buf = nullptr;
        return nullptr;
    }
    cap = want;
    return buf;
}

// hipMemcpy from pageable memory reaches about 30 GB/s on this platform, from pinned memory 56 GB/s; four
// host threads that copy 8 MB chunks into pinned buffers and send each on a stream of their own reach
// 40-50 GB/s from pageable memory.  Streams, events and the 64 MB of pinned buffers are made once per process
// and device; one upload at a time uses them.
namespace {
constexpr int STAGE_T = 4;
constexpr size_t STAGE_CHUNK = (size_t)8 << 20;
struct Stager {
    int dev = -1;
    bool ok = false;
    hipStream_t st[STAGE_T] = {};
    hipEvent_t ev[2 * STAGE_T] = {};
    void *buf[2 * STAGE_T] = {};
};
std::mutex g_stage_mu;
Stager g_stage;

void stage_drop()
{
    if (!g_stage.ok)
        return;
    for (int t = 0; t < STAGE_T; t++)
        (void)hipStreamDestroy(g_stage.st[t]);
    for (int b = 0; b < 2 * STAGE_T; b++) {
        (void)hipEventDestroy(g_stage.ev[b]);
        (void)hipHostFree(g_stage.buf[b]);
    }
    g_stage = Stager{};
}

bool stage_make(int dev)
{
    if (g_stage.ok && g_stage.dev == dev)
        return true;
    stage_drop();
    Stager s;
    s.dev = dev;
    bool good = true;
    for (int t = 0; t < STAGE_T && good; t++)
        good = hipStreamCreateWithFlags(&s.st[t], hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 * STAGE_T && good; b++)
        good = hipEventCreateWithFlags(&s.ev[b], hipEventDisableTiming) == hipSuccess &&
               hipHostMalloc(&s.buf[b], STAGE_CHUNK, hipHostMallocDefault) == hipSuccess;
    if (!good) {
        (void)hipGetLastError();
        return false;                                   // (what was made is leaked only if creation failed half way)
    }
    s.ok = true;
    g_stage = s;
    return true;
}
}  // namespace

int upload_staged(void *dst, const void *src, size_t bytes)
{
    static const bool off = [] { const char *e = getenv("PCFIT_STAGED_UPLOAD"); return e && *e == '0'; }();
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    std::unique_lock<std::mutex> lock(g_stage_mu);
    if (off || !stage_make(dev)) {
        lock.unlock();
        PCFIT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return PCFIT_OK;
    }
    // the block may still be read by work queued on the null stream before it was handed out again
    PCFIT_HIP(hipStreamSynchronize(nullptr));
    const size_t chunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    int rc[STAGE_T] = {};
    auto work = [&](int t) {
        if (hipSetDevice(dev) != hipSuccess) { rc[t] = 1; return; }
        size_t use = 0;
        for (size_t c = t; c < chunks; c += STAGE_T, use++) {
            const int b = 2 * t + (int)(use & 1);
            if (use >= 2 && hipEventSynchronize(g_stage.ev[b]) != hipSuccess) { rc[t] = 1; return; }
            const size_t at = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - at);
            memcpy(g_stage.buf[b], static_cast<const char *>(src) + at, len);
            if (hipMemcpyAsync(static_cast<char *>(dst) + at, g_stage.buf[b], len, hipMemcpyHostToDevice,
                               g_stage.st[t]) != hipSuccess ||
                hipEventRecord(g_stage.ev[b], g_stage.st[t]) != hipSuccess) { rc[t] = 1; return; }
        }
        if (hipStreamSynchronize(g_stage.st[t]) != hipSuccess)
            rc[t] = 1;
    };
    std::thread pool[STAGE_T - 1];
    for (int t = 1; t < STAGE_T; t++)
        pool[t - 1] = std::thread(work, t);
    work(0);
    for (auto &th : pool)
        th.join();
    for (int t = 0; t < STAGE_T; t++)
        if (rc[t]) {
            (void)hipGetLastError();
            set_error("staged upload of %zu bytes failed", bytes);
            return PCFIT_ERR_HIP;
        }
    return PCFIT_OK;
}

int download_staged(void *dst, const void *src, size_t bytes)
{
    static const bool off = [] { const char *e = getenv("PCFIT_STAGED_UPLOAD"); return e && *e == '0'; }();
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    std::unique_lock<std::mutex> lock(g_stage_mu);
    if (off || !stage_make(dev)) {
        lock.unlock();
        PCFIT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return PCFIT_OK;
    }
    PCFIT_HIP(hipStreamSynchronize(nullptr));              // what the null stream still computes is what is fetched
    const size_t chunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    int rc[STAGE_T] = {};
    auto work = [&](int t) {
        if (hipSetDevice(dev) != hipSuccess) { rc[t] = 1; return; }
        // two buffers per thread: the transfer of the next chunk runs while this one is copied out
        auto fetch = [&](size_t c, int b) {
            const size_t at = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - at);
            return hipMemcpyAsync(g_stage.buf[b], static_cast<const char *>(src) + at, len, hipMemcpyDeviceToHost,
                                  g_stage.st[t]) == hipSuccess &&
                   hipEventRecord(g_stage.ev[b], g_stage.st[t]) == hipSuccess;
        };
        size_t use = 0;
        if (t < (int)chunks && !fetch(t, 2 * t)) { rc[t] = 1; return; }
        for (size_t c = t; c < chunks; c += STAGE_T, use++) {
            const int b = 2 * t + (int)(use & 1);
            if (c + STAGE_T < chunks && !fetch(c + STAGE_T, 2 * t + (int)((use + 1) & 1))) { rc[t] = 1; return; }
            if (hipEventSynchronize(g_stage.ev[b]) != hipSuccess) { rc[t] = 1; return; }
            const size_t at = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - at);
            memcpy(static_cast<char *>(dst) + at, g_stage.buf[b], len);
        }
    };
    std::thread pool[STAGE_T - 1];
    for (int t = 1; t < STAGE_T; t++)
        pool[t - 1] = std::thread(work, t);
    work(0);
    for (auto &th : pool)
        th.join();
    for (int t = 0; t < STAGE_T; t++)
        if (rc[t]) {
            (void)hipGetLastError();
            set_error("staged download of %zu bytes failed", bytes);
            return PCFIT_ERR_HIP;
        }
    return PCFIT_OK;
}

// ---- the part of a scan that one GPU of several needs ----
// Host passes over the z column of a scan in pageable memory, on several threads (a slice of the rows each).
namespace {
int host_threads(int32_t asked)
{
    if (asked > 0)
        return std::min<int32_t>(asked, 64);
    return (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
}

template <class F>
void host_parallel(int T, F &&f)
{
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++)
        pool.emplace_back(f, t);
    f(0);
    for (auto &th : pool)
        th.join();
}
}  // namespace

// Rows lo <= z <= hi of xyz (n rows), in file order, to a fresh device block: counted per chunk first (so that
// every chunk knows where its rows go), then packed chunk by chunk into the pinned staging buffers and sent.
int upload_rows_in_z_range(const double *xyz, int64_t n, double lo, double hi, DevBuf &out, int64_t *kept)
{
    constexpr int64_t ROWS = (int64_t)(STAGE_CHUNK / 24);
    const int64_t chunks = (n + ROWS - 1) / ROWS;
    std::vector<int64_t> at((size_t)chunks + 1, 0);
    {
        const int T = (int)std::min<int64_t>(host_threads(0), std::max<int64_t>(chunks, 1));
        host_parallel(T, [&](int t) {
            for (int64_t c = t; c < chunks; c += T) {
                const int64_t r0 = c * ROWS, r1 = std::min(n, r0 + ROWS);
                int64_t k = 0;
                for (int64_t i = r0; i < r1; i++) {
                    const double z = xyz[3 * i + 2];
                    k += (z >= lo && z <= hi) ? 1 : 0;
                }
                at[(size_t)c + 1] = k;
            }
        });
    }
    for (int64_t c = 0; c < chunks; c++)
        at[(size_t)c + 1] += at[(size_t)c];
    *kept = at[(size_t)chunks];
    PCFIT_TRY(out.alloc((size_t)*kept * 24));
    if (*kept == 0)
        return PCFIT_OK;
    int dev = 0;
    PCFIT_HIP(hipGetDevice(&dev));
    std::unique_lock<std::mutex> lock(g_stage_mu);
    if (!stage_make(dev)) {
        set_error("upload of a z range: no pinned staging buffers");
        return PCFIT_ERR_HIP;
    }
    PCFIT_HIP(hipStreamSynchronize(nullptr));
    int rc[STAGE_T] = {};
    auto work = [&](int t) {
        if (hipSetDevice(dev) != hipSuccess) { rc[t] = 1; return; }
        size_t use = 0;
        for (int64_t c = t; c < chunks; c += STAGE_T) {
            const int64_t rows = at[(size_t)c + 1] - at[(size_t)c];
            if (rows == 0)
                continue;
            const int b = 2 * t + (int)(use & 1);
            if (use >= 2 && hipEventSynchronize(g_stage.ev[b]) != hipSuccess) { rc[t] = 1; return; }
            use++;
            double *dstp = static_cast<double *>(g_stage.buf[b]);
            const int64_t r0 = c * ROWS, r1 = std::min(n, r0 + ROWS);
            for (int64_t i = r0; i < r1; i++) {
                const double z = xyz[3 * i + 2];
                if (z >= lo && z <= hi) {
                    dstp[0] = xyz[3 * i];
                    dstp[1] = xyz[3 * i + 1];
                    dstp[2] = z;
                    dstp += 3;
                }
            }
            if (hipMemcpyAsync(out.as<char>() + (size_t)at[(size_t)c] * 24, g_stage.buf[b], (size_t)rows * 24,
                               hipMemcpyHostToDevice, g_stage.st[t]) != hipSuccess ||
                hipEventRecord(g_stage.ev[b], g_stage.st[t]) != hipSuccess) { rc[t] = 1; return; }
        }
        if (hipStreamSynchronize(g_stage.st[t]) != hipSuccess)
            rc[t] = 1;
    };
    std::thread pool[STAGE_T - 1];
    for (int t = 1; t < STAGE_T; t++)
        pool[t - 1] = std::thread(work, t);
    work(0);
    for (auto &th : pool)
        th.join();
    for (int t = 0; t < STAGE_T; t++)
        if (rc[t]) {
            (void)hipGetLastError();
            set_error("staged upload of a z range failed");
            return PCFIT_ERR_HIP;
        }
    return PCFIT_OK;
}

int pool_release()
{
    {
        std::lock_guard<std::mutex> lock(g_stage_mu);
        stage_drop();
    }
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (int d = 0; d < POOL_MAX_DEV; d++)
        pool_drop(d);
    return PCFIT_OK;
}

}  // namespace pcfit

using namespace pcfit;

// ------------------------------------------------------------------ host passes over a scan's z column
PCFIT_API int pcfit_host_z_range(const double *xyz, int64_t n_points, int32_t n_threads, double *z_min_max)
{
    PCFIT_REQUIRE(z_min_max && n_points >= 0 && (xyz || n_points == 0), "host_z_range: null pointer or negative size");
    const double qnan = __builtin_nan("");
    z_min_max[0] = z_min_max[1] = qnan;
    if (n_points == 0)
        return PCFIT_OK;
    const int T = (int)std::min<int64_t>(host_threads(n_threads), std::max<int64_t>(1, n_points / 65536));
    std::vector<double> lo((size_t)T, __builtin_inf()), hi((size_t)T, -__builtin_inf());
    std::vector<int> bad((size_t)T, 0);
    host_parallel(T, [&](int t) {
        const int64_t r0 = n_points * t / T, r1 = n_points * (t + 1) / T;
        double l = __builtin_inf(), h = -__builtin_inf();
        int b = 0;
        for (int64_t i = r0; i < r1; i++) {
            const double z = xyz[3 * i + 2];
            b |= (z != z);
            l = z < l ? z : l;
            h = z > h ? z : h;
        }
        lo[(size_t)t] = l;
        hi[(size_t)t] = h;
        bad[(size_t)t] = b;
    });
    double l = lo[0], h = hi[0];
    int b = bad[0];
    for (int t = 1; t < T; t++) {
        l = std::min(l, lo[(size_t)t]);
        h = std::max(h, hi[(size_t)t]);
        b |= bad[(size_t)t];
    }
    if (!b) {
        z_min_max[0] = l;
        z_min_max[1] = h;
    }
    return PCFIT_OK;
}

PCFIT_API int pcfit_host_window_counts(const double *xyz, int64_t n_points, const double *z_lo, const double *z_hi,
                                       int64_t n_slices, int32_t n_threads, int64_t *counts)
{
    PCFIT_REQUIRE(n_points >= 0 && n_slices >= 1 && (xyz || n_points == 0) && z_lo && z_hi && counts,
                  "host_window_counts: null pointer or bad size");
    double geom[3];
    PCFIT_TRY(bin_check_windows(z_lo, z_hi, n_slices, geom));
    const int64_t S = n_slices;
    const double width = geom[1] - geom[0];
    const double step = S > 1 ? (geom[2] - geom[0]) / (double)(S - 1) : (width > 0 ? width : 1.0);
    PCFIT_REQUIRE(step > 0 && std::isfinite(step), "host_window_counts: windows must be ascending with finite bounds");
    const double inv_step = 1.0 / step, lo0 = geom[0];
    const int T = (int)std::min<int64_t>(host_threads(n_threads), std::max<int64_t>(1, n_points / 65536));
    std::vector<std::vector<int64_t>> part((size_t)T, std::vector<int64_t>((size_t)S, 0));
    host_parallel(T, [&](int t) {
        int64_t *c = part[(size_t)t].data();
        const int64_t r0 = n_points * t / T, r1 = n_points * (t + 1) / T;
        for (int64_t i = r0; i < r1; i++) {
            const double z = xyz[3 * i + 2];
            const double u = (z - lo0) * inv_step;
            if (!(u > -2.0) || !(u < (double)S + 2.0))
                continue;                                       // (NaN lands here too)
            const int64_t k0 = (int64_t)floor(u);
            for (int64_t k = std::max<int64_t>(k0 - 1, 0); k <= std::min<int64_t>(k0 + 1, S - 1); k++)
                c[k] += (z >= z_lo[k] && z <= z_hi[k]) ? 1 : 0;
        }
    });
    for (int64_t s = 0; s < S; s++) {
        int64_t v = 0;
        for (int t = 0; t < T; t++)
            v += part[(size_t)t][(size_t)s];
        counts[s] = v;
    }
    return PCFIT_OK;
}

PCFIT_API int pcfit_scan_upload_z_range(const double *xyz, int64_t n_points, double z_lo, double z_hi,
                                        pcfit_scan **scan, int64_t *n_kept)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(scan && n_kept && n_points >= 0 && (xyz || n_points == 0),
                  "scan_upload_z_range: null pointer or negative size");
    *scan = nullptr;
    std::unique_ptr<pcfit_scan> sc(new (std::nothrow) pcfit_scan);
    PCFIT_REQUIRE(sc != nullptr, "scan_upload_z_range: out of host memory");
    PCFIT_HIP(hipGetDevice(&sc->device));
    PCFIT_TRY(upload_rows_in_z_range(xyz, n_points, z_lo, z_hi, sc->xyz, n_kept));
    sc->n_points = *n_kept;
    PCFIT_TRY(scan_z_column(*sc, nullptr));
    PCFIT_HIP(hipStreamSynchronize(nullptr));
    *scan = sc.release();
    return PCFIT_OK;
}

// ------------------------------------------------------------------ plumbing
PCFIT_API int pcfit_version(void) { return 100; }
PCFIT_API const char *pcfit_last_error(void) { return g_err; }

PCFIT_API int pcfit_release_cache(void) { return pool_release(); }

PCFIT_API int pcfit_device_count(int *count)
{
    PCFIT_REQUIRE(count, "count is null");
    PCFIT_HIP(hipGetDeviceCount(count));
    return PCFIT_OK;
}
PCFIT_API int pcfit_set_device(int device)
{
    PCFIT_HIP(hipSetDevice(device));
    return PCFIT_OK;
}
PCFIT_API int pcfit_get_device(int *device)
{
    PCFIT_REQUIRE(device, "device is null");
    PCFIT_HIP(hipGetDevice(device));
    return PCFIT_OK;
}
PCFIT_API int pcfit_malloc(void **ptr, size_t bytes)
{
    PCFIT_ENTER();
    PCFIT_REQUIRE(ptr, "ptr is null");
    PCFIT_HIP(hipMalloc(ptr, bytes ? bytes : 8));
    return PCFIT_OK;
}
PCFIT_API int pcfit_free(void *ptr)
{
    PCFIT_ENTER();
    if (ptr) PCFIT_HIP(hipFree(ptr));
    return PCFIT_OK;
}
PCFIT_API int pcfit_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    PCFIT_ENTER();
    if (bytes) PCFIT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return PCFIT_OK;
}
PCFIT_API int pcfit_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    PCFIT_ENTER();
    if (bytes) PCFIT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return PCFIT_OK;
}
PCFIT_API int pcfit_memset(void *dst, int value, size_t bytes, void *stream)
{
    PCFIT_ENTER();
    if (bytes) PCFIT_HIP(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
    return PCFIT_OK;
}
PCFIT_API int pcfit_stream_synchronize(void *stream)
{
    PCFIT_ENTER();
    PCFIT_HIP(hipStreamSynchronize(as_stream(stream)));
    return PCFIT_OK;
}

// ------------------------------------------------------------ sampling tables
PCFIT_API int pcfit_seed_state(uint64_t seed, uint64_t state[4])
{
    PCFIT_REQUIRE(state, "state is null");
    // SeedSequence: the seed as little-endian 32-bit words (at least one), mixed
    // into a pool of four, from which eight output words are drawn
    const uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int nent = ent[1] ? 2 : 1;
    uint32_t pool[4], hc = 0x43b0d7e5u;
    auto hashmix = [&](uint32_t v) {
        v ^= hc;
        hc *= 0x931e8875u;
        v *= hc;
        return v ^ (v >> 16);
    };
    auto mix = [](uint32_t a, uint32_t b) {
        uint32_t r = 0xca01f9ddu * a - 0x4973f715u * b;
        return r ^ (r >> 16);
    };
    for (int i = 0; i < 4; i++)
        pool[i] = hashmix(i < nent ? ent[i] : 0u);
    for (int s = 0; s < 4; s++)
        for (int d = 0; d < 4; d++)
            if (s != d)
                pool[d] = mix(pool[d], hashmix(pool[s]));
    uint32_t w[8], hb = 0x8b51f9ddu;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3] ^ hb;
        hb *= 0x58f38dedu;
        v *= hb;
        w[i] = v ^ (v >> 16);
    }
    uint64_t q[4];
    for (int i = 0; i < 4; i++)
        q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    // PCG64 seeding: state = 0, inc = (seq << 1) | 1, step, add the seed, step
    const u128 initstate = ((u128)q[0] << 64) | q[1], initseq = ((u128)q[2] << 64) | q[3];
    const u128 inc = (initseq << 1) | 1;
    u128 st = inc;                       // 0 * mult + inc
    st += initstate;
    st = st * PCG_MULT + inc;
    state[0] = (uint64_t)(st >> 64);
    state[1] = (uint64_t)st;
    state[2] = (uint64_t)(inc >> 64);
    state[3] = (uint64_t)inc;
    return PCFIT_OK;
}

PCFIT_API int pcfit_choice3_table(const uint64_t state[4], int64_t n, int32_t trials, int32_t *out)
{
    PCFIT_REQUIRE(state && (out || trials == 0), "choice3: null pointer");
    PCFIT_REQUIRE(n >= 3 && n <= 0xFFFFFFFFll, "choice3: n must be 3..2^32-1 (got %lld)", (long long)n);
    PCFIT_REQUIRE(trials >= 0, "choice3: trials negative");
    choice3_table(state, n, trials, out);
    return PCFIT_OK;
}

// ------------------------------------------------- host-pointer entry points
namespace pcfit {
int launch_kasa(const double *, const double *, const int64_t *, int64_t, int64_t, double *, int32_t *, hipStream_t);
}

// Host-pointer calls hand a batch to the segment-parallel kernels when its slices are long and too few to
// fill the chip with one workgroup each (measured on 50 M points: 57 and 200 windows run 5-19x faster
// there, 2000 windows of 25 000 points slightly slower).
static bool use_split(int64_t max_len, int64_t n_slices)
{
    return max_len > 8192 && (n_slices < 1024 || max_len > 65536);
}

PCFIT_API int pcfit_kasa_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                               double *cxcyr, int32_t *status)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr && (total == 0 || (x && y)), "kasa: null pointer");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dout, dst;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dout.alloc(n_slices * 24));
    PCFIT_TRY(dst.alloc(n_slices * 4));
    if (use_split(max_len, n_slices)) {
        DevBuf dws;
        const size_t wsb = pcfit_split_workspace_bytes(total, n_slices, 0);
        PCFIT_TRY(dws.alloc(wsb));
        PCFIT_TRY(pcfit_kasa_split_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, total,
                                       dout.as<double>(), dst.as<int32_t>(), dws.p, wsb, st));
        PCFIT_TRY(dout.download(cxcyr, n_slices * 24, st));
        PCFIT_TRY(dst.download(status, n_slices * 4, st));
        PCFIT_HIP(hipStreamSynchronize(st));
        return PCFIT_OK;
    }
    PCFIT_TRY(launch_kasa(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, max_len,
                          dout.as<double>(), dst.as<int32_t>(), st));
    PCFIT_TRY(dout.download(cxcyr, n_slices * 24, st));
    PCFIT_TRY(dst.download(status, n_slices * 4, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_lm_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                             const double *init, double huber_delta, int32_t max_iter, double lm_lambda0,
                             double tol_update, double tol_cost, double *out, int32_t *iters)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(init && out && (total == 0 || (x && y)), "lm: null pointer");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dinit, dout, dit;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dinit.upload(init, n_slices * 24, st));
    PCFIT_TRY(dout.alloc(n_slices * 24));
    PCFIT_TRY(dit.alloc(n_slices * 16));
    DevBuf dws;
    if (use_split(max_len, n_slices)) {
        const size_t wsb = pcfit_split_workspace_bytes(total, n_slices, 0);
        PCFIT_TRY(dws.alloc(wsb));
        PCFIT_TRY(pcfit_lm_split_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, total,
                                     dinit.as<double>(), huber_delta, max_iter, lm_lambda0, tol_update, tol_cost,
                                     dout.as<double>(), dit.as<int32_t>(), dws.p, wsb, st));
    } else {
        PCFIT_TRY(pcfit_lm_batch_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, max_len,
                                     dinit.as<double>(), huber_delta, max_iter, lm_lambda0, tol_update, tol_cost,
                                     dout.as<double>(), dit.as<int32_t>(), st));
    }
    PCFIT_TRY(dout.download(out, n_slices * 24, st));
    PCFIT_TRY(dit.download(iters, n_slices * 16, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_ransac_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                 double inlier_tol, int32_t trials, double min_inliers_frac, const double *r_range,
                                 uint64_t seed, double *model, int32_t *score, int32_t *best_trial, uint8_t *mask,
                                 int32_t *status)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(model && score && best_trial && status && (total == 0 || (x && y && mask)), "ransac: null pointer");
    PCFIT_REQUIRE(trials >= 0, "ransac: trials negative");
    // one sampling table per distinct slice population
    uint64_t state[4];
    PCFIT_TRY(pcfit_seed_state(seed, state));
    std::map<int64_t, int32_t> table_id;
    std::vector<int32_t> of_slice(n_slices, 0), tables;
    for (int64_t s = 0; s < n_slices; s++) {
        const int64_t n = offsets[s + 1] - offsets[s];
        if (n < 3) continue;
        auto it = table_id.find(n);
        if (it == table_id.end()) {
            it = table_id.emplace(n, (int32_t)table_id.size()).first;
            tables.resize(tables.size() + (size_t)trials * 3);
            PCFIT_TRY(pcfit_choice3_table(state, n, trials, tables.data() + (size_t)it->second * trials * 3));
        }
        of_slice[s] = it->second;
    }
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dtab, dof, dmodel, dscore, dbest, dmask, dst, dws;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dtab.upload(tables.data(), tables.size() * 4, st));
    PCFIT_TRY(dof.upload(of_slice.data(), n_slices * 4, st));
    PCFIT_TRY(dmodel.alloc(n_slices * 24));
    PCFIT_TRY(dscore.alloc(n_slices * 4));
    PCFIT_TRY(dbest.alloc(n_slices * 4));
    PCFIT_TRY(dmask.alloc(total));
    PCFIT_TRY(dst.alloc(n_slices * 4));
    PCFIT_TRY(dws.alloc(pcfit_ransac_workspace_bytes(n_slices)));
    PCFIT_TRY(pcfit_ransac_batch_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, inlier_tol,
                                     trials, min_inliers_frac, r_range, dtab.as<int32_t>(), dof.as<int32_t>(),
                                     dmodel.as<double>(), dscore.as<int32_t>(), dbest.as<int32_t>(),
                                     dmask.as<uint8_t>(), dst.as<int32_t>(), dws.p, dws.bytes, st));
    PCFIT_TRY(dmodel.download(model, n_slices * 24, st));
    PCFIT_TRY(dscore.download(score, n_slices * 4, st));
    PCFIT_TRY(dbest.download(best_trial, n_slices * 4, st));
    PCFIT_TRY(dmask.download(mask, total, st));
    PCFIT_TRY(dst.download(status, n_slices * 4, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_residuals_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                    const double *cxcyr, double *theta, double *dr, double *radius, int32_t n_bins,
                                    const double *edges, int64_t *hist, double *rmax_rmin, int64_t *argmax_argmin)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr && rmax_rmin && argmax_argmin && (total == 0 || (x && y)), "residuals: null pointer");
    PCFIT_REQUIRE(n_bins >= 0 && (n_bins == 0 || (edges && hist)), "residuals: bad histogram arguments");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dfit, dth, ddr, drad, dedge, dhist, drm, dam;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dfit.upload(cxcyr, n_slices * 24, st));
    if (theta) PCFIT_TRY(dth.alloc(total * 8));
    if (dr) PCFIT_TRY(ddr.alloc(total * 8));
    if (radius) PCFIT_TRY(drad.alloc(total * 8));
    if (n_bins) {
        PCFIT_TRY(dedge.upload(edges, (size_t)(n_bins + 1) * 8, st));
        PCFIT_TRY(dhist.alloc((size_t)n_slices * n_bins * 8));
    }
    PCFIT_TRY(drm.alloc(n_slices * 16));
    PCFIT_TRY(dam.alloc(n_slices * 16));
    DevBuf dws;
    if (use_split(max_len, n_slices)) {
        const size_t wsb = pcfit_split_workspace_bytes(total, n_slices, 0);
        PCFIT_TRY(dws.alloc(wsb));
        PCFIT_TRY(pcfit_residuals_split_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, total,
                                            dfit.as<double>(), dth.as<double>(), ddr.as<double>(), drad.as<double>(),
                                            n_bins, dedge.as<double>(), dhist.as<int64_t>(), drm.as<double>(),
                                            dam.as<int64_t>(), dws.p, wsb, st));
    } else {
        PCFIT_TRY(pcfit_residuals_batch_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices,
                                            dfit.as<double>(), dth.as<double>(), ddr.as<double>(), drad.as<double>(),
                                            n_bins, dedge.as<double>(), dhist.as<int64_t>(), drm.as<double>(),
                                            dam.as<int64_t>(), st));
    }
    if (theta) PCFIT_TRY(dth.download(theta, total * 8, st));
    if (dr) PCFIT_TRY(ddr.download(dr, total * 8, st));
    if (radius) PCFIT_TRY(drad.download(radius, total * 8, st));
    if (n_bins) PCFIT_TRY(dhist.download(hist, (size_t)n_slices * n_bins * 8, st));
    PCFIT_TRY(drm.download(rmax_rmin, n_slices * 16, st));
    PCFIT_TRY(dam.download(argmax_argmin, n_slices * 16, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_boundary_residuals_batch(const double *x, const double *y, const int64_t *offsets,
                                             int64_t n_slices, const double *cxcyr, int32_t bins, int64_t *sel,
                                             int32_t n_bins, const double *edges, int64_t *hist, double *rmax_rmin,
                                             int64_t *argmax_argmin)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr && sel && rmax_rmin && argmax_argmin && (total == 0 || (x && y)),
                  "boundary_residuals: null pointer");
    PCFIT_REQUIRE(bins >= 1 && n_bins >= 0, "boundary_residuals: bins must be >= 1, n_bins >= 0");
    PCFIT_REQUIRE(n_bins == 0 || (edges && hist), "boundary_residuals: histogram asked for without edges/hist");
    PCFIT_REQUIRE(max_len < 0x7fffffffll, "boundary_residuals: slice longer than 2^31-2 points");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dfit, dsel, dedge, dhist, drm, dam;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dfit.upload(cxcyr, n_slices * 24, st));
    PCFIT_TRY(dsel.alloc((size_t)n_slices * bins * 8));
    if (n_bins > 0) {
        PCFIT_TRY(dedge.upload(edges, (size_t)(n_bins + 1) * 8, st));
        PCFIT_TRY(dhist.alloc((size_t)n_slices * n_bins * 8));
    }
    PCFIT_TRY(drm.alloc(n_slices * 16));
    PCFIT_TRY(dam.alloc(n_slices * 16));
    PCFIT_TRY(pcfit_boundary_residuals_batch_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices,
                                                 dfit.as<double>(), bins, dsel.as<int64_t>(), n_bins, dedge.as<double>(),
                                                 dhist.as<int64_t>(), drm.as<double>(), dam.as<int64_t>(), st));
    PCFIT_TRY(dsel.download(sel, (size_t)n_slices * bins * 8, st));
    if (n_bins > 0) PCFIT_TRY(dhist.download(hist, (size_t)n_slices * n_bins * 8, st));
    PCFIT_TRY(drm.download(rmax_rmin, n_slices * 16, st));
    PCFIT_TRY(dam.download(argmax_argmin, n_slices * 16, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}

PCFIT_API int pcfit_angle_max_batch(const double *x, const double *y, const int64_t *offsets, int64_t n_slices,
                                    const double *cxcyr, int32_t bins, int64_t *sel)
{
    PCFIT_ENTER();
    int64_t total, max_len;
    PCFIT_TRY(check_offsets(offsets, n_slices, &total, &max_len));
    if (n_slices == 0) return PCFIT_OK;
    PCFIT_REQUIRE(cxcyr && sel && (total == 0 || (x && y)), "angle_max: null pointer");
    PCFIT_REQUIRE(max_len < 0xffffffffll, "angle_max: slice longer than 2^32-2 points");
    hipStream_t st = nullptr;
    DevBuf dx, dy, doff, dfit, dsel;
    PCFIT_TRY(dx.upload(x, total * 8, st));
    PCFIT_TRY(dy.upload(y, total * 8, st));
    PCFIT_TRY(doff.upload(offsets, (n_slices + 1) * 8, st));
    PCFIT_TRY(dfit.upload(cxcyr, n_slices * 24, st));
    PCFIT_TRY(dsel.alloc((size_t)n_slices * (bins > 0 ? bins : 1) * 8));
    DevBuf dws;
    if (use_split(max_len, n_slices) && bins >= 1) {
        const size_t wsb = pcfit_split_workspace_bytes(total, n_slices, bins);
        PCFIT_TRY(dws.alloc(wsb));
        PCFIT_TRY(pcfit_angle_max_split_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices, total,
                                            dfit.as<double>(), bins, dsel.as<int64_t>(), dws.p, wsb, st));
    } else {
        PCFIT_TRY(pcfit_angle_max_batch_dev(dx.as<double>(), dy.as<double>(), doff.as<int64_t>(), n_slices,
                                            dfit.as<double>(), bins, dsel.as<int64_t>(), st));
    }
    PCFIT_TRY(dsel.download(sel, (size_t)n_slices * bins * 8, st));
    PCFIT_HIP(hipStreamSynchronize(st));
    return PCFIT_OK;
}
