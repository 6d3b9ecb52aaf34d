// qm_runtime.hip -- what the engine's host side stands on: the thread's error text, pooled HIP
// streams, the pinned bounce buffers every host pointer travels through, and the process-wide pool
// of recycled device memory.  Host code only.
#include "qm_engine.hpp"

namespace {
thread_local std::string g_error;
}

int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return 1;
}
void clear_error() { g_error.clear(); }
const char *error_text() { return g_error.c_str(); }

// Three habits that come from one study (round 4; DESIGN.md section 6, profiles/r04_gpu_sharing_study.txt):
// with 16 processes sharing the GPU and an engine made per call, about one call in 1e4 went wrong --
// host outputs with holes (the runtime's copy into the caller's pageable memory), wrong values from
// an engine's first step (freshly allocated device memory), and, in a program without this library,
// stale data behind a stream created per call (tools/micro/d2h_order.hip).  One process alone, or
// one long-lived engine under the same sharing: never.  So an engine's own stream comes from a
// per-device pool and goes back to it (pooled streams are never destroyed), inputs and results
// travel between host and device memory through pinned memory (below), and device memory is
// recycled (pool_alloc).
static std::mutex g_stream_mutex;
static std::vector<std::pair<int, hipStream_t>> g_idle_streams;
hipError_t acquire_stream(int device, hipStream_t *out) {
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (size_t i = 0; i < g_idle_streams.size(); ++i)
            if (g_idle_streams[i].first == device) {
                *out = g_idle_streams[i].second;
                g_idle_streams.erase(g_idle_streams.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void park_stream(int device, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    g_idle_streams.emplace_back(device, s);
}

// Results travel back to the host through a pinned bounce buffer and a CPU copy; inputs the same way
// in the other direction (the caller's buffer is copied into pinned memory by the CPU before the call
// returns, so it may be a temporary).  Both directions return when the data is in place.  One buffer
// per DEVICE (its own lock: engines on different GPUs of a threaded process do not queue behind each
// other), 32 MB in two halves that ping-pong: while the DMA engine fills / drains one half the CPU
// copies the other (round 5; round 4's single buffer did DMA -> wait -> memcpy strictly in turn), and
// pieces of 2 MB and more are copied by a few host threads (one core moves ~10 GB/s, a PCIe 5 link
// four times that).  (The statistics-only flag ring of the screened sweep is pinned memory of its own
// and stays asynchronous.)
static constexpr size_t kBounceHalf = 16u << 20;
struct Bounce {
    std::mutex mutex;
    char *half[2] = {nullptr, nullptr};
    hipEvent_t landed[2] = {nullptr, nullptr};      // the DMA that last touched the half is done
    bool in_flight[2] = {false, false};             // ... and has not been waited for yet (uploads return
                                                    // without waiting: the data is in pinned memory, the
                                                    // copy is ordered on the stream like the kernel after it)
};
static std::mutex g_bounce_map_mutex;
static std::vector<std::pair<int, Bounce *>> g_bounces;    // (device, buffer): never freed
static Bounce *bounce_of_current_device() {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_bounce_map_mutex);
    for (auto &b : g_bounces)
        if (b.first == device) return b.second;
    g_bounces.emplace_back(device, new Bounce());
    return g_bounces.back().second;
}
static hipError_t ensure_bounce(Bounce *b) {               // (caller holds b->mutex)
    if (b->half[0]) return hipSuccess;
    char *p = nullptr;
    hipError_t r = hipHostMalloc(reinterpret_cast<void **>(&p), 2 * kBounceHalf, hipHostMallocPortable);
    if (r != hipSuccess) return r;
    for (int i = 0; i < 2 && r == hipSuccess; ++i) r = hipEventCreateWithFlags(&b->landed[i], hipEventDisableTiming);
    if (r != hipSuccess) {
        (void)hipHostFree(p);
        return r;
    }
    b->half[0] = p;
    b->half[1] = p + kBounceHalf;
    return hipSuccess;
}
// CPU copy of n bytes; from 2 MB on, on up to 8 threads of >= 1 MB each (one core moves ~8 GB/s: a
// 3 MB timestep of a long-window config would otherwise cost the pushing thread 0.4 ms)
void host_copy(void *dst, const void *src, size_t n) {
    constexpr size_t kGrain = (size_t)1 << 20;
    unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = std::min<size_t>({n / kGrain, hw ? hw : 4u, (size_t)8});
    if (nt < 2) {
        std::memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = ((n + nt - 1) / nt + 63) / 64 * 64;
    for (size_t t = 1; t < nt; ++t) {
        const size_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        pool.emplace_back([=] { std::memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, hi - lo); });
    }
    std::memcpy(dst, src, std::min(n, per));
    for (auto &th : pool) th.join();
}
// rows of `width` bytes between a packed buffer and a pitched one (either direction)
static void host_copy_rows(char *dst, size_t dpitch, const char *src, size_t spitch, size_t width, size_t rows) {
    if (dpitch == width && spitch == width) return host_copy(dst, src, width * rows);
    unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = std::min<size_t>({width * rows / ((size_t)2 << 20), hw ? hw : 4u, (size_t)8});
    auto run = [=](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) std::memcpy(dst + i * dpitch, src + i * spitch, width);
    };
    if (nt < 2) return run(0, rows);
    std::vector<std::thread> pool;
    const size_t per = (rows + nt - 1) / nt;
    for (size_t t = 1; t < nt; ++t)
        if (t * per < rows) pool.emplace_back(run, t * per, std::min(rows, (t + 1) * per));
    run(0, std::min(rows, per));
    for (auto &th : pool) th.join();
}

// The generic two-half pipeline.  `pieces` pieces; issue(i, half) enqueues piece i's DMA on stream s
// (device -> half for downloads, half -> device for uploads), host(i, half) is the CPU side of piece i
// (half -> caller for downloads, caller -> half for uploads).
template <typename Issue, typename Host>
static hipError_t bounce_pipeline(bool download, size_t pieces, hipStream_t s, Issue issue, Host host) {
    Bounce *b = bounce_of_current_device();
    if (!b) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(b->mutex);
    hipError_t r = ensure_bounce(b);
    if (r != hipSuccess) return r;
    // a half whose last DMA (an upload's) nobody has waited for yet must land before it is touched again
    auto settle = [&](int h) -> hipError_t {
        if (!b->in_flight[h]) return hipSuccess;
        b->in_flight[h] = false;
        return hipEventSynchronize(b->landed[h]);
    };
    if (download) {
        // DMA of piece i runs while the CPU empties piece i - 1
        for (size_t i = 0; i <= pieces; ++i) {
            if (i < pieces) {
                r = settle((int)(i & 1));
                if (r == hipSuccess) r = issue(i, b->half[i & 1]);
                if (r == hipSuccess) r = hipEventRecord(b->landed[i & 1], s);
                if (r != hipSuccess) return r;
            }
            if (i > 0) {
                r = hipEventSynchronize(b->landed[(i - 1) & 1]);
                if (r != hipSuccess) return r;
                host(i - 1, b->half[(i - 1) & 1]);
            }
        }
        return hipSuccess;
    }
    // upload: the CPU fills piece i while the DMA of piece i - 1 drains the other half.  Returns when the
    // caller's data is in pinned memory (it may be a temporary); the DMA itself is stream-ordered work.
    for (size_t i = 0; i < pieces; ++i) {
        r = settle((int)(i & 1));
        if (r != hipSuccess) return r;
        host(i, b->half[i & 1]);
        r = issue(i, b->half[i & 1]);
        if (r == hipSuccess) r = hipEventRecord(b->landed[i & 1], s);
        if (r != hipSuccess) return r;
        b->in_flight[i & 1] = true;
    }
    return hipSuccess;
}

hipError_t copy_back(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const size_t pieces = (bytes + kBounceHalf - 1) / kBounceHalf;
    auto len = [=](size_t i) { return std::min(kBounceHalf, bytes - i * kBounceHalf); };
    return bounce_pipeline(true, pieces, s,
        [&](size_t i, char *half) {
            return hipMemcpyAsync(half, static_cast<const char *>(src) + i * kBounceHalf, len(i),
                                  hipMemcpyDeviceToHost, s);
        },
        [&](size_t i, char *half) { host_copy(static_cast<char *>(dst) + i * kBounceHalf, half, len(i)); });
}
hipError_t copy_back_pieces(void *const *dst, const void *src, int pieces, size_t bytes, hipStream_t s) {
    if ((size_t)pieces * bytes > kBounceHalf) {          // (too long for one half: piece by piece)
        for (int i = 0; i < pieces; ++i) {
            const hipError_t r = copy_back(dst[i], static_cast<const char *>(src) + (size_t)i * bytes, bytes, s);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    }
    return bounce_pipeline(true, 1, s,
        [&](size_t, char *half) { return hipMemcpyAsync(half, src, (size_t)pieces * bytes, hipMemcpyDeviceToHost, s); },
        [&](size_t, char *half) {
            for (int i = 0; i < pieces; ++i) std::memcpy(dst[i], half + (size_t)i * bytes, bytes);
        });
}
hipError_t copy_in(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const size_t pieces = (bytes + kBounceHalf - 1) / kBounceHalf;
    auto len = [=](size_t i) { return std::min(kBounceHalf, bytes - i * kBounceHalf); };
    return bounce_pipeline(false, pieces, s,
        [&](size_t i, char *half) {
            return hipMemcpyAsync(static_cast<char *>(dst) + i * kBounceHalf, half, len(i),
                                  hipMemcpyHostToDevice, s);
        },
        [&](size_t i, char *half) { host_copy(half, static_cast<const char *>(src) + i * kBounceHalf, len(i)); });
}
// pitched forms: `height` rows of `width` bytes, dpitch / spitch bytes apart
hipError_t copy_back_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width,
                        size_t height, hipStream_t s) {
    if (width == dpitch && width == spitch) return copy_back(dst, src, width * height, s);
    if (width > kBounceHalf) {                      // (rows longer than a half: one by one)
        for (size_t row = 0; row < height; ++row) {
            const hipError_t r = copy_back(static_cast<char *>(dst) + row * dpitch,
                                           static_cast<const char *>(src) + row * spitch, width, s);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    }
    const size_t rows_at_once = kBounceHalf / width;
    const size_t pieces = (height + rows_at_once - 1) / rows_at_once;
    auto rows = [=](size_t i) { return std::min(rows_at_once, height - i * rows_at_once); };
    return bounce_pipeline(true, pieces, s,
        [&](size_t i, char *half) {
            return hipMemcpy2DAsync(half, width, static_cast<const char *>(src) + i * rows_at_once * spitch,
                                    spitch, width, rows(i), hipMemcpyDeviceToHost, s);
        },
        [&](size_t i, char *half) {
            host_copy_rows(static_cast<char *>(dst) + i * rows_at_once * dpitch, dpitch, half, width, width, rows(i));
        });
}
hipError_t copy_in_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width,
                      size_t height, hipStream_t s) {
    if (width == dpitch && width == spitch) return copy_in(dst, src, width * height, s);
    if (width > kBounceHalf) {                      // (rows longer than a half: one by one)
        for (size_t row = 0; row < height; ++row) {
            const hipError_t r = copy_in(static_cast<char *>(dst) + row * dpitch,
                                         static_cast<const char *>(src) + row * spitch, width, s);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    }
    const size_t rows_at_once = kBounceHalf / width;
    const size_t pieces = (height + rows_at_once - 1) / rows_at_once;
    auto rows = [=](size_t i) { return std::min(rows_at_once, height - i * rows_at_once); };
    return bounce_pipeline(false, pieces, s,
        [&](size_t i, char *half) {
            return hipMemcpy2DAsync(static_cast<char *>(dst) + i * rows_at_once * dpitch, dpitch, half, width,
                                    width, rows(i), hipMemcpyHostToDevice, s);
        },
        [&](size_t i, char *half) {
            host_copy_rows(half, width, static_cast<const char *>(src) + i * rows_at_once * spitch, spitch, width, rows(i));
        });
}

// Device memory is recycled inside the process: a released block is parked and handed to the next
// request of (about) its size instead of going back to the driver -- what every long-running GPU
// runtime does, here for a reason found the hard way (round 4, profiles/r04_gpu_sharing_study.txt):
// with 16 processes sharing the GPU, an engine made, used once and destroyed in a loop returned wrong
// maxima on 10-100 % of the samples about once per 1e4 engines (two in 21 000, the round-2 kernels on
// a fresh engine; none in 220 000 steps of ONE engine under the same sharing) -- results of kernels
// that read buffers another kernel had just written into freshly mapped memory.  With recycled
// blocks the address space of a process stops changing after its first engines.  Per DEVICE at most
// pool_keep_bytes() of released blocks stay parked (default 2 GB, QM_HIP_POOL_KEEP_MB in the
// environment; beyond it the largest go back to the driver -- many processes sharing one GPU must not
// each sit on memory the others need); qm_release_cached_memory() returns them all.
//
// QM_HIP_POOL_POISON=1 (round 5, the question the study left open -- does anything read device memory
// it has not written?): every block the pool hands out, recycled or fresh, is filled with 0xFF bytes
// first (a NaN in every double, -1 in every integer).  A kernel that depends on the content of a
// fresh buffer then computes NaN / indexes out of range instead of happening to find the right derived
// data in a recycled block.  The GPU suite and tools/stress_engines.py run green with it (DESIGN.md
// section 6).
struct PoolBlock {
    int device;
    size_t bytes;
    void *p;
};
static std::mutex g_pool_mutex;
static std::vector<PoolBlock> g_pool_idle, g_pool_live;

static size_t pool_keep_bytes() {
    static const size_t keep = [] {
        const char *v = getenv("QM_HIP_POOL_KEEP_MB");
        const long long mb = v ? atoll(v) : 2048;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return keep;
}
static bool pool_poison() {
    static const bool on = [] {
        const char *v = getenv("QM_HIP_POOL_POISON");
        return v && atoi(v) != 0;
    }();
    return on;
}

// give the largest idle blocks of `device` (any device: -1) back to the driver until at most `keep`
// bytes of it stay parked
static void pool_trim_locked(int device, size_t keep) {
    for (;;) {
        size_t total = 0, big = g_pool_idle.size();
        for (size_t i = 0; i < g_pool_idle.size(); ++i) {
            if (device >= 0 && g_pool_idle[i].device != device) continue;
            total += g_pool_idle[i].bytes;
            if (big == g_pool_idle.size() || g_pool_idle[i].bytes > g_pool_idle[big].bytes) big = i;
        }
        if (total <= keep || big == g_pool_idle.size()) return;
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (prev != g_pool_idle[big].device) (void)hipSetDevice(g_pool_idle[big].device);
        (void)hipFree(g_pool_idle[big].p);
        if (prev >= 0 && prev != g_pool_idle[big].device) (void)hipSetDevice(prev);
        g_pool_idle.erase(g_pool_idle.begin() + (long)big);
    }
}

// (the caller has made the engine's device current)
hipError_t pool_alloc(void **out, size_t bytes) {
    const size_t unit = bytes < ((size_t)1 << 20) ? 256 : (size_t)2 << 20;
    const size_t want = (std::max<size_t>(bytes, 1) + unit - 1) / unit * unit;
    int device = 0;
    hipError_t r = hipGetDevice(&device);
    if (r != hipSuccess) return r;
    size_t got = 0;
    *out = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t best = g_pool_idle.size();
        for (size_t i = 0; i < g_pool_idle.size(); ++i) {
            const PoolBlock &b = g_pool_idle[i];
            if (b.device != device || b.bytes < want || b.bytes > want + want / 4) continue;
            if (best == g_pool_idle.size() || b.bytes < g_pool_idle[best].bytes) best = i;
        }
        if (best != g_pool_idle.size()) {
            *out = g_pool_idle[best].p;
            got = g_pool_idle[best].bytes;
            g_pool_live.push_back(g_pool_idle[best]);
            g_pool_idle.erase(g_pool_idle.begin() + (long)best);
        }
    }
    if (!*out) {
        r = hipMalloc(out, want);
        if (r != hipSuccess) {                          // make room: everything parked here goes back
            (void)hipGetLastError();
            {
                std::lock_guard<std::mutex> lock(g_pool_mutex);
                pool_trim_locked(device, 0);
            }
            r = hipMalloc(out, want);
            if (r != hipSuccess) return r;
        }
        got = want;
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_pool_live.push_back(PoolBlock{device, want, *out});
    }
    if (pool_poison()) {
        r = hipMemset(*out, 0xFF, got);                 // (the null stream: ordered before everything after)
        if (r == hipSuccess) r = hipDeviceSynchronize();
    }
    return r;
}

// Releasing: as hipFree, nothing that was enqueued before may still be using a block when somebody
// else gets it -- one device-wide wait.  An engine's teardown or a table's eviction releases a few
// dozen buffers: inside a PoolReleaseScope they are collected and parked behind ONE wait (round 4
// waited once per buffer).
static thread_local int g_pool_scope = 0;
static thread_local std::vector<void *> g_pool_deferred;

static void pool_park(void *p) {                               // (the device is idle; caller holds no lock)
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (size_t i = 0; i < g_pool_live.size(); ++i) {
        if (g_pool_live[i].p != p) continue;
        const int device = g_pool_live[i].device;
        g_pool_idle.push_back(g_pool_live[i]);
        g_pool_live.erase(g_pool_live.begin() + (long)i);
        pool_trim_locked(device, pool_keep_bytes());
        return;
    }
    (void)hipFree(p);                                   // (not one of ours: cannot happen)
}
void pool_free(void *p) {
    if (g_pool_scope > 0) {
        g_pool_deferred.push_back(p);
        return;
    }
    (void)hipDeviceSynchronize();
    pool_park(p);
}
PoolReleaseScope::PoolReleaseScope() { ++g_pool_scope; }
PoolReleaseScope::~PoolReleaseScope() {
    if (--g_pool_scope > 0 || g_pool_deferred.empty()) return;
    (void)hipDeviceSynchronize();
    for (void *p : g_pool_deferred) pool_park(p);
    g_pool_deferred.clear();
}


void pool_release_all_idle() {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    pool_trim_locked(-1, 0);
}

extern "C" {

const char *qm_last_error(void) { return error_text(); }

int qm_release_cached_memory(void) {
    pool_release_all_idle();
    return 0;
}

int qm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// What the library was built from: the constants and source digest of the generated
// shift-reuse loops (gen_shift_asm.py: build_info) and the development defines the kernels' unit was compiled
// with.  The product build reads "overlay=none" and "defines=none".
const char *qm_build_info(void) {
    static const std::string info = std::string("shift loops: ") + qm::kShiftGenInfo + "; defines=" +
                                    qm::shift_unit_defines();
    return info.c_str();
}

}  // extern "C"
