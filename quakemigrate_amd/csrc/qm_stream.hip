// qm_stream.hip -- the continuous detect sweep as a native pipeline (C ABI part 3, include/qmhip.h).
//
// The reference's QuakeScan._continuous_compute (quakemigrate/signal/scan.py:407-470; the loop
// itself :434-448) walks the timesteps serially: read -> onsets -> migrate -> find_max_coa -> append.
// Timesteps are independent given their onsets, so the hot-path part of that loop is a pipeline over
// a ring of `depth` slots, each holding `steps_per_launch` timesteps:
//
//   caller's thread : qm_stream_push   CPU copy of a step's log-onsets into the slot's PINNED input
//   copy stream     : H2D of the slot's inputs                 (overlaps the previous launch's kernel)
//                     -- or, for slots of up to kPullBytes (the example-sized grids, one timestep per launch: a
//                     step is a fraction of a millisecond and the wait for another stream's copy costs a quarter
//                     of it), NO copy command at all: a small kernel on the engine's stream PULLS the pinned
//                     input over the bus into the slot's device buffer, in order before the detect launch
//   engine's stream : ONE fused-detect launch for the slot's K steps (qm_engine_detect_batch); its
//                     combine kernel writes the results, packed as [3][K x n_samples] float64, STRAIGHT
//                     into the slot's pinned host buffer (a few KB per timestep; no D2H command)
//   caller's thread : qm_stream_pop    waits for the launch's event, CPU copy of a step's three series
//
// Round 4 drove a pipeline like this from Python (three D2H copies, three events and NumPy staging per
// launch): with the copies inside the clock the example-sized grids lost 30-40 % of their kernel
// rate and K steps per launch bought nothing (profiles/r04_bench_{C1,E1,E2}_k*.json).  Here a launch
// costs the host one memcpy per pushed step and a handful of enqueues; ordering is HIP events, the
// host blocks only in qm_stream_pop, and only for the oldest launch: 8 timesteps per launch run at
// x1.00 of the resident step on those grids.  The caller's pointers are never handed to the HIP
// runtime.
#include "qm_engine.hpp"

#include <deque>


struct qm_stream {
    qm_engine *e = nullptr;
    int n_rows = 0, T = 0, fsmp = 0, lsmp = 0, available = 0, ns = 0, K = 1, depth = 2;
    int64_t n_nodes_total = 0;
    uint64_t table_serial = 0;          // the table the stream was made on (TableState::serial)
    hipStream_t copy_stream = nullptr;
    struct Slot {
        double *h_on = nullptr;         // pinned [K][n_rows][T]
        double *d_on = nullptr;         // device, the same
        double *h_out = nullptr;        // pinned [3][K * ns]: max_coa, max_norm_coa, indices (int64 bits) --
                                        // written by the kernels themselves
        hipEvent_t copied = nullptr;    // the inputs are on the device
        hipEvent_t done = nullptr;      // the launch has finished: the results are in h_out
        int n = 0;                      // steps launched from this slot
        int taken = 0;                  // ... of which popped
        bool in_flight = false;
    };
    std::vector<Slot> slots;
    std::deque<int> order;              // slots in flight, oldest first
    int fill_slot = 0, fill_n = 0;      // slot being filled, steps pushed into it so far
    int64_t launched_steps = 0, popped_steps = 0, launches = 0;
    unsigned long long *h_stamp = nullptr;   // ("stream_stamps") pinned [2][4096]
    bool pulled = false;                // the last launch's inputs were pulled by a kernel (no copy command)
};

namespace {

// Slots this small are pulled by a kernel instead of copied by a command on another stream (round 6).  One
// timestep per launch on the example-sized grids ran at x1.27-1.30 of the resident step with the copy command --
// back to back under the tracer, so not the copy's duration: the dependency across two streams (a signal the
// runtime's host thread has to forward) -- profiles/r06_trace_stream_k1.json.  A kernel that reads pinned host
// memory moves ~170 KB (C1) in a few microseconds and needs no event: the engine's stream orders it.  Large
// slots (C3: 1.5 MB per timestep, 41 ms of kernel to hide the copy behind) keep the copy stream.
constexpr size_t kPullBytes = 1u << 20;

using pull2 = double __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void pull_kernel(const pull2 *__restrict__ src, pull2 *__restrict__ dst,
                                                   size_t n2, const double *__restrict__ src1, double *__restrict__ dst1,
                                                   size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride)
        dst[i] = __builtin_nontemporal_load(src + i);
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) dst1[n - 1] = src1[n - 1];
}

// ("stream_stamps": the GPU's own clock before and behind every launch of a stream, digested at its destruction)
__global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }

size_t step_in(const qm_stream *s) { return (size_t)s->n_rows * s->T; }

int alive(const qm_stream *s, const char *what) {
    if (!s) return fail("%s: NULL argument", what);
    if (!s->e) return fail("%s: the stream's engine has been destroyed", what);
    return 0;
}

void free_slot(qm_stream::Slot &sl) {
    if (sl.h_on) (void)hipHostFree(sl.h_on);
    if (sl.h_out) (void)hipHostFree(sl.h_out);
    if (sl.d_on) pool_free(sl.d_on);
    for (hipEvent_t ev : {sl.copied, sl.done})
        if (ev) (void)hipEventDestroy(ev);
    sl = qm_stream::Slot{};
}

// everything the stream holds on its engine's device goes back (the engine's device is current)
void release_stream(qm_stream *s) {
    (void)hipStreamSynchronize(s->e->stream);
    if (s->h_stamp) {
        // launch i: [begin_i, end_i]; gap_i = begin_i - end_(i-1)   (wall_clock64: 100 MHz)
        const int n = (int)std::min<int64_t>(s->launches, 4096);
        std::vector<double> busy, gap;
        for (int i = 8; i < n; ++i) {
            busy.push_back((double)(s->h_stamp[4096 + i] - s->h_stamp[i]) * 0.01);
            gap.push_back((double)(s->h_stamp[i] - s->h_stamp[4096 + i - 1]) * 0.01);
        }
        for (int i = 0; i < (int)gap.size(); ++i)
            if (gap[i] > 200.0) std::fprintf(stderr, "qm_stream stamps: gap of %.1f us before launch %d\n", gap[i], i + 8);
        std::sort(busy.begin(), busy.end());
        std::sort(gap.begin(), gap.end());
        if (!busy.empty())
            std::fprintf(stderr, "qm_stream stamps: %d launches; launch busy us median %.1f p90 %.1f; gap between launches "
                         "us median %.1f p90 %.1f max %.1f\n", n, busy[busy.size() / 2], busy[busy.size() * 9 / 10],
                         gap[gap.size() / 2], gap[gap.size() * 9 / 10], gap.back());
        (void)hipHostFree(s->h_stamp);
        s->h_stamp = nullptr;
    }
    if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    {
        PoolReleaseScope one_wait;
        for (qm_stream::Slot &sl : s->slots) free_slot(sl);
    }
    if (s->copy_stream) park_stream(s->e->device, s->copy_stream);
    s->copy_stream = nullptr;
    s->order.clear();
}

// H2D, the launch and D2H of the slot being filled: enqueue only
int launch_slot(qm_stream *s) {
    qm_engine *e = s->e;
    qm_stream::Slot &sl = s->slots[s->fill_slot];
    const int n = s->fill_n;
    // (not the row count alone -- another table of the same shape, e.g. a table_select switch or a
    // foreign load on the shared default engine, would be stacked and normalised as if it were the stream's)
    if (!e->have_lut || e->serial != s->table_serial || e->g.n_rows != s->n_rows)
        return fail("qm_stream: the engine's resident table changed under the stream (table #%llu with %d rows, "
                    "the stream was made on #%llu with %d): select the stream's table again, then push or flush",
                    (unsigned long long)(e->have_lut ? e->serial : 0), e->have_lut ? e->g.n_rows : 0,
                    (unsigned long long)s->table_serial, s->n_rows);
    const size_t kns = (size_t)s->K * s->ns;
    const size_t words = (size_t)n * step_in(s);
    const bool pull = e->cfg_stream_pull > 0 || (e->cfg_stream_pull < 0 && words * sizeof(double) <= kPullBytes);
    // ("stream_stamps" = 1, measurement: the GPU's clock before and behind every launch -- two one-thread kernels;
    // the digest goes to stderr when the stream is destroyed.  What found round 6's one-off stall: tools/diag_stream.py)
    constexpr int kStamps = 4096;
    const bool stamps = e->cfg_stream_stamps != 0 && s->launches < kStamps;   // (the first 4096 launches of a stream)
    if (stamps && !s->h_stamp) {
        QM_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_stamp), 2 * kStamps * sizeof(unsigned long long),
                             hipHostMallocDefault));
        std::memset(s->h_stamp, 0, 2 * kStamps * sizeof(unsigned long long));
    }
    if (stamps) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, e->stream, s->h_stamp + (s->launches % kStamps));
    if (pull) {
        const unsigned blocks = (unsigned)std::min<size_t>(4 * (size_t)e->n_cu, (words / 2 + 255) / 256 + 1);
        hipLaunchKernelGGL(pull_kernel, dim3(blocks), dim3(256), 0, e->stream,
                           reinterpret_cast<const pull2 *>(sl.h_on), reinterpret_cast<pull2 *>(sl.d_on),
                           words / 2, (const double *)sl.h_on, sl.d_on, words);
        QM_HIP(hipGetLastError());
    } else {
        if (!s->copy_stream && acquire_stream(e->device, &s->copy_stream) != hipSuccess)
            return fail("qm_stream: no HIP stream for the input copies");
        QM_HIP(hipMemcpyAsync(sl.d_on, sl.h_on, words * sizeof(double), hipMemcpyHostToDevice, s->copy_stream));
        QM_HIP(hipEventRecord(sl.copied, s->copy_stream));
        QM_HIP(hipStreamWaitEvent(e->stream, sl.copied, 0));
    }
    s->pulled = pull;
    // The results are a few KB per timestep: the combine kernel writes them STRAIGHT into the slot's
    // pinned host buffer (pinned memory is device-accessible under one address; the event behind the
    // launch makes them visible to the host) -- no D2H copy command, no third stream.  (With a copy on a
    // download stream of its own, 8 timesteps per launch ran at x1.03-1.09 of the resident step on the
    // example-sized grids; now x1.00, profiles/r05_ab_runs.txt.)
    double *out = sl.h_out;
    if (qm_engine_detect_batch(e, sl.d_on, 1, n, s->T, s->fsmp, s->lsmp, s->available, s->n_nodes_total,
                               out, out + kns, reinterpret_cast<int64_t *>(out + 2 * kns), 1))
        return 1;
    if (stamps)
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, e->stream, s->h_stamp + kStamps + (s->launches % kStamps));
    QM_HIP(hipEventRecord(sl.done, e->stream));
    sl.n = n;
    sl.taken = 0;
    sl.in_flight = true;
    s->order.push_back(s->fill_slot);
    s->launched_steps += n;
    ++s->launches;
    s->fill_slot = (s->fill_slot + 1) % s->depth;
    s->fill_n = 0;
    return 0;
}

}  // namespace

void streams_orphan(qm_engine *e) {
    for (qm_stream *s : e->streams) {
        release_stream(s);
        s->e = nullptr;
    }
    e->streams.clear();
}

extern "C" {

int qm_stream_create(qm_engine *e, int32_t t_samples, int32_t fsmp, int32_t lsmp, int32_t available,
                     int64_t n_nodes_total, int32_t steps_per_launch, int32_t depth, qm_stream **out) {
    if (!e || !out) return fail("qm_stream_create: NULL argument");
    *out = nullptr;
    if (steps_per_launch < 1 || steps_per_launch > 4096)
        return fail("qm_stream_create: steps_per_launch must be in 1..4096 (got %d)", steps_per_launch);
    if (depth < 2 || depth > 64) return fail("qm_stream_create: depth must be in 2..64 (got %d)", depth);
    int ns = 0;
    if (check_step(e, t_samples, fsmp, lsmp, available, &ns)) return 1;
    if ((int64_t)steps_per_launch * ns >= INT32_MAX) return fail("qm_stream_create: too many samples per launch");
    DeviceGuard guard(e->device);
    qm_stream *s = new qm_stream();
    s->e = e;
    s->n_rows = e->g.n_rows;
    s->table_serial = e->serial;
    s->T = t_samples; s->fsmp = fsmp; s->lsmp = lsmp; s->available = available; s->ns = ns;
    s->K = steps_per_launch; s->depth = depth;
    s->n_nodes_total = n_nodes_total > 0 ? n_nodes_total : e->n_nodes;
    auto bail = [&](int rc) {
        qm_stream_destroy(s);
        return rc;
    };
    // (the copy stream is taken at the first launch that copies: slots that are pulled never need one)
    s->slots.resize((size_t)depth);
    const size_t in_bytes = (size_t)s->K * step_in(s) * sizeof(double);
    const size_t out_bytes = 3 * (size_t)s->K * ns * sizeof(double);
    for (qm_stream::Slot &sl : s->slots) {
        hipError_t r = hipHostMalloc(reinterpret_cast<void **>(&sl.h_on), in_bytes, hipHostMallocDefault);
        if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&sl.h_out), out_bytes, hipHostMallocDefault);
        if (r == hipSuccess) r = pool_alloc(reinterpret_cast<void **>(&sl.d_on), in_bytes);
        for (hipEvent_t *ev : {&sl.copied, &sl.done})
            if (r == hipSuccess) r = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (r != hipSuccess)
            return bail(fail("qm_stream_create: %s (%d steps of %zu bytes per slot, %d slots)",
                             hipGetErrorString(r), s->K, step_in(s) * sizeof(double), depth));
    }
    e->streams.push_back(s);
    *out = s;
    return 0;
}

void qm_stream_destroy(qm_stream *s) {
    if (!s) return;
    if (s->e) {                                         // (else: orphaned by qm_engine_destroy)
        DeviceGuard guard(s->e->device);
        release_stream(s);
        auto &list = s->e->streams;
        list.erase(std::remove(list.begin(), list.end(), s), list.end());
    }
    delete s;
}

int qm_stream_push(qm_stream *s, const double *log_onsets) {
    if (!log_onsets) return fail("qm_stream_push: NULL argument");
    if (alive(s, "qm_stream_push")) return 1;
    DeviceGuard guard(s->e->device);
    // A full slot whose launch FAILED (the table changed under the stream, no memory, a refused step) is still
    // waiting to go out: it goes first, or this call fails as that one did -- never a copy past the slot's
    // K timesteps (the next push used to write one timestep behind the pinned buffer and launch
    // K + 1 of them).
    if (s->fill_n >= s->K && launch_slot(s)) return 1;
    qm_stream::Slot &sl = s->slots[s->fill_slot];
    if (s->fill_n == 0 && sl.in_flight) {
        (void)fail("qm_stream_push: all %d slots hold results that have not been popped", s->depth);
        return 2;
    }
    // (the slot's previous H2D has finished: its launch's results were popped)
    host_copy(sl.h_on + (size_t)s->fill_n * step_in(s), log_onsets, step_in(s) * sizeof(double));
    if (++s->fill_n < s->K) return 0;
    return launch_slot(s);
}

int qm_stream_flush(qm_stream *s) {
    if (alive(s, "qm_stream_flush")) return 1;
    if (s->fill_n == 0) return 0;
    DeviceGuard guard(s->e->device);
    return launch_slot(s);
}

int qm_stream_pop(qm_stream *s, int32_t n_steps, double *max_coa, double *max_norm_coa,
                  int64_t *max_coa_idx) {
    if (!max_coa || !max_norm_coa || !max_coa_idx) return fail("qm_stream_pop: NULL argument");
    if (alive(s, "qm_stream_pop")) return 1;
    if (n_steps < 0 || n_steps > s->launched_steps - s->popped_steps)
        return fail("qm_stream_pop: %d steps asked for, %lld launched and not yet popped (%d pushed into "
                    "a launch that has not gone out: qm_stream_flush)", n_steps,
                    (long long)(s->launched_steps - s->popped_steps), s->fill_n);
    DeviceGuard guard(s->e->device);
    const size_t ns = (size_t)s->ns, kns = (size_t)s->K * ns;
    for (int k = 0; k < n_steps;) {
        qm_stream::Slot &sl = s->slots[s->order.front()];
        QM_HIP(hipEventSynchronize(sl.done));
        const int take = std::min(n_steps - k, sl.n - sl.taken);
        const double *src = sl.h_out + (size_t)sl.taken * ns;
        std::memcpy(max_coa + (size_t)k * ns, src, (size_t)take * ns * sizeof(double));
        std::memcpy(max_norm_coa + (size_t)k * ns, src + kns, (size_t)take * ns * sizeof(double));
        std::memcpy(max_coa_idx + (size_t)k * ns, src + 2 * kns, (size_t)take * ns * sizeof(int64_t));
        k += take;
        sl.taken += take;
        s->popped_steps += take;
        if (sl.taken == sl.n) {
            sl.in_flight = false;
            s->order.pop_front();
        }
    }
    return 0;
}

int qm_stream_pending(qm_stream *s, int32_t *launched_not_popped, int32_t *pushed_not_launched) {
    if (alive(s, "qm_stream_pending")) return 1;
    if (launched_not_popped) *launched_not_popped = (int32_t)(s->launched_steps - s->popped_steps);
    if (pushed_not_launched) *pushed_not_launched = s->fill_n;
    return 0;
}

}  // extern "C"
