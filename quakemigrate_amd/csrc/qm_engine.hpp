// qm_engine.hpp -- internal header of the engine's host side (not installed; the C ABI is
// include/qmhip.h).  The host runtime is split by what it deals with:
//   qm_runtime.hip   error text, pooled streams, the pinned bounce buffers, the device-memory pool
//   qm_tables.hip    everything derived from ONE travel-time table: load, layout search, the kernels'
//                    derived tables (round-2 offsets, paired, shift-reuse, screening), parked tables,
//                    on-device serving
//   qm_engine.hip    engine handle, tunables, the stacking launches and the step entry points
//                    (detect / detect_batch / partial / finalize / migrate / marginal / find_max_coa)
//   qm_screen.hip    the opt-in screened detect's launch sequence
//   qm_stream.hip    the continuous detect pipeline (pinned ring, copies overlapped with compute)
//   qm_widen.hip     the rows next to the path: onset stage, locate fits, RBF peak
//   qm_compat.hip    the five reference-signature symbols (qmlib.h:28-44)
// Everything declared here lives in the library only (hidden visibility).
#pragma once
#include "../../include/qmhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "qm_kernels.hpp"
#include "qm_launch.hpp"
#include "qm_screen.hpp"
#include "qm_pair.hpp"
#include "qm_shift.hpp"
#include "qm_ties.hpp"

#pragma GCC visibility push(hidden)

// ---- qm_runtime.hip ---------------------------------------------------------------------------
int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();
const char *error_text();

#define QM_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t err__ = (call);                                                           \
        if (err__ != hipSuccess)                                                             \
            return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__,   \
                        __LINE__);                                                           \
    } while (0)
// HIP status of a launcher of qm_launch.hpp -> the same error convention
#define QM_TABLE(call) QM_HIP(call)

hipError_t acquire_stream(int device, hipStream_t *out);
void park_stream(int device, hipStream_t s);
hipError_t copy_back(void *dst, const void *src, size_t bytes, hipStream_t s);
hipError_t copy_in(void *dst, const void *src, size_t bytes, hipStream_t s);
// `pieces` consecutive device pieces of `bytes` each to `pieces` separate host destinations: one DMA
// where they fit a half together
hipError_t copy_back_pieces(void *const *dst, const void *src, int pieces, size_t bytes, hipStream_t s);
hipError_t copy_back_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width,
                        size_t height, hipStream_t s);
hipError_t copy_in_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width,
                      size_t height, hipStream_t s);
void host_copy(void *dst, const void *src, size_t n);
hipError_t pool_alloc(void **out, size_t bytes);
void pool_free(void *p);
void pool_release_all_idle();
// buffers released inside a scope are parked behind ONE device-wide wait when the outermost ends
struct PoolReleaseScope {
    PoolReleaseScope();
    ~PoolReleaseScope();
    PoolReleaseScope(const PoolReleaseScope &) = delete;
    PoolReleaseScope &operator=(const PoolReleaseScope &) = delete;
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int ensure(size_t count) {
        if (count <= n) return 0;
        if (p) pool_free(p);
        p = nullptr;
        n = 0;
        QM_HIP(pool_alloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
        n = count;
        return 0;
    }
    void release() {
        if (p) pool_free(p);
        p = nullptr;
        n = 0;
    }
};

// One shift-reuse layout of a table (qm_shift.hpp): brick grid with even brick dimensions, the row-window
// records of every brick, which bricks fit, the record stream dealt over the workgroup's wavefronts.  A table
// has up to two: `sh`, what rounds 3-5 built (256-sample tiles; any launch kind), and -- round 6 -- `shw` for
// the fused detect where WIDE tiles fit: the 8-wave shape on its own brick grid, with a second set of records
// and a second stream for the 384-sample tiles beside those of the 256-sample and tail tiles behind them.
struct ShiftLayout {
    int nw = 0;                             // workgroup shape the tables were built for
    qm::GridDesc g{};
    DevBuf<int32_t> raw, meta, total, fit, list;   // list: the bricks that do not fit (direct kernel)
    DevBuf<uint32_t> stream;
    int n_list = 0, rows2 = 0;
    int nblk = 1, sb = 0;                   // row blocks (tables of more than 64 rows): blocks, rows per block
    int stage_slots = 0, stage_reach = 0;   // ... largest row window (slots), furthest sample it holds
    bool direct = false;                    // ... staged by LDS-direct loads (stack_shift_rows2_kernel)
    bool quad = false;                      // ... by two 4-wave workgroups per CU (stack_shift_rows4_kernel)
    bool built = false, ok = false;
    int64_t quads = 0, group_rows = 0;      // register-window quads fetched / (group, row)s
    // wide tiles (shw only)
    bool wide = false;
    DevBuf<int32_t> wmeta, wtotal;
    DevBuf<uint32_t> wstream;
    int64_t wquads = 0;                     // quads fetched by the wide tiles' windows (per 48 adds, not 32)

    void release() {
        PoolReleaseScope one_wait;
        raw.release(); meta.release(); total.release(); fit.release(); list.release(); stream.release();
        wmeta.release(); wtotal.release(); wstream.release();
    }
    size_t device_words() const {
        return raw.n + meta.n + total.n + fit.n + list.n + stream.n + wmeta.n + wtotal.n + wstream.n;
    }
};

// Everything that is derived from ONE travel-time table: the table itself, its brick records and
// window offsets, the layouts of the paired / screened / shift-reuse kernels built from it on first
// use, and the launch shape the table's layout search picked.  The engine works on the state it
// inherits; qm_engine_table_select parks it in a slot and brings another one in (a swap of pointers:
// no device work), so that a change of station availability -- a different served table,
// lut.py:529-537 -- costs a rebuild only the first time that table is seen.
struct TableState {
    bool have_lut = false;
    uint64_t serial = 0;            // identity of the loaded table (process-unique; travels with the state
                                    // through qm_engine_table_select): what a qm_stream checks before a launch
    qm::GridDesc g{};
    int64_t n_nodes = 0;
    int64_t node_offset = 0;
    int32_t lut_max = 0;
    int n_rows_hint = 0;            // row count the automatic choice is based on
    int auto_j = 0;                 // samples per lane picked by the table's layout search (> 64 rows)
    int tab_waves = 0, tab_lds_bytes = 0;   // workgroup shape the layout search picked (0: none yet)
    DevBuf<int32_t> d_lut, d_bmeta, d_btotal, d_wide;
    DevBuf<uint16_t> d_rel;
    bool rel_built = false;         // d_rel holds this table's offsets (built on first use)
    std::vector<int32_t> h_btotal;
    int n_wide = 0;
    int plan_j = -1, plan_cap = -1;

    // float32 screening (qm_screen.hpp): staggered-copy offset table
    DevBuf<int32_t> d_smeta, d_smeta_raw, d_stotal, d_swide;
    qm::GridDesc sg{};                      // the sweep's own brick grid
    DevBuf<uint16_t> d_srel;
    int n_swide = 0;
    int screen_kt = 0, screen_wb = 0;       // what the screening table was built for

    // paired (16-byte operand) layout of the float64 kernel (qm_pair.hpp): own brick grid
    qm::GridDesc pg{};
    DevBuf<int32_t> d_pmeta, d_pmeta_raw, d_ptotal, d_pwide;
    DevBuf<uint16_t> d_prel;
    int n_pwide = 0;
    int pair_kt = 0;                        // tile length the paired tables were built for
    bool pair_ok = false;                   // ... and whether (almost) every brick fits

    // shift-reuse layouts (qm_shift.hpp): own brick grids, row-window slots, record streams
    ShiftLayout sh, shw;

    void release_all() {
        PoolReleaseScope one_wait;
        d_lut.release(); d_bmeta.release(); d_btotal.release(); d_wide.release(); d_rel.release();
        d_smeta.release(); d_smeta_raw.release(); d_stotal.release(); d_swide.release(); d_srel.release();
        d_pmeta.release(); d_pmeta_raw.release(); d_ptotal.release(); d_pwide.release(); d_prel.release();
        sh.release(); shw.release();
    }
    size_t device_bytes() const {
        return (d_lut.n + d_bmeta.n + d_btotal.n + d_wide.n + d_smeta.n + d_smeta_raw.n + d_stotal.n +
                d_swide.n + d_pmeta.n + d_pmeta_raw.n + d_ptotal.n + d_pwide.n + sh.device_words() +
                shw.device_words()) * 4 +
               (d_rel.n + d_srel.n + d_prel.n) * 2;
    }
};

struct TableSlot {
    TableState state;
    uint64_t key = 0;
    uint64_t stamp = 0;             // last use (the engine's table clock): the oldest slot is evicted
    bool used = false;
};

struct qm_stream;                    // a continuous-detect pipeline on an engine (qm_stream.hip)

struct qm_engine : TableState {
    int device = 0;
    std::vector<qm_stream *> streams;   // pipelines alive on this engine: orphaned when it is destroyed
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // optional per-call timing log (bench): pairs of events around every stacking launch
    bool log_timing = false;
    std::vector<hipEvent_t> ev_log;     // 2 events per recorded call
    size_t ev_used = 0;

    // parked tables (qm_engine_table_select) and the key of the one being worked on
    std::vector<TableSlot> slots;
    uint64_t cur_key = 0, table_clock = 0;
    bool cur_keyed = false;
    int64_t table_hits = 0, table_misses = 0, table_evictions = 0;

    // tunables
    int cfg_bx = 0, cfg_by = 0, cfg_bz = 0;      // 0 = choose the brick shape per table
    int cfg_j = 0;                  // samples per lane (time tile = 64*J); 0 = by table width
    int cfg_waves = 8;
    bool user_waves = false, user_lds = false;   // set explicitly: no automatic layout
    int cfg_groups = 0;
    int cfg_rounds = 12;            // automatic group count: grid = this many rounds over the slots
    bool user_rounds = false;       // ... set explicitly
    int cfg_lds_bytes = 80 * 1024;
    int cfg_force_direct = 0;
    int cfg_generic = 0;            // 1 = always the generic (any row count) LDS kernel
    int cfg_scan_waves = 32;        // find_max_coa of a volume: wavefronts per CU over the whole grid
    int cfg_exact = 1;              // 1 = the exact-row-count kernel where one is built (see
                                    //     qm_launch.hpp), 0 = the chunked kernels only
    int64_t cfg_chunk_bytes = (int64_t)4 << 30;
    int cfg_pair = 1;               // 1 = the 16-byte-operand kernel (qm_pair.hpp) where it applies
    int cfg_screen = 0;             // 1 (opt-in): detect = float32 screening sweep + exact float64
                                    // refinement (qm_screen.hpp); 0: every node-sample in float64
    int cfg_screen_pairs = 0;       // pairs of samples per lane in the sweep (0 = automatic)
    int cfg_screen_brick16 = 0;     // also try 16x8x8 bricks for the sweep
    int cfg_screen_big = -1;        // 1: one 16-wave workgroup per CU with 160 KB of LDS; -1 = automatic
    int cfg_shift = -1;                     // -1: where the table qualifies, 0: never, 1: as -1 (explicit)
    int cfg_shift_waves = 0;                // workgroup shape: 4 (two per CU), 12 (one per CU), 0 = automatic
    int cfg_shift_lazy = -1;                // detect loop flavour: -1 automatic, 0 eager, 1 lazy arg-max
    int cfg_shift_tail = 1;                 // 1: a scan's remainder of <= 192 samples runs as one tail tile of
                                            // 64 / 128 / 192 samples; 0: whole tiles only (round 3)
    int cfg_shift_rows_direct = 1;
    int cfg_shift_wide = -1;                // fused detect on WIDE tiles (384 samples, six per lane; round 6): -1 where
                                            // they fit and the scan holds at least one, 0 never, 1 as -1 (explicit)
    int cfg_shift_wide_rows = 1;            // ... on ROW BLOCKS where the windows of all rows do not fit (0: never,
                                            // 2: row blocks whatever fits -- tests)
    int cfg_stream_pull = -1;               // qm_stream: a slot's pinned inputs pulled by a kernel on the engine's stream
                                            // instead of a copy command on another (-1: slots of <= 1 MB, 0, 1)
    int cfg_stream_stamps = 0;              // qm_stream, measurement: GPU-clock stamps around every launch (stderr digest)
    int cfg_tie_rule = 0;                   // 0: largest float64 sum, lowest index among equal ones (default);
                                            // 1: the reference's rule on near-ties (qm_ties.hpp)
    int cfg_tie_sets = 1;                   // ... refined from a partial set PER BRICK where the stacking kernel has
                                            // that flavour (the shift-reuse fused detect); 0: round 5's sets of
                                            // four bricks everywhere (measurements)

    // per-step scratch of the screened detect (qm_screen.hpp) and its statistics
    DevBuf<int32_t> d_scalar, d_counts, d_cells, d_work, d_flags;
    DevBuf<int32_t> d_onq, d_cell, d_gmax, d_pm, d_sparams;
    DevBuf<double> d_rowmax, d_ssum, d_cand_z;
    DevBuf<int64_t> d_cand_idx;
    int64_t screened_steps = 0, fallback_steps = 0, last_candidates = 0;
    int last_plan_jp = 0, last_plan_big = 0;
    int last_kernel = 0, last_j = 0;        // stacking kernel of the last launch: 0 chunked, 1 exact-row-count, 2 paired
    int32_t *h_flags = nullptr;             // pinned ring of per-step (flags, candidates) pairs
    int flags_pending = 0, flags_head = 0;  // not yet folded into the counters
    int shift_lazy_last = 0;                // loop flavour the last shift-reuse launch took
    int shift_tail_last = 0;                // samples per lane of the last launch's tail tile (0: none)
    int shift_wide_last = 0;                // wide tiles of the last shift-reuse launch
    int last_batched = 1;                   // timesteps the last detect_batch put into one launch
    // which bricks the partial sets of the last stacking launch stand for (qm_ties.hpp)
    qm::GridDesc last_g{};
    int last_groups_lds = 0, last_groups_direct = 0, last_n_list = 0;
    int last_brick_rows = 0;                            // (a row of maxima per brick besides: StackArgs::brick_max)
    DevBuf<double> d_bmax;
    int last_sets = 0, last_scan_n = 0;                 // their number, the samples each one spans
    bool last_sets_own = false;                         // ... left by a float64 detect (not by the screened sweep)
    DevBuf<double> d_tie_zext;                          // the largest z per sample, left by the combine of the own sets
    DevBuf<double> d_tie_zgrid;                         // sharded detects: the GRID's largest z per sample
    const int32_t *last_list = nullptr;
    DevBuf<double> d_tie_z;
    DevBuf<int32_t> d_tie_pairs, d_tie_imin, d_tie_count, d_tie_cands;
    DevBuf<unsigned long long> d_tie_emax, d_tie_keys;
    int64_t tie_refined_steps = 0, tie_overflow_last = 0, tie_pairs_last = 0;
    bool tie_counts_pending = false;        // the last refinement's counters are still on the device

    // float64 travel-time grids in seconds (optional; on-device table serving)
    DevBuf<double> d_grids;
    DevBuf<int32_t> d_rows, d_served;
    int gx = 0, gy = 0, gz = 0, g_rows = 0;

    // onset stage scratch
    DevBuf<double> d_sig, d_sta, d_lta, d_raw;
    DevBuf<int32_t> d_onset_meta;

    // scratch
    DevBuf<double> d_onsets, d_pmax, d_psum, d_out_a, d_chunk, d_marg, d_marg_out;
    int marg_tiles = 0;             // time tiles of the last marginal-map launch (rows of d_marg)
    DevBuf<int64_t> d_pidx;
    // locate fits: three map-sized work buffers, reduction partials, device-side scalars
    DevBuf<double> d_fit_a, d_fit_b, d_fit_c, d_fit_part, d_fit_val, d_fit_win;
    DevBuf<int64_t> d_fit_pidx;
};

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// pairs of samples per lane: time tile = 128 * JP; 0 = this table is not screened
// How the sweep is launched: JP pairs of samples per lane (time tile 128*JP) and either two
// 8-wave workgroups per CU with 80 KB of LDS each, or ("big") one 16-wave workgroup with all
// 160 KB -- twice the tile for the same rows, so fewer address / epilogue instructions per sample.
struct ScreenPlan {
    int jp = 0;                     // 0 = this table is not screened
    bool big = false;
    int kt() const { return 128 * jp; }
    int lds_bytes(const qm_engine *e) const {
        return big ? 160 * 1024 : (e->user_lds ? e->cfg_lds_bytes : 80 * 1024);
    }
    int window_bytes(const qm_engine *e) const {       // minus the cell-maximum row
        return (lds_bytes(e) - kt() * 4) / 16 * 16;
    }
    int threads() const { return big ? 1024 : 512; }
};

// where the kernels write the three series; copies back afterwards if the caller is on host
struct OutStage {
    double *a, *b;
    int64_t *i;
};

// ---- qm_tables.hip ------------------------------------------------------------------------------
int lds_cap_doubles(const qm_engine *e);
int ensure_rel(qm_engine *e);
int eff_j(const qm_engine *e);
int run_j(const qm_engine *e, int n_chunk);
int plan_wide(qm_engine *e, int J);
int pair_jp(const qm_engine *e, int n_chunk, bool volume);
int ensure_pair_tables(qm_engine *e, int jp);
int ensure_shift_tables(qm_engine *e, ShiftLayout &L);
bool screen_plan_feasible(const qm_engine *e, int S, const ScreenPlan &p);
ScreenPlan screen_plan(const qm_engine *e, int S, int n_samples);
int ensure_screen_tables(qm_engine *e, const ScreenPlan &plan);

// ---- qm_engine.hip ------------------------------------------------------------------------------
int auto_groups(const qm_engine *e, int ntiles, int units, int blocks_per_cu, int rounds = 0);
int run_stack(qm_engine *e, const double *d_onsets, int T, int fsmp, int n_samples, int available,
              int sample0, int n_chunk, double *volume, int64_t vol_stride, int accumulate,
              bool want_scan, int *n_sets, bool marginal = false, int m0 = 0, int m1 = 0,
              const int32_t *run_if = nullptr, int n_steps = 1, int64_t step_stride = 0,
              bool *batched = nullptr);
int combine(qm_engine *e, const double *pmax, const int64_t *pidx, const double *psum, int sets,
            int n, int mode, int64_t node_offset, int64_t n_nodes_total, double *o_max,
            double *o_second, int64_t *o_idx, const int32_t *run_if = nullptr,
            int64_t set_stride = 0);
int refine_ties(qm_engine *e, const double *d_on, int T, int fsmp, int available, int sample0,
                int n_chunk, int sets, int64_t *o_idx, int n_steps = 1, int64_t step_stride = 0,
                const double *zext = nullptr, unsigned long long *o_key = nullptr);
int detect_core(qm_engine *e, const double *d_on, int T, int fsmp, int ns, int available, int mode,
                int64_t n_nodes_total, double *o_max, double *o_second, int64_t *o_idx);
int check_step(qm_engine *e, int T, int fsmp, int lsmp, int available, int *n_samples);
int stage_onsets(qm_engine *e, const double *onsets, int on_device, int T, const double **out);
int stage_out(qm_engine *e, int n, int out_on_device, double *max_coa, double *max_norm,
              int64_t *idx, OutStage *st);
int fetch_out(qm_engine *e, int n, int out_on_device, const OutStage &st, double *max_coa,
              double *max_norm, int64_t *idx);

// ---- qm_stream.hip ------------------------------------------------------------------------------
// the engine is going away: its pipelines give their buffers back and refuse further calls
void streams_orphan(qm_engine *e);

// ---- qm_screen.hip ------------------------------------------------------------------------------
constexpr int kFlagRing = 1024;
int drain_flags(qm_engine *e);
int run_screen(qm_engine *e, const double *d_onsets, int T, int fsmp, int ns, int available,
               int *n_sets, bool *screened);

#pragma GCC visibility pop
