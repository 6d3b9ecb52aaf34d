/*
 * qmhip.h -- C ABI of the MI355X (gfx950) coalescence-migration engine.
 *
 * The shared library (quakemigrate_amd/csrc/libqmhip.so, also installed under
 * the reference's own name qmlib<EXT_SUFFIX>) is the drop-in boundary for the
 * hot path of QuakeMigrate: quakemigrate.core's `migrate` + `find_max_coa`.
 * Signatures use plain pointers and sizes only (no torch / HIP types).
 *
 * Part 1 are the five symbols the reference binds at import
 * (quakemigrate/core/lib.py:38-49,128,173,211,249; prototypes
 * quakemigrate/core/src/qmlib.h:28-44, export list qmlib.def:3-7) with
 * identical signatures, so the UNMODIFIED reference front-end runs on this
 * library.  Part 2 is the handle API (resident travel-time table, fused
 * stack + exp + scan that never materialises the 4-D volume) that this
 * repository's own host side (quakemigrate_amd/) drives.  Part 3 is the
 * continuous detect sweep as a pipeline (pinned ring, copies overlapped with
 * compute) -- the loop of QuakeScan._continuous_compute around part 2's call.
 *
 * Conventions: every function of part 2 returns 0 on success, non-zero on
 * failure; qm_last_error() returns a thread-local message.  `*_on_device`
 * flags say whether a pointer is host memory (synchronous, copied internally)
 * or device memory on the engine's GPU (asynchronous on the engine stream).
 * Host buffers may be any memory, pageable included: inputs and results are moved
 * through a pinned buffer of the library's own and copied with the CPU, the
 * caller's pointers are never handed to the HIP runtime.
 * An engine's private stream and its device memory come from process-wide pools
 * and return to them (qm_release_cached_memory()).
 * Layouts are the reference's: onsets f64 [n_rows][T] (already log(clip)),
 * travel-times i32 [nx][ny][nz][n_rows] (C order, flat node = (ix*ny+iy)*nz+iz,
 * quakemigrate/lut/lut.py:165-166), volume f64 [n_nodes][n_samples].
 */
#ifndef QMHIP_H
#define QMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ part 1 */
/* reference-compatible symbols (host pointers, synchronous, `threads` ignored) */

/* replaces migrate(), quakemigrate/core/src/migratelib.c:40-65 (qmlib.h:28-29).
 * map4d is accumulated INTO (the reference's `+=`), then exp(./available). */
void migrate(double *onsets, int32_t *lookup_tables, double *map4d,
             int32_t fsmp, int32_t lsmp, int32_t n_samples, int32_t n_stations,
             int32_t available, int64_t n_nodes, int64_t threads);

/* replaces find_max_coa(), migratelib.c:85-111 (qmlib.h:31-32). */
void find_max_coa(double *map4d, double *max_coa, double *max_norm_coa,
                  int64_t *max_coa_idx, int32_t n_samples, int64_t n_nodes,
                  int64_t threads);

/* The two symbols above return void, like the reference's.  If one of them fails (no HIP device,
 * a travel time beyond the post-pad, ...) it prints the reason to stderr, fills its outputs with
 * NaN (indices with 0) and leaves a non-zero status here until the next call (text:
 * qm_last_error()); QM_HIP_COMPAT_ON_ERROR=abort in the environment aborts the process instead.
 * The travel-time table is kept resident between calls and re-uploaded only when its content or
 * shape changes.  "Content" = two independent 64-bit hashes of every word (qm_table_hash): a stale
 * table would need both to collide at once; callers who will not accept even that set
 * QM_HIP_COMPAT_REUPLOAD=1 (upload on every call, the reference's cost model).
 * QM_HIP_GRID=nx,ny,nz tells migrate() the grid shape its signature cannot carry (nx*ny*nz must be
 * n_nodes): the table is then bricked in 3-D (8x8x8 where it fits) instead of 1x1x32 along the
 * flat index.  QM_HIP_ASSUME_ZERO_MAP=1 skips the scan of map4d for non-zero content (the
 * reference's binding always passes zeros, lib.py:101). */
int qm_compat_status(void);
/* the two content hashes the drop-in migrate() keys its resident table on (host code) */
void qm_table_hash(const int32_t *table, int64_t n_words, uint64_t *hash_a, uint64_t *hash_b);

typedef struct {
    int n;
    int nsta;
    int nlta;
} stalta_header; /* qmlib.h:34-38, numpy mirror lib.py:33-36 */

/* onsetlib.c:35-59, :79-108, :126-148.  Upstream of the hot path and serial
 * O(n) per trace in the reference; host code here too (must be exported
 * because lib.py binds them at import). */
void overlapping_sta_lta(const double *signal, const stalta_header *head,
                         double *onset);
void centred_sta_lta(const double *signal, const stalta_header *head,
                     double *onset);
void recursive_sta_lta(const double *signal, const stalta_header *head,
                       double *onset);

/* ------------------------------------------------------------------ part 2 */
typedef struct qm_engine qm_engine;

const char *qm_last_error(void);
/* number of HIP devices visible, or -1 */
int qm_device_count(void);
/* What the library was built from: the constants and source digest of the generated shift-reuse loops and
 * the development defines of the kernels' unit ("overlay=none", "defines=none" in the product build;
 * tools/shift_variants.sh makes the others).  bench.py prints it, the GPU suite asserts it. */
const char *qm_build_info(void);
/* Device memory released by engines (destroyed engines, replaced tables) is parked in the process
 * and reused by the next request of its size (up to QM_HIP_POOL_KEEP_MB, default 2 GB, stay parked per device); this returns all of it to
 * the driver.  Always 0. */
int qm_release_cached_memory(void);

int qm_engine_create(int device_id, qm_engine **out);
void qm_engine_destroy(qm_engine *e);

/* run all asynchronous work on this hipStream_t (e.g. torch's current stream; a
 * NULL handle is the device's default stream), or, with use_own != 0, on the
 * engine's private non-blocking stream (the initial state). */
int qm_engine_set_stream(qm_engine *e, void *hip_stream, int use_own);
int qm_engine_synchronize(qm_engine *e);

/* Tunables (qm_engine_config) and read-outs (qm_engine_get).  Nothing has to be set: every default is
 * the automatic choice, every setting gives the same max_coa / max_coa_idx bits
 * and the same stored values unless the row says otherwise.  Layout keys take effect at the next
 * qm_engine_load_lut / qm_engine_serve.
 *
 * key (config)          values [default]         meaning
 * --------------------  -----------------------  -------------------------------------------------------
 * -- precision / rule (opt-in; the only keys that change results) --
 * screen                0 / 1 [0]                1: screened detect (qm_screen.hpp): exact-integer sweep over
 *                                                every node-sample + float64 re-evaluation of every cell that
 *                                                can hold the maximum; max_coa, max_coa_idx identical,
 *                                                max_norm_coa within 6.7e-7 (checked bound, else redone in f64)
 * tie_rule              0 / 1 [0]                0: largest float64 sum, lowest flat index among equal sums;
 *                                                1: the reference's rule on near-ties (csrc/qm_ties.hpp): sums
 *                                                within two ulps of a sample's largest compared on a correctly
 *                                                rounded exp(sum / available), lowest index among equal values
 *                                                (migratelib.c:98-105 as its scalar-libm build computes it);
 *                                                final series of detect / detect_batch (the step axis kept) /
 *                                                migrate / marginal, and sharded detects through
 *                                                qm_engine_tie_partial / _tie_fold; values unchanged.  Device
 *                                                memory: 8 bytes per brick (512-1024 nodes) and scanned sample
 *                                                besides the partial sets (C3: 227 MB per timestep of a launch)
 * tie_sets              0 / 1 [1]                (measurements) 1: with tie_rule = 1 the shift-reuse fused detect
 *                                                publishes a partial set per brick, the refinement re-stacks one
 *                                                brick per sample; 0: sets of four bricks from more workgroups
 * -- layout of the round-2 kernels (qm_kernels.hpp, qm_pair.hpp) --
 * brick_x, _y, _z       0..64 [0 = automatic]    node-brick shape; automatic = the largest of 8x8x8 .. 1x1x1
 *                                                whose windows fit LDS for >= 99.5 % of the bricks
 * samples_per_lane      0, 1, 2, 4 [0]           time tile = 64 x this; 0 = by table width and scan length
 * waves                 1..16 [by table]         wavefronts per workgroup
 * lds_bytes             1 KiB..160 KiB [by table] window budget per workgroup
 * groups                >= 0 [0 = automatic]     brick groups per time tile (= partial sets)
 * rounds                1..1024 [12; 3 for       grid size of the automatic group count, in rounds over the
 *                       bricks of <= 64 nodes]   resident workgroup slots
 * exact                 0 / 1 [1]                the exact-row-count kernels where one is built
 * pair                  0, 1, 2 [1]              16-byte-operand kernel: 1 = volume-writing launches the
 *                                                shift-reuse kernel does not take, 2 = every launch, 0 = never
 * generic, force_direct 0 / 1 [0]                the any-row-count LDS kernel / the direct (no LDS) kernel
 *                                                for everything: cross-checks
 * scan_waves            1..4096 [32]             find_max_coa of a volume: wavefronts per CU
 * chunk_bytes           >= 1 MiB [4 GiB]         device chunk of a HOST volume (migrate / find_max_coa)
 * -- the shift-reuse kernel (qm_shift.hpp) --
 * shift                 -1, 0, 1 [-1]            -1: where the table qualifies (every 2x2x2 group's delay
 *                                                spread <= 20 samples for >= 99.5 % of the bricks, no grid
 *                                                dimension of 1); 0: never (round-2 kernels); 1: as -1, also
 *                                                on grids one node thick
 * shift_waves           0, 4, 8, 12 [0]          workgroup shape: automatic = two 4-wave workgroups per CU up to
 *                                                ~32 rows, one 8-wave workgroup (all 160 KB) to 64 rows; 12 =
 *                                                running state in LDS (measured no faster)
 * shift_lazy            -1, 0, 1 [-1]            detect loop flavour: lazy arg-max recovery from 160 groups per
 *                                                wavefront on (max_norm_coa within 1e-15 of the eager one)
 * shift_tail            0 / 1 [1]                a scan's remainder of <= 192 samples as ONE tail tile of
 *                                                64 / 128 / 192 samples; 0 = whole 256-sample tiles only
 * shift_wide            -1, 0, 1 [-1]            fused detect on WIDE tiles (384 samples, six per lane, 8-wave
 *                                                workgroups, own brick grid) where the table's windows fit and the
 *                                                scan holds at least one; 0 = the 256-sample tiles of rounds 3-5.
 *                                                max_coa / max_coa_idx bit-equal, max_norm_coa within 1e-14 (the
 *                                                sum over the nodes is formed in another order)
 * shift_wide_rows       0, 1, 2 [1]              wide tiles on ROW BLOCKS (4x4x4 bricks, blocks of <= 20 rows through a
 *                                                double-buffered LDS) where the 384-sample windows of all rows do not
 *                                                fit a CU's LDS (BASELINE configs[3]: 60 rows); 0 = never, 2 = always
 * shift_rows_direct     0, 1, 2 [1]              tables of > 64 rows (row blocks): 1 = blocks of <= 34 rows,
 *                                                double-buffered LDS, LDS-direct loads; 0 = blocks of <= 64
 *                                                through registers; 2 = two 4-wave workgroups per CU
 * -- screened detect's launch shape --
 * screen_pairs          0, 1, 2, 4 [0]           pairs of samples per lane in the sweep
 * screen_big            -1, 0, 1 [-1]            one 16-wave workgroup with 160 KB
 * screen_brick16        0 / 1 [0]                also try 16x8x8 bricks
 * -- the continuous pipeline (part 3) --
 * stream_pull           -1, 0, 1 [-1]            a slot's pinned inputs are PULLED by a small kernel on the engine's
 *                                                stream instead of copied by a command on a second stream: -1 =
 *                                                slots of <= 1 MB (example-sized grids, few steps per launch), 1 =
 *                                                always, 0 = never; same results
 * -- measurement --
 * stream_stamps         0 / 1 [0]                qm_stream: a one-thread kernel before and behind every launch stores
 *                                                the GPU's clock; busy time per launch and the gaps between launches
 *                                                go to stderr when the stream is destroyed
 * log_timing            0 / 1 [0]                HIP events around every stacking launch (qm_engine_kernel_log)
 *
 * read-outs (qm_engine_get), besides the keys above: n_cu, n_nodes, n_rows, nx, ny, nz, n_bricks,
 * n_wide_bricks, mean_span; last_kernel (0 chunked, 1 exact-row-count, 2 paired, 3 shift-reuse),
 * last_kernel_j, steps_per_launch (timesteps the last detect_batch put into one launch);
 * shift_ok, shift_brick_nodes, shift_wide_bricks, shift_row_blocks, shift_operands_per_add_x1000,
 * shift_tail_spl; shift_wide_ok, shift_wide_tiles (of the last launch), shift_wide_brick_nodes,
 * shift_wide_direct_bricks, shift_wide_row_blocks, shift_wide_operands_per_add_x1000; pair_brick_nodes, pair_wide_bricks, pair_tile; screened_steps, fallback_steps,
 * last_candidates, screen_brick_nodes; tie_refined_steps, tie_pairs and tie_overflow_samples (of the last
 * refined launch), tie_brick_rows (rows of per-brick maxima the last stacking launch left: 0 = it refined from
 * sets of bricks);
 * table_hits, table_misses, table_evictions, tables_parked, table_bytes, tables_parked_bytes. */
int qm_engine_config(qm_engine *e, const char *key, int64_t value);
int qm_engine_get(qm_engine *e, const char *key, int64_t *value);

/* Make a travel-time table resident.  lut: i32 [nx][ny][nz][n_rows].
 * node_offset: flat index of this table's first node inside the full grid
 * (non-zero when the grid is sharded over GPUs by x-planes).  Builds the
 * per-brick window tables on the device.  Replaces the per-call
 * `lookup_tables` argument of migrate() (lib.py:112-123). */
int qm_engine_load_lut(qm_engine *e, const int32_t *lut, int lut_on_device,
                       int32_t nx, int32_t ny, int32_t nz, int32_t n_rows,
                       int64_t node_offset);
/* Several resident tables.  The reference serves a table per timestep from the stations available
 * in it (LUT.serve_traveltimes, quakemigrate/lut/lut.py:502-538; QuakeScan._compute,
 * signal/scan.py:619-634): when a station drops out and comes back, the same few tables alternate.
 * qm_engine_table_select(key) declares which table the following calls work on: the state of the
 * current one (table, brick records, window offsets, the kernels' derived layouts -- about 4x the
 * table's size) is parked under its key, at most `capacity` of them, least recently used evicted,
 * and the state parked under `key` is brought back -- a swap of pointers, no device work.
 * *resident = 1: the table is there, go on; 0: nothing is resident now, load it (qm_engine_load_lut
 * / qm_engine_serve), it is then known under `key`.  The float64 grids of the serving path and all
 * per-step scratch are shared by every table.  capacity = 0 parks nothing (one resident table, as
 * without this call).  qm_engine_get: "table_hits", "table_misses", "table_evictions",
 * "tables_parked", "table_bytes", "tables_parked_bytes". */
int qm_engine_table_select(qm_engine *e, uint64_t key, int32_t capacity, int32_t *resident);

/* On-device table serving -- replaces LUT.serve_traveltimes (quakemigrate/lut/lut.py:502-538)
 * and Grid3D.decimate (lut.py:102-140) on the host.  Upload the float64 travel-time grids (seconds,
 * [nx][ny][nz] each, one per station/phase) once; qm_engine_serve then builds and makes
 * resident the int32 table of the selected grids, rint(tt * sampling_rate) in half-to-even
 * rounding like np.rint, decimated by (dfx, dfy, dfz) exactly like the reference. */
int qm_engine_grids_begin(qm_engine *e, int32_t nx, int32_t ny, int32_t nz, int32_t n_grids);
int qm_engine_grids_set(qm_engine *e, int32_t index, const double *grid, int on_device);
int qm_engine_serve(qm_engine *e, double sampling_rate, const int32_t *rows, int32_t n_rows,
                    int32_t dfx, int32_t dfy, int32_t dfz, int64_t node_offset);
/* copy the resident int32 table ([n_nodes][n_rows]) back to the host (tests, inspection) */
int qm_engine_lut_download(qm_engine *e, int32_t *out);

/* largest (clamped) delay in the resident table; callers must keep it <= lsmp */
int qm_engine_lut_max(qm_engine *e, int32_t *max_delay);

/* Fused detect step == migrate() + find_max_coa() of QuakeScan._compute
 * (quakemigrate/signal/scan.py:635-638) without the volume, float64 throughout (the screened
 * sweep is opt-in: qm_engine_config "screen"; with device buffers the call never waits on the
 * host).
 * n_nodes_total: node count of the FULL grid (normalisation, migratelib.c:108).
 * Outputs [n_samples]: max_coa f64, max_norm_coa f64, max_coa_idx i64. */
int qm_engine_detect(qm_engine *e, const double *log_onsets, int onsets_on_device,
                     int32_t t_samples, int32_t fsmp, int32_t lsmp,
                     int32_t available, int64_t n_nodes_total, double *max_coa,
                     double *max_norm_coa, int64_t *max_coa_idx,
                     int out_on_device);

/* n_steps consecutive timesteps of the detect sweep in ONE launch: log_onsets f64
 * [n_steps][n_rows][t_samples] (every step with the same table, pads and `available`), outputs
 * [n_steps][n_samples].  What QuakeScan._continuous_compute (quakemigrate/signal/scan.py:407-470)
 * does one timestep at a time; timesteps are independent given their onsets, so a launch can hold
 * several -- which is what fills the GPU on the grids the reference's examples use (1e4-3e5 nodes:
 * one timestep is a fraction of a millisecond of work and a few workgroup rounds).  Every step's
 * result is the bits qm_engine_detect gives for it.  Launches that cannot carry a step axis (the
 * screened detect, tables of more than 64 rows on row blocks) run step by step inside the call;
 * qm_engine_get "steps_per_launch" reports what the last call did. */
int qm_engine_detect_batch(qm_engine *e, const double *log_onsets, int onsets_on_device,
                           int32_t n_steps, int32_t t_samples, int32_t fsmp, int32_t lsmp,
                           int32_t available, int64_t n_nodes_total, double *max_coa,
                           double *max_norm_coa, int64_t *max_coa_idx, int out_on_device);

/* Same step, but stop before the final normalisation: this engine's partial
 * (log2-domain maximum, global node index, sum of coalescence) per sample, on
 * the device, ready for the cross-GPU exchange.  [n_samples] each. */
int qm_engine_detect_partial(qm_engine *e, const double *log_onsets,
                             int onsets_on_device, int32_t t_samples,
                             int32_t fsmp, int32_t lsmp, int32_t available,
                             double *d_part_max, int64_t *d_part_idx,
                             double *d_part_sum);

/* Combine n_sets partials (device, [n_sets][n_samples]) -- e.g. the all-gathered
 * per-GPU partials, in rank order -- into the final series.  Ties go to the
 * lowest node index, as in migratelib.c:102. */
int qm_engine_finalize(qm_engine *e, const double *d_part_max,
                       const int64_t *d_part_idx, const double *d_part_sum,
                       int32_t n_sets, int32_t n_samples, int64_t n_nodes_total,
                       double *max_coa, double *max_norm_coa,
                       int64_t *max_coa_idx, int out_on_device);

/* The same for the packed layout of the cross-GPU exchange: d_packed is f64 [n_sets][3][n_samples]
 * on the device, per set the rows (maxima, indices as int64 bit patterns, sums) -- what one
 * all-gather of every rank's [3][n_samples] partial produces ("one
 * ncclAllGather of a packed [3][ns] buffer + local combine"). */
int qm_engine_finalize_packed(qm_engine *e, const double *d_packed, int32_t n_sets,
                              int32_t n_samples, int64_t n_nodes_total, double *max_coa,
                              double *max_norm_coa, int64_t *max_coa_idx, int out_on_device);

/* "tie_rule" = 1 on a sharded detect -- the reference's arg-max rule on near-ties (migratelib.c:98-105: strict
 * '>' over EXPONENTIATED stacks) across ranks.  After the exchange of the partials:
 * qm_engine_tie_partial examines the partial sets this engine's last qm_engine_detect_partial of the same step
 * left behind against the GRID's largest z per sample (taken from d_packed = the gathered partials, f64
 * [n_sets][3][n_samples] as for qm_engine_finalize_packed) and writes d_tie_packed f64 [2][n_samples] (bit
 * patterns): row 0 the largest correctly rounded exp among its nodes within the slack (0: none), row 1 the
 * lowest GLOBAL node index reaching it -- ready for one more all-gather;
 * qm_engine_tie_fold takes the gathered [n_sets][2][n_samples] and overwrites, in the device series
 * d_max_coa_idx [n_samples], every sample some rank refined: largest exp, lowest index among the ranks reaching
 * it (what the reference's loop over ascending flat indices returns).  Samples nobody refined keep the default
 * rule's index.  Both calls only enqueue. */
int qm_engine_tie_partial(qm_engine *e, const double *log_onsets, int onsets_on_device,
                          int32_t t_samples, int32_t fsmp, int32_t lsmp, int32_t available,
                          const double *d_packed, int32_t n_sets, double *d_tie_packed);
int qm_engine_tie_fold(qm_engine *e, const double *d_tie_gathered, int32_t n_sets, int32_t n_samples,
                       int64_t *d_max_coa_idx);

/* Materialising step (locate): volume f64 [n_nodes_local][n_samples] is written
 * (accumulate != 0: added on top of its current content first, the reference's
 * `+=`), and, if max_coa != NULL, the scan outputs too (same meaning as
 * qm_engine_detect).  map4d may be host or device memory. */
int qm_engine_migrate(qm_engine *e, const double *log_onsets, int onsets_on_device,
                      int32_t t_samples, int32_t fsmp, int32_t lsmp,
                      int32_t available, int64_t n_nodes_total, double *map4d,
                      int map_on_device, int accumulate, double *max_coa,
                      double *max_norm_coa, int64_t *max_coa_idx,
                      int out_on_device);

/* Locate without the volume: the coalescence of every node summed over the scanned samples
 * [first_sample, end_sample) -- what `event.trim2window()` followed by
 * `np.sum(event.map4d, axis=-1)` computes from the 4-D map (quakemigrate/io/event.py:421-439,
 * signal/scan.py:720) -- written as f64 [n_nodes_local]; the scan outputs as in
 * qm_engine_migrate if max_coa != NULL.  Nothing of size n_nodes x n_samples is stored. */
int qm_engine_marginal(qm_engine *e, const double *log_onsets, int onsets_on_device,
                       int32_t t_samples, int32_t fsmp, int32_t lsmp, int32_t available,
                       int64_t n_nodes_total, int32_t first_sample, int32_t end_sample,
                       double *coa_map, int map_on_device, double *max_coa,
                       double *max_norm_coa, int64_t *max_coa_idx, int out_on_device);

/* Locate post-reductions of a marginalised 3-D coalescence map f64 [nx][ny][nz] (the output of
 * qm_engine_marginal), on the device -- QuakeScan._calculate_location's array work
 * (quakemigrate/signal/scan.py:696-733):
 *   - coa_map / nanmax(coa_map)                                       (scan.py:721)
 *   - _gaufilt3d: fftconvolve with util.gaussian_3d(nx, ny, nz, sgm), mode "same", twice, the
 *     second time mirrored, each normalised by its maximum           (scan.py:1008-1043,
 *     util.py:76-116), evaluated as a separable direct convolution
 *   - _covfit3d: weights = map where map > cov_thresh; expectation and 3x3 covariance of the
 *     node positions ix*node_spacing[0], ...                          (scan.py:939-1005)
 *   - the 7x7x7 window of the smoothed map around its maximum that _gaufit3d fits and the
 *     5x5x5 window of the normalised map that _splineloc interpolates (scan.py:736-936);
 *     nodes outside the grid are NaN.  The 10-parameter least squares / RBF solves on those
 *     few hundred values stay with the caller.
 * norm_map / smoothed_map: optional f64 [nx*ny*nz] outputs (host or device per out_on_device).
 * summary (host, 16 doubles): 0 nanmax of the input; 1 flat index of the first maximum of the
 * normalised map; 2 mean of the smoothed map; 3 flat index of the first maximum of the
 * smoothed map; 4 total weight; 5-7 expectation (xe, ye, ze); 8-13 covariance xx, yy, zz, xy,
 * xz, yz; 14, 15 the maxima the two smoothing passes were normalised by.
 * gau_window: host f64 [343]; spline_window: host f64 [125]. */
int qm_engine_locate_fits(qm_engine *e, const double *coa_map, int map_on_device, int32_t nx,
                          int32_t ny, int32_t nz, double sgm, double cov_thresh,
                          const double *node_spacing, double *norm_map, double *smoothed_map,
                          int out_on_device, double *summary, double *gau_window,
                          double *spline_window);

/* _splineloc's refinement (quakemigrate/signal/scan.py:777-812): the cubic radial-basis
 * interpolant through an n x n x n window -- scipy.interpolate.Rbf(x, y, z, values,
 * function="cubic") with the reference's "xy" meshgrid pairing -- evaluated on the device on the
 * `upscale`-times finer grid of ((n-1)*upscale + 1)^3 points, and its first maximum.
 * weights: host f64 [n][n][n], the solution of  |c_i - c_j|^3 w = values  (a 125 x 125 solve for
 * the reference's 5^3 window: left to the caller's LAPACK).  peak_index: flat C-order index
 * (i*m + j)*m + k into the fine grid, m = (n-1)*upscale + 1 -- what
 * np.unravel_index(np.nanargmax(dense), dense.shape) receives in scan.py:806-808. */
int qm_engine_rbf_peak(qm_engine *e, const double *weights, int32_t n, int32_t upscale,
                       double *peak_value, int64_t *peak_index);

/* Onset stage on the device -- the step immediately upstream of the path:
 * STALTAOnset._onset (quakemigrate/signal/onsets/stalta.py:491-548: signal transform, STA/LTA
 * per component trace with the arithmetic of core/src/onsetlib.c, taper windows :550-583,
 * root-mean-square over the components, clip at min_onset_value) and then lib.migrate's
 * log(clip(., 0.01)) (core/lib.py:93-94).
 *   signals    f64 [n_traces][t_samples], pre-processed (filtered / resampled) waveforms
 *   trace_row  [n_traces] onset row each trace feeds (the components of one station/phase)
 *   nsta/nlta  [n_rows] window lengths in samples;  transform 0 = energy (x*x), 1 = abs.
 *              The reference's "env" / "env_squared" (stalta.py:518-521) are abs / energy of the
 *              envelope |hilbert(x)|: the Hilbert transform is an upstream signal transform like
 *              the band-pass filters and stays with the caller (the Python binding applies
 *              scipy.signal.hilbert, the reference's own call, to host signals).
 *   position   0 = classic / overlapping (onsetlib.c:35-59), 1 = centred (:79-108),
 *              2 = recursive (:126-148; exported by the reference's lib, not used by
 *              STALTAOnset);  taper_pad < 0 = no taper windows
 *   raw_onsets (optional) and log_onsets: f64 [n_rows][t_samples]; log_onsets is what
 *   qm_engine_detect takes (pass it with onsets_on_device = 1 to keep everything on the GPU). */
int qm_engine_onsets(qm_engine *e, const double *signals, int signals_on_device,
                     int32_t n_traces, int32_t t_samples, const int32_t *trace_row,
                     int32_t n_rows, const int32_t *nsta, const int32_t *nlta, int transform,
                     int position, int32_t taper_pad, double min_onset_value,
                     double *raw_onsets, double *log_onsets, int out_on_device);

/* exp(x) rounded to nearest from a double-double evaluation (csrc/qm_ties.hpp): the function the
 * opt-in arg-max rule "tie_rule" = 1 compares near-tied nodes on (the reference exponentiates, then
 * compares: migratelib.c:60-62, :98-105).  Host code, exported for the tests that pin it. */
double qm_exp_correctly_rounded(double x);
/* ... and the GPU's evaluation of the same source, host arrays in and out (the tests pin the two together) */
int qm_engine_exp_correctly_rounded(qm_engine *e, const double *x, int64_t n, double *out);

/* Self-check behind the screened detect's error bound (qm_screen.hpp): the largest relative
 * deviation of the device's v_exp_f32 from the float64 exp2 over EVERY float32 in [lo, hi]
 * (lo <= hi, same sign).  The bound quotes <= 2^-23 for it. */
int qm_exp2f_max_error(qm_engine *e, float lo, float hi, double *max_rel_error);

/* Scan of an existing volume (find_max_coa semantics, no table needed). */
int qm_engine_find_max_coa(qm_engine *e, const double *map4d, int map_on_device,
                           int32_t n_samples, int64_t n_nodes, double *max_coa,
                           double *max_norm_coa, int64_t *max_coa_idx,
                           int out_on_device);

/* ------------------------------------------------------------------ part 3 */
/* The continuous detect sweep as a pipeline -- the hot-path part of the loop of
 * QuakeScan._continuous_compute (quakemigrate/signal/scan.py:434-448: one timestep after the other:
 * onsets -> migrate -> find_max_coa -> append), with host-resident onsets going in and the three
 * series coming out per timestep.  A ring of `depth` slots of `steps_per_launch` timesteps each;
 * qm_stream_push copies one timestep's log-onsets (host f64 [n_rows][t_samples], the resident table's
 * row count) into pinned memory and, when a slot is full, enqueues its H2D copy (own stream) and ONE
 * fused-detect launch for its timesteps on the engine's stream (qm_engine_detect_batch: every
 * timestep's bits are the single-step call's) whose last kernel writes the results straight into the
 * slot's pinned host buffer; nothing waits.  qm_stream_pop hands out the results of the oldest timesteps in push order and is the
 * only call that blocks (for the launch they belong to).
 * qm_stream_push returns 0, or 2 when every slot holds results that have not been popped (pop, then
 * push again), or 1 on error.  qm_stream_flush launches a partly filled slot (end of the data).
 * The engine's table, stream and tunables must not change while a stream exists on it. */
typedef struct qm_stream qm_stream;
int qm_stream_create(qm_engine *e, int32_t t_samples, int32_t fsmp, int32_t lsmp, int32_t available,
                     int64_t n_nodes_total, int32_t steps_per_launch, int32_t depth, qm_stream **out);
void qm_stream_destroy(qm_stream *s);
int qm_stream_push(qm_stream *s, const double *log_onsets);
int qm_stream_flush(qm_stream *s);
/* the next n_steps timesteps (<= launched and not yet popped): host f64 / f64 / i64 [n_steps][n_samples] */
int qm_stream_pop(qm_stream *s, int32_t n_steps, double *max_coa, double *max_norm_coa,
                  int64_t *max_coa_idx);
int qm_stream_pending(qm_stream *s, int32_t *launched_not_popped, int32_t *pushed_not_launched);

/* Duration (ms, HIP events on the engine stream) of the stacking kernel(s) of
 * the most recent detect / migrate call; negative if none.  Synchronises. */
int qm_engine_last_kernel_ms(qm_engine *e, double *ms);

/* With config "log_timing" = 1 every stacking launch is bracketed by its own
 * pair of HIP events (on the engine stream); this returns the summed duration
 * and the number of launches since the last call, and resets the log. */
int qm_engine_kernel_log(qm_engine *e, double *total_ms, int32_t *n_calls);

#ifdef __cplusplus
}
#endif
#endif /* QMHIP_H */
