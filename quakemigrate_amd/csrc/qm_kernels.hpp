// qm_kernels.hpp -- gfx950 (CDNA4) device code of the coalescence-migration engine.
//
// What is computed (reference: quakemigrate/core/src/migratelib.c:40-65 and :85-111):
//   stack[n][t] = sum_{r=0..S-1} L[r][max(0,tt[n][r]) + fsmp + t]    (float64, ascending r)
//   coa[n][t]   = exp(stack[n][t] * (1/available))
//   per t: max_n coa, first n reaching it, sum_n coa.
//
// Design: a gather + reduce, so no MFMA.  The grid is cut into small
// 3-D bricks of nodes whose delays to any one station differ little; for one brick and one
// tile of 64*J samples the window of every log-onset row that the brick can touch is staged
// in LDS once, then every node of the brick streams its S operands per sample out of LDS
// (lanes = consecutive samples -> conflict-free ds_read_b64).  The per-node window offsets
// come from a brick-relative 16-bit table built once per travel-time table.  exp, the running
// max / argmax / sum are fused, so the detect path never writes the 4-D volume.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// The stacking kernels are templates, instantiated family by family in the launch-table units
// (qm_launch_*.hip); each plain kernel of this header is compiled by ONE of the engine's translation
// units, the one that launches it: QM_TU_TABLES (qm_tables.hip), QM_TU_STEPS (qm_engine.hip),
// QM_TU_WIDEN (qm_widen.hip), QM_TU_SCREEN (qm_screen.hip) -- each defines its macro before including it.
namespace qm {

constexpr int kWave = 64;
constexpr int64_t kNoIndex = INT64_MAX;
constexpr int kMaxSpanBytes = 65535;   // window offsets are stored as uint16 BYTE offsets

struct GridDesc {
    int nx, ny, nz;        // nodes of the resident (possibly sharded) table
    int bx, by, bz;        // brick shape
    int nbx, nby, nbz;     // bricks per axis
    int nbricks;
    int brick_nodes;       // bx*by*bz
    int n_rows;            // S
    int row_pad;           // S rounded up to a multiple of 8 (uint16 entries per node row)
};

struct StackArgs {
    GridDesc g;
    const double *onsets;          // [S][T] log-onsets
    const int32_t *lut;            // [N][S] original table
    const uint16_t *rel;           // [nbricks][brick_nodes][row_pad] window byte offsets
    const int32_t *brick_meta;     // [nbricks][S][4] = (min delay, span, exclusive prefix of
                                   //                    span, 0) per table row
    const int32_t *brick_total;    // [nbricks] sum of span
    const int32_t *brick_list;     // direct kernel: bricks to process (or nullptr = all)
    int n_list;
    int T, fsmp, n_samples;
    int sample0;                   // first scanned sample handled by this launch (chunking)
    int n_chunk;                   // samples handled by this launch
    int ntiles, ngroups;
    int tail_spl;                  // shift-reuse kernels: samples per lane of the scan's tail tile (1..3;
                                   // 0: whole 256-sample tiles only, the last one pulled back)
    int wide_tiles;                // shift-reuse kernels, round 6: the launch's first tiles are this many WIDE
                                   // tiles of 384 samples (six per lane, qm_shift.hpp); 256-sample tiles follow
    // K timesteps in one launch (detect-type launches): step k scans onsets + k * step_stride with
    // the same table and geometry and publishes its partial sets at column k * n_chunk of rows that
    // are part_stride long.  The time-tile axis of the grid is n_steps * ntiles long.
    int n_steps;                   // 0 or 1: a single step
    int64_t step_stride;           // doubles between the onset arrays of consecutive steps
    int64_t part_stride;           // elements per partial set (0: n_chunk)
    int cap_doubles;               // LDS window capacity in doubles
    double z_scale;                // log2(e) / available: z = stack * z_scale, coa = 2^z
    double *volume;                // [N][vol_stride] or nullptr
    int64_t vol_stride;            // samples per node row in `volume`
    int accumulate;                // start from the volume's content (reference '+=')
    double *part_max;              // [sets][n_chunk]  log2-domain maxima (z)
    int64_t *part_idx;             // [sets][n_chunk]  local flat node index
    double *part_sum;              // [sets][n_chunk]
    int set0;                      // first partial set written by this launch
    double *brick_max;             // [nbricks][part_stride] shift-reuse fused detect, tie_rule = 1: the largest z per
                                   // brick and sample besides the workgroups' partial sets (nullptr: not wanted)
    int want_scan;                 // write partials at all
    double *marginal;              // [ntiles][n_nodes] per-tile sums over samples [m0, m1) of the
    int m0, m1;                    //   coalescence (VOLUME kernels; replaces the volume store)
    int64_t n_nodes;
    const int32_t *run_if;         // if set: do nothing unless *run_if != 0 (device-side fallback
                                   // of a screened step, qm_screen.hpp)
};

// max of two non-NaN-producing operands without the canonicalising copy clang adds to fmax()
// (NaN in `x` is ignored exactly like the reference's `current > max` test ignores it)
__device__ __forceinline__ double max_keep(double best, double x) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(best), "v"(x));
    return r;
}

template <typename I>
__device__ __forceinline__ bool better(double v, I i, double best, I bi) {
    return (v > best) || (v == best && i < bi);
}

// XCD-aware workgroup -> (time tile, brick group) map of the stacking kernels: workgroup b is observed
// to run on XCD b % 8, each XCD has its own L2; all time tiles of one brick group read the same slice
// of the offset table / record stream, so whole groups are dealt to XCDs (group = xcd + 8 * k) and a
// slice is fetched into one L2 instead of eight.  The grid is padded to a multiple of 8 groups.
// With several timesteps per launch the tile axis runs over (step, tile); `step_view` is the
// launch's arguments as that step sees them (its onsets, its columns of the partial sets).
__device__ __forceinline__ void stack_tile_group(const StackArgs &a, int &tile, int &group) {
    const int slot = blockIdx.x >> 3;
    const int all = a.ntiles * (a.n_steps > 1 ? a.n_steps : 1);
    tile = (slot % all) % a.ntiles;
    group = (int)(blockIdx.x & 7) + 8 * (slot / all);
}
__device__ __forceinline__ StackArgs step_view(StackArgs a) {
    if (a.part_stride == 0) a.part_stride = a.n_chunk;
    if (a.n_steps > 1) {
        const int step = (int)((blockIdx.x >> 3) % (unsigned)(a.ntiles * a.n_steps)) / a.ntiles;
        a.onsets += (int64_t)step * a.step_stride;
        a.part_max += (int64_t)step * a.n_chunk;
        a.part_idx += (int64_t)step * a.n_chunk;
        a.part_sum += (int64_t)step * a.n_chunk;
        if (a.brick_max) a.brick_max += (int64_t)step * a.n_chunk;
    }
    return a;
}

// a brick fits the LDS-tiled kernel iff its byte offsets fit uint16 and its windows fit LDS
__host__ __device__ __forceinline__ bool brick_fits(int64_t total_span, int n_rows, int kt,
                                                    int cap_doubles) {
    return total_span * 8 <= kMaxSpanBytes && total_span + (int64_t)n_rows * kt <= cap_doubles;
}

// ---------------------------------------------------------------------------------------
// Table preparation (once per qm_engine_load_lut)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void brick_extents(const GridDesc &g, int b, int &x0, int &y0, int &z0,
                                              int &vx, int &vy, int &vz) {
    const int bzi = b % g.nbz, byi = (b / g.nbz) % g.nby, bxi = b / (g.nbz * g.nby);
    x0 = bxi * g.bx; y0 = byi * g.by; z0 = bzi * g.bz;
    vx = min(g.bx, g.nx - x0); vy = min(g.by, g.ny - y0); vz = min(g.bz, g.nz - z0);
}

// m-th node of brick b in walk order: only nodes inside the grid, lexicographic in (x, y, z)
// over the brick's valid extents (ascending flat index, no gaps)
__device__ __forceinline__ int brick_walk_node(const GridDesc &g, int x0, int y0, int z0, int vy,
                                               int vz, int m) {
    const int lz = m % vz, ly = (m / vz) % vy, lx = m / (vz * vy);
    return ((x0 + lx) * g.ny + (y0 + ly)) * g.nz + (z0 + lz);
}

// one workgroup per brick, thread r <-> table row r: min / span of the clamped delays.
#ifdef QM_TU_TABLES
__global__ void brick_minmax_kernel(GridDesc g, const int32_t *__restrict__ lut,
                                    int4 *__restrict__ meta, int32_t *__restrict__ global_max) {
    const int b = blockIdx.x;
    int x0, y0, z0, vx, vy, vz;
    brick_extents(g, b, x0, y0, z0, vx, vy, vz);
    const int nvalid = vx * vy * vz;
    for (int r = threadIdx.x; r < g.n_rows; r += blockDim.x) {
        int lo = INT32_MAX, hi = 0;
        for (int m = 0; m < nvalid; ++m) {
            const int node = brick_walk_node(g, x0, y0, z0, vy, vz, m);
            int d = lut[(int64_t)node * g.n_rows + r];
            d = d < 0 ? 0 : d;                           // migratelib.c:55
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
        meta[(int64_t)b * g.n_rows + r] = make_int4(lo, hi - lo, 0, 0);
        atomicMax(global_max, hi);
    }
}
#endif  // QM_TU_TABLES

#ifdef QM_TU_TABLES
__global__ void brick_prefix_kernel(GridDesc g, int4 *__restrict__ meta,
                                    int32_t *__restrict__ btotal) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.nbricks) return;
    int64_t run = 0;
    for (int r = 0; r < g.n_rows; ++r) {
        meta[(int64_t)b * g.n_rows + r].z = (int32_t)(run > INT32_MAX ? INT32_MAX : run);
        run += meta[(int64_t)b * g.n_rows + r].y;
    }
    btotal[b] = (int32_t)(run > INT32_MAX ? INT32_MAX : run);
}
#endif  // QM_TU_TABLES

// Brick-relative window offsets: rel[b][m][r] = 8 * (off_r + clamp(tt) - min_r) BYTES, uint16,
// one row of row_pad entries per node, nodes in walk order (see brick_walk_node).
#ifdef QM_TU_TABLES
__global__ void brick_rel_kernel(GridDesc g, const int32_t *__restrict__ lut,
                                 const int4 *__restrict__ meta,
                                 const int32_t *__restrict__ btotal,
                                 uint16_t *__restrict__ rel) {
    const int b = blockIdx.x;
    const bool narrow = (int64_t)btotal[b] * 8 <= kMaxSpanBytes;
    const int per = g.brick_nodes * g.row_pad;
    int x0, y0, z0, vx, vy, vz;
    brick_extents(g, b, x0, y0, z0, vx, vy, vz);
    const int nvalid = vx * vy * vz;
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const int m = i / g.row_pad, r = i % g.row_pad;
        uint16_t v = 0;
        if (narrow && r < g.n_rows && m < nvalid) {
            const int node = brick_walk_node(g, x0, y0, z0, vy, vz, m);
            int d = lut[(int64_t)node * g.n_rows + r];
            d = d < 0 ? 0 : d;
            const int4 rec = meta[(int64_t)b * g.n_rows + r];
            v = (uint16_t)(8 * (rec.z + d - rec.x));
        }
        rel[(int64_t)b * per + i] = v;
    }
}
#endif  // QM_TU_TABLES

// ---------------------------------------------------------------------------------------
// Onset stage on the device: STALTAOnset._onset (quakemigrate/signal/onsets/stalta.py:491-548,
// trim/taper :550-583) followed by lib.migrate's clip + log (core/lib.py:93-94).
//   pass 1  one thread per trace, sequential: the running short / long sums of the transformed
//           signal in exactly the operation order of onsetlib.c:35-59 / :79-108 (the order fixes
//           their rounding), stored per sample; position 2: the exponentially weighted averages
//           of recursive_sta_lta (onsetlib.c:126-148; not used by STALTAOnset itself, offered
//           because the reference exports it), multiplications and additions kept unfused;
//   pass 2  parallel: ratio * nlta/nsta, the 1.0 fill outside the valid range, the taper
//           windows, then per onset row the root-mean-square over its component traces,
//           clip at min_onset_value (raw onset) and log(clip(., 0.01)) (what the stack reads).
// ---------------------------------------------------------------------------------------
struct OnsetArgs {
    const double *signals;     // [n_traces][T] pre-processed waveforms
    const int32_t *trace_row;  // [n_traces] onset row of each trace (ascending)
    const int32_t *nsta;       // [n_rows]
    const int32_t *nlta;       // [n_rows]
    double *sta, *lta;         // [n_traces][T] scratch
    double *raw;               // [n_rows][T] or nullptr
    double *logged;            // [n_rows][T]
    int n_traces, n_rows, T;
    int transform;             // 0: energy x*x, 1: abs
    int position;              // 0: classic (overlapping windows), 1: centred, 2: recursive
    int taper_pad;             // samples; < 0: no taper windows
    double min_onset_value;
};

__device__ __forceinline__ double onset_transform(double x, int transform) {
    return transform == 0 ? x * x : __builtin_fabs(x);
}

// One workgroup per trace.  The transformed signal is staged in LDS by all threads; thread 0 then
// runs the recurrences from LDS (the sums are a serial chain -- keeping the reference's operation
// order is what makes degenerate stretches, e.g. a dead trace whose window sums are rounding
// residue, come out as in the reference), unrolled so that the LDS reads of the next steps are in
// flight while the current additions retire.  `f` = LDS copy, or the global signal if the trace
// does not fit (then the transform is applied on the fly).
template <bool IN_LDS>
__device__ __forceinline__ double onset_sample(const double *f, const double *x, int i, int tf) {
    return IN_LDS ? f[i] : onset_transform(x[i], tf);
}

template <bool IN_LDS>
__device__ __forceinline__ void stalta_recurrence(const OnsetArgs &a, const double *f,
                                                  const double *x, double *S, double *L, int ns,
                                                  int nl, int n) {
    const int tf = a.transform;
    double s_short = 0.0, s_long = 0.0;
    if (a.position == 0) {                              // onsetlib.c:35-59
        for (int i = 0; i < ns; ++i) s_short += onset_sample<IN_LDS>(f, x, i, tf);
        s_long = s_short;
        for (int i = ns; i < nl; ++i) {
            const double in = onset_sample<IN_LDS>(f, x, i, tf);
            s_long += in;
            s_short += in - onset_sample<IN_LDS>(f, x, i - ns, tf);
        }
        S[nl - 1] = s_short;
        L[nl - 1] = s_long;
        int i = nl;
        for (; i + 4 <= n; i += 4) {
            double in[4], os[4], ol[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                in[k] = onset_sample<IN_LDS>(f, x, i + k, tf);
                os[k] = onset_sample<IN_LDS>(f, x, i + k - ns, tf);
                ol[k] = onset_sample<IN_LDS>(f, x, i + k - nl, tf);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_short += in[k] - os[k];
                s_long += in[k] - ol[k];
                S[i + k] = s_short;
                L[i + k] = s_long;
            }
        }
        for (; i < n; ++i) {
            const double in = onset_sample<IN_LDS>(f, x, i, tf);
            s_short += in - onset_sample<IN_LDS>(f, x, i - ns, tf);
            s_long += in - onset_sample<IN_LDS>(f, x, i - nl, tf);
            S[i] = s_short;
            L[i] = s_long;
        }
    } else if (a.position == 2) {                       // onsetlib.c:126-148
#pragma clang fp contract(off)
        const double cs = 1.0 / (double)ns, cl = 1.0 / (double)nl;
        S[0] = 0.0;
        L[0] = 0.0;
        for (int i = 1; i < n; ++i) {
            const double in = onset_sample<IN_LDS>(f, x, i, tf);
            s_short = cs * in + (1 - cs) * s_short;
            s_long = cl * in + (1 - cl) * s_long;
            S[i] = s_short;
            L[i] = s_long;
        }
    } else {                                            // onsetlib.c:79-108
        if (nl + ns > n) return;
        for (int i = 0; i < nl; ++i) s_long += onset_sample<IN_LDS>(f, x, i, tf);
        for (int i = nl; i < nl + ns; ++i) s_short += onset_sample<IN_LDS>(f, x, i, tf);
        S[nl - 1] = s_short;
        L[nl - 1] = s_long;
        int i = nl;
        for (; i + 4 <= n - ns; i += 4) {
            double ahead[4], here[4], old[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ahead[k] = onset_sample<IN_LDS>(f, x, i + k + ns, tf);
                here[k] = onset_sample<IN_LDS>(f, x, i + k, tf);
                old[k] = onset_sample<IN_LDS>(f, x, i + k - nl, tf);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_short += ahead[k] - here[k];
                s_long += here[k] - old[k];
                S[i + k] = s_short;
                L[i + k] = s_long;
            }
        }
        for (; i < n - ns; ++i) {
            s_short += onset_sample<IN_LDS>(f, x, i + ns, tf) - onset_sample<IN_LDS>(f, x, i, tf);
            s_long += onset_sample<IN_LDS>(f, x, i, tf) - onset_sample<IN_LDS>(f, x, i - nl, tf);
            S[i] = s_short;
            L[i] = s_long;
        }
    }
}

#ifdef QM_TU_WIDEN
__global__ __launch_bounds__(256) void stalta_sums_kernel(OnsetArgs a, int in_lds) {
    extern __shared__ double fx[];
    const int tr = blockIdx.x;
    const int row = a.trace_row[tr];
    const int ns = a.nsta[row], nl = a.nlta[row], n = a.T;
    const double *x = a.signals + (int64_t)tr * n;
    double *S = a.sta + (int64_t)tr * n, *L = a.lta + (int64_t)tr * n;
    if (a.position != 2 && (nl > n || ns > nl || ns < 1)) return;   // pass 2 leaves such a trace at 1.0
    if (a.position == 2 && (ns < 1 || nl < 1)) return;
    if (in_lds) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) fx[i] = onset_transform(x[i], a.transform);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    if (in_lds) stalta_recurrence<true>(a, fx, x, S, L, ns, nl, n);
    else stalta_recurrence<false>(a, fx, x, S, L, ns, nl, n);
}
#endif  // QM_TU_WIDEN

// thread <-> (row, sample); the components of a row are consecutive traces
#ifdef QM_TU_WIDEN
__global__ void onset_rows_kernel(OnsetArgs a) {
#pragma clang fp contract(off)                           // numpy squares, then adds (stalta.py:544)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.n_rows * a.T) return;
    const int row = (int)(i / a.T), t = (int)(i % a.T);
    const int ns = a.nsta[row], nl = a.nlta[row], n = a.T;
    const double frac = (double)nl / (double)ns;
    double sumsq = 0.0;
    int count = 0;
    for (int tr = 0; tr < a.n_traces; ++tr) {
        if (a.trace_row[tr] != row) continue;
        double v = 1.0;                                  // lib.py pre-fills with ones
        if (a.position == 2) {
            // recursive: the binding pre-fills with zeros (lib.py:279), sample 0 is never written,
            // the first nlta samples are nulled to 1 when the trace is longer than that
            v = 0.0;
            if (ns >= 1 && nl >= 1) {
                if (t >= 1) v = a.sta[(int64_t)tr * n + t] / a.lta[(int64_t)tr * n + t];
                if (nl < n && t < nl) v = 1.0;
            }
        } else {
            const bool sane = !(nl > n || ns > nl || ns < 1) && (a.position == 0 || nl + ns <= n);
            const int last = a.position == 0 ? n - 1 : n - ns - 1;
            if (sane && t >= nl - 1 && t <= last) {
                const double s = a.sta[(int64_t)tr * n + t], l = a.lta[(int64_t)tr * n + t];
                if (a.position == 0 || t == nl - 1 || l > 0.0) v = s / l * frac;
            }
        }
        if (a.taper_pad >= 0 && (t < a.taper_pad + nl - 1 || t >= n - (ns + a.taper_pad)))
            v = 1.0;                                     // stalta.py:579-581
        sumsq += v * v;
        ++count;
    }
    double onset = __builtin_sqrt(sumsq / (double)count);             // stalta.py:544
    onset = onset < a.min_onset_value ? a.min_onset_value : onset;     // :546
    if (a.raw) a.raw[i] = onset;
    a.logged[i] = log(onset < 0.01 ? 0.01 : onset);                    // lib.py:93-94
}
#endif  // QM_TU_WIDEN

// ---------------------------------------------------------------------------------------
// Table serving on the device (what LUT.serve_traveltimes does on the host every timestep,
// quakemigrate/lut/lut.py:502-538): from float64 travel-time grids in seconds, one [N_full] grid
// per station/phase, build the int32 [N][S] table of the selected rows as rint(tt * rate)
// (v_rndne_f64 = numpy's half-to-even rint), optionally decimated like Grid3D.decimate
// (lut.py:102-140: every df-th node starting at c1).  64 nodes x all rows per workgroup, staged
// through LDS so that both the reads (along nodes) and the writes (along rows) are coalesced.
// ---------------------------------------------------------------------------------------
struct ServeArgs {
    const double *grids;       // [rows_total][nx_full*ny_full*nz_full]
    const int32_t *rows;       // [S] selected grid index per table row
    int32_t *out;              // [N][S]
    int nxf, nyf, nzf;         // full grid
    int nx, ny, nz;            // served (decimated) grid
    int dfx, dfy, dfz, c1x, c1y, c1z;
    int S;
    double rate;
};

#ifdef QM_TU_TABLES
// NPB nodes x all rows per workgroup of 256 threads: thread t reads the grids of node t, t + 256, ...
// of the workgroup row by row (coalesced along the nodes, eight loads in flight), the rounded
// delays are transposed through LDS, and the workgroup's NPB x S block of the table -- contiguous
// in memory -- is written coalesced.  One pass: 8 bytes read and 4 written per table entry.
template <int NPB>
__global__ __launch_bounds__(256) void serve_table_kernel(ServeArgs a) {
    extern __shared__ int32_t tile[];                   // [NPB][pitch]
    const int64_t n_out = (int64_t)a.nx * a.ny * a.nz;
    const int64_t n_full = (int64_t)a.nxf * a.nyf * a.nzf;
    const int64_t n0 = (int64_t)blockIdx.x * NPB;
    const int S = a.S, pitch = (S + 1) | 1;            // (odd: the row-wise stores spread over the banks)
    for (int m = threadIdx.x; m < NPB; m += blockDim.x) {
        const int64_t n = n0 + m;
        if (n >= n_out) break;
        const int iz = (int)(n % a.nz), iy = (int)((n / a.nz) % a.ny), ix = (int)(n / ((int64_t)a.nz * a.ny));
        const int64_t src = ((int64_t)(a.c1x + ix * a.dfx) * a.nyf + (a.c1y + iy * a.dfy)) * a.nzf +
                            (a.c1z + iz * a.dfz);
        constexpr int U = 8;
        for (int s0 = 0; s0 < S; s0 += U) {
            double t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u < S ? s0 + u : S - 1;
                t[u] = __builtin_nontemporal_load(a.grids + (int64_t)a.rows[s] * n_full + src);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + u >= S) break;
                // np.rint(tt * sr).astype(np.int32), lut.py:538.  What the cast gives outside int32
                // is the host's: on x86-64 (cvttsd2si) NaN, the infinities and everything beyond
                // [-2^31, 2^31) become INT32_MIN -- a negative delay, which migrate clamps to 0
                // (migratelib.c:55).  The GPU's own conversion would saturate (NaN -> 0, +huge ->
                // INT32_MAX): reproduce the host (fixture serve_nonfinite.npz, from the reference's
                // own class).
                const double r = __builtin_rint(t[u] * a.rate);
                tile[m * pitch + s0 + u] = (r >= -2147483648.0 && r <= 2147483647.0) ? (int32_t)r : INT32_MIN;
            }
        }
    }
    __syncthreads();
    const int64_t count = (n_out - n0 < NPB ? n_out - n0 : NPB) * S;
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) {
        const int node = (int)(i / S), s = (int)(i % S);
        a.out[n0 * S + i] = tile[node * pitch + s];
    }
}
#endif  // QM_TU_TABLES

// ---------------------------------------------------------------------------------------
// exp in float64, in the base-2 domain, without the device library's special-case handling: the
// argument is a mean of log-onsets (|x| < ~50 for any finite input), far from overflow.
//   z = stack * (log2(e)/available) = k + f, |f| <= 1/2;   coa = 2^z = 2^k * 2^f
//   2^f = sum_i (f ln2)^i / i!  as a Horner polynomial, scaled with v_ldexp_f64.
// Per node-sample: a minimax polynomial (below) -- this feeds the volume and the sum behind
// max_norm_coa (contract: 1e-6).  The running maximum is tracked on z itself (monotone in the
// stack), and the final peak 2^z_best is evaluated once per sample with the degree-13 Taylor
// form (~1 ulp + the 1-ulp rounding of z).  The reference's own exp is glibc libmvec (<= 4 ulp,
// SURVEY 8c).
// ---------------------------------------------------------------------------------------
#define QM_LOG2E 1.4426950408889634074
// Polynomial of the per-node-sample 2^f, |f| <= 1/2.  Coefficients: Remez minimax in relative
// error (tools/exp2_minimax.py, 60-digit arithmetic, then rounded to float64; the bound is that
// of the ROUNDED polynomial evaluated exactly; the D Horner roundings add <= D * 1.2e-16):
//   degree  6: 1.86e-09     degree  7: 4.03e-11     degree  8: 7.75e-13
//   degree  9: 1.36e-14     degree 10: 2.8e-16
// Detect (the value only feeds the sum behind max_norm_coa; every term is positive, so the sum
// inherits at most the terms' relative error): QM_EXP2_DEGREE_SUM, default 8 -- the accuracy of
// the degree-10 Taylor form used before (2.2e-13 .. 7.8e-13), two multiply-adds cheaper.  Stored
// in the volume: QM_EXP2_DEGREE_VOLUME, default 10 (the accuracy of a degree-12 Taylor form).
#ifndef QM_EXP2_DEGREE_SUM
#define QM_EXP2_DEGREE_SUM 8
#endif
#ifndef QM_EXP2_DEGREE_VOLUME
#define QM_EXP2_DEGREE_VOLUME 10
#endif
template <bool VOLUME> struct Exp2Degree {
    static constexpr int value = VOLUME ? QM_EXP2_DEGREE_VOLUME : QM_EXP2_DEGREE_SUM;
};

// coefficient of f^i in the degree-D polynomial
template <int D>
__device__ __forceinline__ constexpr double exp2_coeff(int i) {
    static_assert((D >= 6 && D <= 10) || D == 13, "no coefficient set for this degree");
    if constexpr (D == 6) {
        constexpr double c[7] = {1.0000000005541665, 0.6931472057372681, 0.2402264689063409,
                                 0.055503287769647254, 0.009618488957115071, 0.001339993121934089,
                                 0.00015345812002903349};
        return c[i];
    } else if constexpr (D == 7) {
        constexpr double c[8] = {0.999999999961682, 0.6931471807284456, 0.24022651198157205,
                                 0.05550410353453057, 0.009618027253714257, 0.0013333922559115655,
                                 0.00015469291129672317, 1.5201922643927036e-05};
        return c[i];
    } else if constexpr (D == 8) {
        constexpr double c[9] = {0.9999999999997623, 0.6931471805465141, 0.24022650698880368,
                                 0.05550410939341707, 0.00961812854286291, 0.0013333452062251631,
                                 0.00015403851748367633, 1.5309737421285583e-05,
                                 1.3175858190329987e-06};
        return c[i];
    } else if constexpr (D == 9) {
        constexpr double c[10] = {1.0000000000000127, 0.693147180559871, 0.24022650695649653,
                                  0.05550410866868561, 0.009618129192067245, 0.0013333557617604443,
                                  0.0001540343494807179, 1.5252984838653427e-05,
                                  1.3259405609345135e-06, 1.0150336705309649e-07};
        return c[i];
    } else if constexpr (D == 10) {
        constexpr double c[11] = {1.0, 0.6931471805599497, 0.24022650695908768,
                                  0.05550410866445883, 0.009618129108034596, 0.0013333558228561797,
                                  0.0001540352996112841, 1.5252658116392011e-05,
                                  1.3215662835262992e-06, 1.020853793302931e-07,
                                  7.0372789704963916e-09};
        return c[i];
    } else {                                            // 13: Taylor, ln2^i / i! (the final peak)
        constexpr double c[14] = {1.0,
                                  0.6931471805599453,    0.24022650695910072,   0.05550410866482158,
                                  0.009618129107628477,  0.0013333558146428443, 0.0001540353039338161,
                                  1.5252733804059841e-05, 1.321548679014431e-06, 1.01780860092397e-07,
                                  7.054911620801123e-09, 4.4455382718708116e-10, 2.5678435993488206e-11,
                                  1.3691488853904128e-12};
        return c[i];
    }
}

// Horner step I of a degree-D polynomial: I = 0 starts with the two highest coefficients
template <int D, int I>
__device__ __forceinline__ double exp2_horner(double p, double f) {
    return __builtin_fma(I == 0 ? exp2_coeff<D>(D) : p, f, exp2_coeff<D>(D - 1 - I));
}

template <int D, int I = 0>
__device__ __forceinline__ double exp2_poly(double p, double f) {
    if constexpr (I < D) return exp2_poly<D, I + 1>(exp2_horner<D, I>(p, f), f);
    else return p;
}

// 2^z for the per-node path
template <int D>
__device__ __forceinline__ double qm_exp2(double z) {
    const double kf = __builtin_rint(z);
    const double f = z - kf;
    return __builtin_amdgcn_ldexp(exp2_poly<D>(0.0, f), (int)kf);
}

// 2^z for the final peak (once per sample)
__device__ __forceinline__ double qm_exp2_peak(double z) {
    const double kf = __builtin_rint(z);
    const double f = z - kf;
    return __builtin_amdgcn_ldexp(exp2_poly<13>(0.0, f), (int)kf);
}

// ---------------------------------------------------------------------------------------
// Per-wavefront reduction state: J samples per lane (t = t_first + lane + 64*j).
// `b*` follows the nodes of the current brick, which a wave visits in ascending flat index,
// so a strict '>' keeps the first maximum (migratelib.c:102); bricks are not visited in
// ascending order, so at the end of a brick the pair is merged into `v*` with the explicit
// lowest-index tie-break.
// ---------------------------------------------------------------------------------------
template <int J>
struct Running {
    double vmax[J], vsum[J], bmax[J];
    int vidx[J], bidx[J];           // local flat node index (< 2^31), INT32_MAX = none
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            vmax[j] = -__builtin_inf();
            vsum[j] = 0.0;
            vidx[j] = INT32_MAX;
        }
        reset_brick();
    }
    __device__ __forceinline__ void reset_brick() {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            bmax[j] = -__builtin_inf();
            bidx[j] = INT32_MAX;
        }
    }
    __device__ __forceinline__ void merge_brick() {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (better(bmax[j], bidx[j], vmax[j], vidx[j])) {
                vmax[j] = bmax[j];
                vidx[j] = bidx[j];
            }
        }
        reset_brick();
    }
};

template <int J, bool VOLUME>
__device__ __forceinline__ void finish_node(const StackArgs &a, Running<J> &run,
                                            const double (&acc)[J], int node, int t_first,
                                            int lane) {
    double e[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {                       // branch-free: the J chains interleave
        double x;                                       // z: log2 of the coalescence
        {   // a rounded product, as in the pipelined epilogues (there z crosses a loop back edge):
            // fused into f = fma(acc, scale, -k) the stored value would differ in its last bits
            // from what the other stacking kernels write for the same node-sample
#pragma clang fp contract(off)
            x = acc[j] * a.z_scale;
        }
        e[j] = qm_exp2<Exp2Degree<VOLUME>::value>(x);
        run.vsum[j] += e[j];
        run.bidx[j] = (x > run.bmax[j]) ? node : run.bidx[j];   // strict: first node wins
        run.bmax[j] = max_keep(run.bmax[j], x);
    }
    if (VOLUME) {
        if (a.marginal != nullptr) {
            // marginalise over time instead of storing: sum of this node's coalescence over the
            // samples [m0, m1) that fall into this tile (event.trim2window + np.sum(axis=-1),
            // quakemigrate/io/event.py:421-439, signal/scan.py:720)
            double m = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int t = t_first + lane + kWave * j;
                m += (t >= a.m0 && t < a.m1) ? e[j] : 0.0;
            }
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1)   // same order as wave_sum_to_last_lane
                m += __shfl_xor(m, off, kWave);
            if (lane == 0) a.marginal[(int64_t)(t_first / (kWave * J)) * a.n_nodes + node] = m;
        } else {
            double *row = a.volume + (int64_t)node * a.vol_stride + (t_first + lane);
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (t_first + lane + kWave * j < a.n_chunk) row[kWave * j] = e[j];
        }
    }
}

// ACCUM: start from the volume's current content (the reference's `+=`, migratelib.c:57); only
// the generic kernels are built with it (the engine routes accumulate requests there).
template <int J, bool ACCUM>
__device__ __forceinline__ void start_node(const StackArgs &a, double (&acc)[J], int node,
                                           int t_first, int lane) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
        acc[j] = 0.0;
        if (ACCUM) {
            const int t = t_first + lane + kWave * j;
            if (a.accumulate && t < a.n_chunk) acc[j] = a.volume[(int64_t)node * a.vol_stride + t];
        }
    }
}

// cross-wave combine through LDS and write of this workgroup's partial set.
template <int J>
__device__ __forceinline__ void publish(const StackArgs &a, Running<J> &run, double *lds,
                                        int wave, int nwaves, int lane, int t_first, int set) {
    constexpr int KT = kWave * J;
    double *smax = lds;
    double *ssum = lds + (size_t)nwaves * KT;
    int64_t *sidx = reinterpret_cast<int64_t *>(lds + (size_t)2 * nwaves * KT);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = wave * KT + j * kWave + lane;
        smax[k] = run.vmax[j];
        ssum[k] = run.vsum[j];
        sidx[k] = run.vidx[j] == INT32_MAX ? kNoIndex : (int64_t)run.vidx[j];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < KT; k += blockDim.x) {
        double best = smax[k], total = ssum[k];
        int64_t bi = sidx[k];
        for (int w = 1; w < nwaves; ++w) {
            const double v = smax[w * KT + k];
            const int64_t i = sidx[w * KT + k];
            total += ssum[w * KT + k];
            if (better(v, i, best, bi)) {
                best = v;
                bi = i;
            }
        }
        const int t = t_first + k;
        if (t < a.n_chunk) {
            const int64_t o = (int64_t)set * (a.part_stride ? a.part_stride : a.n_chunk) + t;
            a.part_max[o] = best;
            a.part_idx[o] = bi;
            a.part_sum[o] = total;
        }
    }
}

// ---------------------------------------------------------------------------------------
// LDS-tiled stacking kernel.  Workgroup = (time tile, brick group); loops over the bricks of
// its group; wavefront w takes brick nodes w, w+nwaves, ... (walk order).
//
// The inner loop is the generated, hand-scheduled asm of qm_ring_asm.inc (8 table rows per
// statement, 8 ds_read_b64 in flight, ascending row order per sample).  Around it the compiler
// only forms LDS addresses: lane column + chunk base + the row's 16-bit byte offset, which is
// prefetched 8 rows at a time with a VECTOR load (vmcnt) -- a scalar load would share lgkmcnt
// with the LDS reads and force every wait down to 0.
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) double lds_f64;

#include "qm_ring_asm.inc"

__device__ __forceinline__ void unpack8(const uint4 &q, unsigned base, unsigned (&addr)[8]) {
    addr[0] = base + (q.x & 0xffffu); addr[1] = base + (q.x >> 16);
    addr[2] = base + (q.y & 0xffffu); addr[3] = base + (q.y >> 16);
    addr[4] = base + (q.z & 0xffffu); addr[5] = base + (q.z >> 16);
    addr[6] = base + (q.w & 0xffffu); addr[7] = base + (q.w >> 16);
}

// vector (not scalar) 16-byte load of 8 packed offsets: a zero made opaque to the compiler is
// added as the 32-bit vector offset, so the address is "uniform base + VGPR offset" (the saddr
// form of global_load: no 64-bit address arithmetic in the VALU) and cannot be turned into an
// s_load.
__device__ __forceinline__ uint4 load_offsets(const uint16_t *rel, int64_t entry) {
    unsigned zero = 0;
    asm volatile("" : "+v"(zero));
    const char *base = reinterpret_cast<const char *>(rel + entry);
    return *reinterpret_cast<const uint4 *>(base + zero);
}

// ---------------------------------------------------------------------------------------
// Stage the windows of brick b for the tile starting at t_first: row r occupies LDS doubles
// [off_r + r*KT, off_r + r*KT + span_r + KT).  Lane r fetches row r's (min, span, off) record
// with one coalesced 16-byte load; a wave then copies its rows with J+1 independent loads in
// flight per pass (a pass covers KT + 64 samples, i.e. the whole row unless its span > 64).
// ---------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void stage_windows(const StackArgs &a, double *win, int b, int wave,
                                              int nwaves, int lane, int t_first) {
    constexpr int KT = kWave * J;
    constexpr int U = J + 1;
    const int S = a.g.n_rows;
    for (int r0 = 0; r0 < S; r0 += kWave) {
        int4 rec = make_int4(0, 0, 0, 0);
        if (r0 + lane < S)
            rec = reinterpret_cast<const int4 *>(a.brick_meta)[(int64_t)b * S + r0 + lane];
        const int rend = (S - r0 < kWave) ? S - r0 : kWave;
        for (int k = wave; k < rend; k += nwaves) {
            const int r = r0 + k;
            const int lo = __builtin_amdgcn_readlane(rec.x, k);
            const int len = __builtin_amdgcn_readlane(rec.y, k) + KT;
            const int dst = __builtin_amdgcn_readlane(rec.z, k) + r * KT;
            const int first = lo + a.fsmp + a.sample0 + t_first;   // index inside the row
            const int room = a.T - first;                          // readable from `first`
            const double *src = a.onsets + (int64_t)r * a.T + first;
            for (int u0 = 0; u0 < len; u0 += kWave * U) {
                double v[U];
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int u = u0 + kWave * i + lane;
                    v[i] = (u < len && u < room) ? src[u] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int u = u0 + kWave * i + lane;
                    if (u < len) win[dst + u] = v[i];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Software-pipelined node loop (NCH > 0 kernels).  While the rows of node B stream out of LDS,
// the epilogue of the previous node A (exp, sum, running maximum: ~22 VALU ops per sample slot)
// is issued in slices between "issue next batch of ds_read_b64" and "wait + add this batch", so
// a wavefront has LDS requests in flight while it does its VALU-only work.  This part is plain
// C++: the LDS loads are volatile (not fused into ds_read2st64_b64, order kept), the compiler
// counts them (s_waitcnt lgkmcnt(N)), and sched_barrier pins the three phases of each batch.
// ---------------------------------------------------------------------------------------
template <int J>
struct Epilogue {              // node whose sums are complete but not yet exponentiated
    double x[J], f[J], p[J];
    int k[J];
    int node;
    double *row;               // volume row of the node + first sample of the tile (wave-uniform)
};

// steps: 0 z | 1 k | 2 f | 3..3+D-1 Horner | then ldexp | sum(+store) | track
template <bool VOLUME> struct EpiSteps { static constexpr int value = 3 + Exp2Degree<VOLUME>::value + 3; };

template <int J, bool VOLUME, int STEP>
__device__ __forceinline__ void epi_step(Epilogue<J> &s, Running<J> &run, const StackArgs &a,
                                         int t_first, int lane) {
    constexpr int D = Exp2Degree<VOLUME>::value;
    constexpr int H0 = 3, H1 = 3 + D;                  // Horner steps [H0, H1)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if constexpr (STEP == 0) {                                     // z = stack * log2e/avail
#pragma clang fp contract(off)                                         // rounded (see finish_node)
            s.x[j] = s.x[j] * a.z_scale;
        }
        else if constexpr (STEP == 1) s.p[j] = __builtin_rint(s.x[j]); // k (as double)
        else if constexpr (STEP == 2) {
            s.f[j] = s.x[j] - s.p[j];                                  // f = z - k
            s.k[j] = (int)s.p[j];
        } else if constexpr (STEP >= H0 && STEP < H1)
            s.p[j] = exp2_horner<D, STEP - H0>(s.p[j], s.f[j]);
        else if constexpr (STEP == H1) s.p[j] = __builtin_amdgcn_ldexp(s.p[j], s.k[j]);
        else if constexpr (STEP == H1 + 1) {
            run.vsum[j] += s.p[j];
            if (VOLUME) {
                const int t = t_first + lane + kWave * j;
                if (t < a.n_chunk) a.volume[(int64_t)s.node * a.vol_stride + t] = s.p[j];
            }
        } else if constexpr (STEP == H1 + 2) {
            const bool gt = s.x[j] > run.bmax[j];                      // strict: first node wins
            run.bidx[j] = gt ? s.node : run.bidx[j];
            run.bmax[j] = max_keep(run.bmax[j], s.x[j]);
        }
    }
}

template <int J, bool VOLUME, int FIRST, int LAST>
__device__ __forceinline__ void epi_steps(Epilogue<J> &s, Running<J> &run, const StackArgs &a,
                                          int t_first, int lane) {
    if constexpr (FIRST < LAST) {
        epi_step<J, VOLUME, FIRST>(s, run, a, t_first, lane);
        epi_steps<J, VOLUME, FIRST + 1, LAST>(s, run, a, t_first, lane);
    }
}

// byte offset of table row r (0..7) inside one packed 16-byte chunk
__device__ __forceinline__ unsigned chunk_entry(const uint4 &q, int r) {
    const unsigned w = (r < 2) ? q.x : (r < 4) ? q.y : (r < 6) ? q.z : q.w;
    return (r & 1) ? (w >> 16) : (w & 0xffffu);
}

template <int J> struct BatchRows { static constexpr int value = (J == 1) ? 4 : (J == 2) ? 2 : 1; };

template <int J, int RB>
__device__ __forceinline__ void issue_rows(double (&buf)[RB * J], const uint4 &q,
                                           unsigned chunk_addr, int r0) {
    constexpr int KT = kWave * J;
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        const volatile lds_f64 *p =
            (const volatile lds_f64 *)(uintptr_t)(chunk_addr + chunk_entry(q, r0 + k));
#pragma unroll
        for (int j = 0; j < J; ++j) buf[k * J + j] = p[(r0 + k) * KT + kWave * j];
    }
}

template <int J, int RB>
__device__ __forceinline__ void retire_rows(double (&acc)[J], const double (&buf)[RB * J]) {
#pragma unroll
    for (int k = 0; k < RB; ++k)                       // ascending row order per sample
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] += buf[k * J + j];
}

// One batch of the software pipeline (compile-time index I of NB): issue batch I+1, run this
// batch's share of the previous node's epilogue, then wait for and add batch I.  As soon as a
// chunk's last row has been issued its offset registers are refilled with the NEXT node's chunk.
template <int J, bool VOLUME, int NCH, bool WITH_EPI, int I>
__device__ __forceinline__ void pipeline_batch(double (&acc)[J],
                                               double (&even)[BatchRows<J>::value * J],
                                               double (&odd)[BatchRows<J>::value * J],
                                               uint4 (&q)[NCH], const uint16_t *next,
                                               unsigned lane_addr, Epilogue<J> &epi,
                                               Running<J> &run, const StackArgs &a, int t_first,
                                               int lane) {
    constexpr int KT = kWave * J;
    constexpr int RB = BatchRows<J>::value;
    constexpr int NB = (NCH - 1) * 8 / RB;              // batches per node (full chunks only)
    constexpr int BPC = 8 / RB;                         // batches per chunk
    if constexpr (I < NB) {
        if constexpr (I + 1 < NB) {
            constexpr int ci = (I + 1) / BPC, bj = (I + 1) % BPC;
            const unsigned ca = lane_addr + (unsigned)(ci * 8 * KT * 8);
            issue_rows<J, RB>((I & 1) ? even : odd, q[ci], ca, bj * RB);
            if constexpr (bj == BPC - 1) q[ci] = load_offsets(next, ci * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WITH_EPI) {
            constexpr int E = EpiSteps<VOLUME>::value;
            epi_steps<J, VOLUME, I * E / NB, (I + 1) * E / NB>(epi, run, a, t_first, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        retire_rows<J, RB>(acc, (I & 1) ? odd : even);
        __builtin_amdgcn_sched_barrier(0);
        pipeline_batch<J, VOLUME, NCH, WITH_EPI, I + 1>(acc, even, odd, q, next, lane_addr, epi,
                                                        run, a, t_first, lane);
    }
}

// Stack the (NCH-1)*8 rows of the full chunks of one node (offsets in q[0..NCH-2]); if WITH_EPI,
// the previous node's epilogue `epi` is interleaved.
template <int J, bool VOLUME, int NCH, bool WITH_EPI>
__device__ __forceinline__ void stack_full_chunks(double (&acc)[J], uint4 (&q)[NCH],
                                                  const uint16_t *next, unsigned lane_addr,
                                                  Epilogue<J> &epi, Running<J> &run,
                                                  const StackArgs &a, int t_first, int lane) {
    constexpr int RB = BatchRows<J>::value;
    if constexpr (NCH == 1) {
        if constexpr (WITH_EPI)
            epi_steps<J, VOLUME, 0, EpiSteps<VOLUME>::value>(epi, run, a, t_first, lane);
    } else {
        double even[RB * J], odd[RB * J];
        issue_rows<J, RB>(even, q[0], lane_addr, 0);
        pipeline_batch<J, VOLUME, NCH, WITH_EPI, 0>(acc, even, odd, q, next, lane_addr, epi, run,
                                                    a, t_first, lane);
    }
}

// NCH > 0: the node's offsets are exactly NCH 16-byte chunks (row_pad == 8*NCH); the whole next
// node is prefetched into registers while the current one is stacked, and the chunk loop is
// unrolled.  NCH == 0: any row count, one-chunk-ahead prefetch (slower, always valid).
template <int J, bool VOLUME, int NCH>
__global__ __launch_bounds__(1024) void stack_lds_kernel(StackArgs a_launch) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    const StackArgs a = step_view(a_launch);
    constexpr int KT = kWave * J;
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    // XCD-aware workgroup -> (time tile, brick group) map: workgroup b is observed to run on XCD
    // b % 8, each XCD has its own L2; all time tiles of one brick group read the same slice of the
    // offset table, so whole groups are dealt to XCDs (group = xcd + 8 * k) and a slice is fetched
    // into one L2 instead of eight.  Purely a placement hint: any mapping gives the same result.
    int tile, group;
    stack_tile_group(a, tile, group);
    if (group >= a.ngroups) return;                   // grid is padded to a multiple of 8 groups
    if (a.run_if != nullptr && *a.run_if == 0) return;
    const int t_first = tile * KT;                    // relative to sample0
    const int S = g.n_rows;
    const int nfull = S >> 3, ntail = S & 7;
    const int nchunks = g.row_pad >> 3;               // nfull + (ntail != 0)
    // LDS byte address of this lane's column in the window area
    const unsigned lane_addr = (unsigned)(uintptr_t)((lds_f64 *)win) + (unsigned)lane * 8u;

    Running<J> run;
    run.reset();

    for (int b = group; b < g.nbricks; b += a.ngroups) {
        if (!brick_fits(a.brick_total[b], S, KT, a.cap_doubles)) continue;   // direct kernel's job
        __syncthreads();                              // previous brick fully consumed
        stage_windows<J>(a, win, b, wave, nwaves, lane, t_first);
        __syncthreads();

        // ---- this wave's walk over the brick's valid nodes (no divisions inside the loop)
        int x0, y0, z0, vx, vy, vz;
        brick_extents(g, b, x0, y0, z0, vx, vy, vz);
        const int nvalid = vx * vy * vz;
        int lz = wave % vz, ly = (wave / vz) % vy, lx = wave / (vz * vy);
        const uint16_t *brick_rel = a.rel + (int64_t)b * g.brick_nodes * g.row_pad;

        if constexpr (NCH > 0 && VOLUME) {
            // Volume-writing variant: asm ring + epilogue in place.  The next node's offsets
            // are loaded BEFORE this node's stores are issued: gfx9 has one vmcnt for loads and
            // stores, so a load issued after the stores could only be waited for together with
            // them (an HBM write round trip per chunk).
            uint4 qn[NCH];
            {
                const uint16_t *p = brick_rel + (int64_t)(wave < nvalid ? wave : 0) * g.row_pad;
#pragma unroll
                for (int c = 0; c < NCH; ++c) qn[c] = load_offsets(p, c * 8);
            }
            const int last_rows = S - 8 * (NCH - 1);
            for (int m = wave; m < nvalid; m += nwaves) {
                const int node = ((x0 + lx) * g.ny + (y0 + ly)) * g.nz + (z0 + lz);
                lz += nwaves;
                while (lz >= vz) { lz -= vz; ++ly; }
                while (ly >= vy) { ly -= vy; ++lx; }
                uint4 qc[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) qc[c] = qn[c];
                {
                    const uint16_t *p =
                        brick_rel + (int64_t)(m + nwaves < nvalid ? m + nwaves : m) * g.row_pad;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) qn[c] = load_offsets(p, c * 8);
                }
                double acc[J];
                start_node<J, false>(a, acc, node, t_first, lane);
                unsigned addr[8];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    unpack8(qc[c], lane_addr + (unsigned)(c * 8 * KT * 8), addr);
                    if (c + 1 < NCH || last_rows == 8) ring_full<J>(acc, addr);
                    else ring_tail<J>(acc, addr, last_rows);
                }
                finish_node<J, VOLUME>(a, run, acc, node, t_first, lane);
            }
        } else if constexpr (NCH > 0) {
            uint4 q[NCH];                              // offsets of the node about to be stacked
            {
                const uint16_t *p = brick_rel + (int64_t)(wave < nvalid ? wave : 0) * g.row_pad;
#pragma unroll
                for (int c = 0; c < NCH; ++c) q[c] = load_offsets(p, c * 8);
            }
            const int last_rows = S - 8 * (NCH - 1);
            Epilogue<J> epi;
            bool pending = false;                      // wave-uniform: epi holds a node
            for (int m = wave; m < nvalid; m += nwaves) {
                const int node = ((x0 + lx) * g.ny + (y0 + ly)) * g.nz + (z0 + lz);
                lz += nwaves;
                while (lz >= vz) { lz -= vz; ++ly; }
                while (ly >= vy) { ly -= vy; ++lx; }
                // the node after this one (or a harmless reload of this one at the end)
                const uint16_t *next =
                    brick_rel + (int64_t)(m + nwaves < nvalid ? m + nwaves : m) * g.row_pad;

                double acc[J];
                start_node<J, false>(a, acc, node, t_first, lane);
                if (pending)
                    stack_full_chunks<J, VOLUME, NCH, true>(acc, q, next, lane_addr, epi, run, a,
                                                            t_first, lane);
                else
                    stack_full_chunks<J, VOLUME, NCH, false>(acc, q, next, lane_addr, epi, run,
                                                             a, t_first, lane);
                {   // last chunk (1..8 rows): generated asm, drains the LDS queue
                    unsigned addr[8];
                    unpack8(q[NCH - 1], lane_addr + (unsigned)((NCH - 1) * 8 * KT * 8), addr);
                    q[NCH - 1] = load_offsets(next, (NCH - 1) * 8);
                    if (last_rows == 8) ring_full<J>(acc, addr);
                    else ring_tail<J>(acc, addr, last_rows);
                }
#pragma unroll
                for (int j = 0; j < J; ++j) epi.x[j] = acc[j];
                epi.node = node;
                pending = true;
            }
            if (pending)                               // the brick's last node: not overlapped
                epi_steps<J, VOLUME, 0, EpiSteps<VOLUME>::value>(epi, run, a, t_first, lane);
        } else {
            // offset-chunk prefetch: one 8-row chunk ahead of consumption, in this wave's order
            int pm = wave, pc = 0;
            auto fetch_next = [&]() -> uint4 {
                const int mm = pm < nvalid ? pm : 0;   // past the end: harmless reload
                const uint4 v = load_offsets(brick_rel, (int64_t)mm * g.row_pad + pc * 8);
                if (++pc == nchunks) { pc = 0; pm += nwaves; }
                return v;
            };
            uint4 q = fetch_next();
            for (int m = wave; m < nvalid; m += nwaves) {
                const int node = ((x0 + lx) * g.ny + (y0 + ly)) * g.nz + (z0 + lz);
                lz += nwaves;
                while (lz >= vz) { lz -= vz; ++ly; }
                while (ly >= vy) { ly -= vy; ++lx; }

                double acc[J];
                start_node<J, VOLUME>(a, acc, node, t_first, lane);
                unsigned chunk_addr = lane_addr;
                unsigned addr[8];
                for (int c = 0; c < nfull; ++c) {
                    unpack8(q, chunk_addr, addr);
                    q = fetch_next();
                    ring_full<J>(acc, addr);
                    chunk_addr += 8 * KT * 8;
                }
                if (ntail) {
                    unpack8(q, chunk_addr, addr);
                    q = fetch_next();
                    ring_tail<J>(acc, addr, ntail);
                }
                finish_node<J, VOLUME>(a, run, acc, node, t_first, lane);
            }
        }
        run.merge_brick();
    }
    if (a.want_scan) publish<J>(a, run, win, wave, nwaves, lane, t_first, a.set0 + group);
}

// ---------------------------------------------------------------------------------------
// Exact-row-count variant of the LDS-tiled kernel (the table's row count S is a template
// parameter): the software pipeline of stack_full_chunks covers ALL rows of a node -- no
// hand-off to the asm ring for the last partial chunk, hence no LDS queue drain in the middle of
// a node and no register copies around an asm statement; the first row lands in the accumulators
// directly (0.0 + x == x, so the reference's `+=` from a zeroed volume is reproduced bit for bit
// without the add); the previous node's epilogue -- including, when VOLUME, its stores -- is
// spread over the rows of the current node.  Same operands, same ascending row order per sample,
// same partial sets as stack_lds_kernel: the two are interchangeable (tests run both).
// ---------------------------------------------------------------------------------------
// Store of the lanes with `on` set, without a branch: EXEC is narrowed and restored inside one asm
// statement.  `row` is wave-uniform (scalar base), `u` the lane's offset in doubles.  (A branch in
// the node body makes LLVM sink the adds of all rows below it: the operands then live in scratch.)
// The compiler's s_waitcnt bookkeeping does not see these stores; its waits can only come out
// stricter than needed for that (the counter is in-order), never too weak.
__device__ __forceinline__ void store_one_masked(double *row, int u, double val, bool on) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(on);
    const unsigned off = (unsigned)u * 8u;
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\t"
                 "global_store_dwordx2 %2, %3, %4 nt\n\ts_mov_b64 exec, %0\n\ts_nop 0"
                 : "=&s"(save)
                 : "s"(mask), "v"(off), "v"(val), "s"(row)
                 : "memory");
}

// Sum over the wavefront without LDS traffic: DPP moves + adds in the order of an xor butterfly
// with strides 1, 2, 4, 8, 16, 32 (row_half_mirror / row_mirror deliver the partner group's sum
// because the sums are uniform within a group by then; the two broadcasts add the row sums in the
// butterfly's pairing, a + b == b + a) -- the total, bit-identical to that butterfly's, lands in
// lane 63.  Other lanes hold partial sums.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_moved(double x) {     // rows outside ROW_MASK receive 0.0
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_last_lane(double m) {
    m += dpp_moved<0xB1, 0xf>(m);                      // quad_perm [1,0,3,2]
    m += dpp_moved<0x4E, 0xf>(m);                      // quad_perm [2,3,0,1]
    m += dpp_moved<0x141, 0xf>(m);                     // row_half_mirror
    m += dpp_moved<0x140, 0xf>(m);                     // row_mirror
    m += dpp_moved<0x142, 0xa>(m);                     // row_bcast:15 into rows 1, 3
    m += dpp_moved<0x143, 0xc>(m);                     // row_bcast:31 into rows 2, 3
    return m;
}

__host__ __device__ constexpr int exact_nch(int S) { return (S + 7) / 8; }   // offset chunks per node
// Offset chunks held in registers.  Normally all of a node's chunks: a chunk is refilled with the
// NEXT node's offsets as soon as its last row has been issued (a whole node of prefetch distance
// at no extra registers).  Four samples per lane beyond 40 rows would need 6-8 chunks on top of
// twice the accumulators and read buffers and spill them inside the node loop (measured: 38 GB of
// scratch traffic per C3-sized step of 60 rows, 114.9 ms; with the ring 107.3 ms and none): there
// a ring of 4 chunks is kept, chunk c in slot c % 4,
// refilled four chunks ahead -- from the same node, or from the next one once past the end (the
// chunk count is padded to a multiple of 4 for that; the padding's refills ride on the last row).
__host__ __device__ constexpr int exact_ring(int J, int S) {
    return (J == 4 && S > 40) ? 4 : exact_nch(S);
}
template <int J, int S> struct ExactPlan {
    static constexpr int RB = BatchRows<J>::value;          // table rows per batch
    static constexpr int NB = (S + RB - 1) / RB;            // batches per node
    static constexpr int NCH = exact_nch(S);
    static constexpr int R = exact_ring(J, S);              // chunks in registers
    static constexpr int NCHP = (NCH + R - 1) / R * R;      // chunk count padded to the ring
};

// steps of the pipelined epilogue: 0 k | 1 f | 2..D+1 Horner | ldexp | sum | track | (store)
template <bool VOLUME> struct XEpiSteps {
    static constexpr int value = 2 + Exp2Degree<VOLUME>::value + 3 + (VOLUME ? 1 : 0);
};

template <int J, bool VOLUME, int TAIL, int STEP>
__device__ __forceinline__ void xepi_step(Epilogue<J> &s, Running<J> &run, const StackArgs &a,
                                          int t_first, int lane) {
    constexpr int D = Exp2Degree<VOLUME>::value;
    constexpr int H0 = 2, H1 = 2 + D;                  // Horner steps [H0, H1)
    if constexpr (VOLUME && TAIL == 3 && STEP == H1 + 3) {
        // TAIL 3, the marginalised map instead of the volume: the node's coalescence summed over
        // the samples of [m0, m1) that fall into this tile (event.trim2window + np.sum(axis=-1),
        // quakemigrate/io/event.py:421-439, signal/scan.py:720) -- per lane over its J samples,
        // then over the wavefront; lane 63 stores the tile's share of the node (s.row is the
        // element's address; no branch: see store_one_masked).  Same order of additions as
        // finish_node: the chunked kernels give the same bits.
        double m = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int t = t_first + lane + kWave * j;
            m += (t >= a.m0 && t < a.m1) ? s.p[j] : 0.0;
        }
        m = wave_sum_to_last_lane(m);
        store_one_masked(s.row, 0, m, lane == kWave - 1);
        return;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if constexpr (STEP == 0) s.p[j] = __builtin_rint(s.x[j]);      // k (as double)
        else if constexpr (STEP == 1) {
            s.f[j] = s.x[j] - s.p[j];                                  // f = z - k
            s.k[j] = (int)s.p[j];
        } else if constexpr (STEP >= H0 && STEP < H1)
            s.p[j] = exp2_horner<D, STEP - H0>(s.p[j], s.f[j]);
        else if constexpr (STEP == H1) s.p[j] = __builtin_amdgcn_ldexp(s.p[j], s.k[j]);
        else if constexpr (STEP == H1 + 1) run.vsum[j] += s.p[j];
        else if constexpr (STEP == H1 + 2) {
            const bool gt = s.x[j] > run.bmax[j];                      // strict: first node wins
            run.bidx[j] = gt ? s.node : run.bidx[j];
            run.bmax[j] = max_keep(run.bmax[j], s.x[j]);
        } else if constexpr (VOLUME && STEP == H1 + 3) {
            // The stores come last: every offset load of the node being stacked has been issued
            // by now, so no later wait on such a load has to sit out these stores as well (loads
            // and stores share one in-order counter on gfx9).  s.row is wave-uniform: scalar base
            // + 32-bit lane offset.  TAIL 0, a full tile: unconditional.  TAIL 1, the scan's last
            // tile, pulled back over its predecessor: the samples of the overlap were written by
            // that tile already and are masked out.  TAIL 2, a scan shorter than one tile: the
            // lanes past the end are masked out.
            const int u = lane + kWave * j;
            if constexpr (TAIL == 0) __builtin_nontemporal_store(s.p[j], s.row + u);
            else if constexpr (TAIL == 1)
                store_one_masked(s.row, u, s.p[j], u >= kWave * J - a.n_chunk % (kWave * J));
            else store_one_masked(s.row, u, s.p[j], t_first + u < a.n_chunk);
        }
    }
}

template <int J, bool VOLUME, int TAIL, int FIRST, int LAST>
__device__ __forceinline__ void xepi_steps(Epilogue<J> &s, Running<J> &run, const StackArgs &a,
                                           int t_first, int lane) {
    if constexpr (FIRST < LAST) {
        xepi_step<J, VOLUME, TAIL, FIRST>(s, run, a, t_first, lane);
        xepi_steps<J, VOLUME, TAIL, FIRST + 1, LAST>(s, run, a, t_first, lane);
    }
}

// issue the LDS reads of batch I (rows I*RB ..) of the node whose offsets are in q; a chunk of
// q is refilled with the NEXT node's offsets as soon as its last row has been issued
template <int J, int S, int I>
__device__ __forceinline__ void xissue(double (&buf)[BatchRows<J>::value * J],
                                       uint4 (&q)[exact_ring(J, S)], const uint16_t *cur,
                                       const uint16_t *next, unsigned lane_addr) {
    constexpr int KT = kWave * J;
    constexpr int RB = ExactPlan<J, S>::RB;
    constexpr int NCH = ExactPlan<J, S>::NCH, R = ExactPlan<J, S>::R;
    constexpr int NCHP = ExactPlan<J, S>::NCHP;
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        const int r = I * RB + k;                      // compile-time after unrolling
        if (r < S) {
            const int ci = r >> 3, e = r & 7;
            const volatile lds_f64 *p = (const volatile lds_f64 *)(uintptr_t)(
                lane_addr + (unsigned)(ci * 8 * KT * 8) + chunk_entry(q[ci % R], e));
#pragma unroll
            for (int j = 0; j < J; ++j) buf[k * J + j] = p[e * KT + kWave * j];
            if constexpr (R == NCH) {
                // (the last chunk is refilled by the node loop: a load issued this late would be
                // waited for at once, by the register copies at the loop's back edge)
                if (e == 7 && ci + 1 < NCH) q[ci] = load_offsets(next, ci * 8);
            } else if (e == 7 && r != S - 1) {         // the chunk's last row: its slot is free
                if (ci + R < NCH) q[ci % R] = load_offsets(cur, (ci + R) * 8);
                else if (ci + R >= NCHP) q[ci % R] = load_offsets(next, (ci + R - NCHP) * 8);
                // (the refills that fall on the node's LAST row -- its last chunk's and the
                // padding chunks' -- are issued by the node loop at the start of the next node: a
                // load issued at the end of the loop body is waited for at once, by the register
                // copies at the back edge)
            }
        }
    }
}

template <int J, int S, int I>
__device__ __forceinline__ void xretire(double (&acc)[J],
                                        const double (&buf)[BatchRows<J>::value * J]) {
    constexpr int RB = ExactPlan<J, S>::RB;
#pragma unroll
    for (int k = 0; k < RB; ++k) {                     // ascending row order per sample
        const int r = I * RB + k;
        if (r < S) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (r == 0) acc[j] = buf[k * J + j];   // 0.0 + x, without the add
                else acc[j] += buf[k * J + j];
            }
        }
    }
}

template <int J, bool VOLUME, int TAIL, int S, bool WITH_EPI, int I>
__device__ __forceinline__ void xbatch(double (&acc)[J],
                                       double (&even)[BatchRows<J>::value * J],
                                       double (&odd)[BatchRows<J>::value * J],
                                       uint4 (&q)[exact_ring(J, S)], const uint16_t *cur,
                                       const uint16_t *next,
                                       unsigned lane_addr, Epilogue<J> &epi, Running<J> &run,
                                       const StackArgs &a, int t_first, int lane) {
    constexpr int NB = ExactPlan<J, S>::NB;
    if constexpr (I < NB) {
        if constexpr (I + 1 < NB)
            xissue<J, S, I + 1>((I & 1) ? even : odd, q, cur, next, lane_addr);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WITH_EPI) {
            constexpr int E = XEpiSteps<VOLUME>::value;
            xepi_steps<J, VOLUME, TAIL, I * E / NB, (I + 1) * E / NB>(epi, run, a, t_first, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        xretire<J, S, I>(acc, (I & 1) ? odd : even);
        __builtin_amdgcn_sched_barrier(0);
        xbatch<J, VOLUME, TAIL, S, WITH_EPI, I + 1>(acc, even, odd, q, cur, next, lane_addr, epi,
                                                    run, a, t_first, lane);
    }
}

template <int J, bool VOLUME, int TAIL, int S>
__device__ __forceinline__ void stack_exact_body(const StackArgs &a, double *win) {
    constexpr int KT = kWave * J;
    constexpr int NCH = ExactPlan<J, S>::NCH;
    constexpr int RB = ExactPlan<J, S>::RB;
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    // XCD-aware workgroup -> (time tile, brick group) map, as stack_lds_kernel
    int tile, group;
    stack_tile_group(a, tile, group);
    // TAIL 1: the last tile of a scan that is not a multiple of the tile length is pulled back so
    // that it ends with the scan (it overlaps its predecessor; both compute the overlap with the
    // same arithmetic and write the same bits to the partial sets): no lane is past the end
    const int t_first = TAIL == 1 ? a.n_chunk - KT : tile * KT;
    const unsigned lane_addr = (unsigned)(uintptr_t)((lds_f64 *)win) + (unsigned)lane * 8u;

    Running<J> run;
    run.reset();

    for (int b = group; b < g.nbricks; b += a.ngroups) {
        if (!brick_fits(a.brick_total[b], S, KT, a.cap_doubles)) continue;   // direct kernel's job
        __syncthreads();                              // previous brick fully consumed
        stage_windows<J>(a, win, b, wave, nwaves, lane, t_first);
        __syncthreads();

        int x0, y0, z0, vx, vy, vz;
        brick_extents(g, b, x0, y0, z0, vx, vy, vz);
        const int nvalid = vx * vy * vz;
        int lz = wave % vz, ly = (wave / vz) % vy, lx = wave / (vz * vy);
        const uint16_t *brick_rel = a.rel + (int64_t)b * g.brick_nodes * g.row_pad;

        constexpr int R = ExactPlan<J, S>::R;
        uint4 q[R];                                    // offsets of the node about to be stacked
        {                                              //   (all its chunks, or the first R)
            const uint16_t *p = brick_rel + (int64_t)(wave < nvalid ? wave : 0) * g.row_pad;
#pragma unroll
            for (int c = 0; c < R; ++c) q[c] = load_offsets(p, c * 8);
        }
        Epilogue<J> epi;
        bool pending = false;                          // wave-uniform: epi holds a node
        for (int m = wave; m < nvalid; m += nwaves) {
            const int node = ((x0 + lx) * g.ny + (y0 + ly)) * g.nz + (z0 + lz);
            lz += nwaves;
            while (lz >= vz) { lz -= vz; ++ly; }
            while (ly >= vy) { ly -= vy; ++lx; }
            // the node after this one (or a harmless reload of this one at the end)
            const uint16_t *cur = brick_rel + (int64_t)m * g.row_pad;
            const uint16_t *next =
                brick_rel + (int64_t)(m + nwaves < nvalid ? m + nwaves : m) * g.row_pad;

            // the next node's last offset chunk, fetched a whole node ahead of its use (when all
            // chunks of a node are held)
            uint4 q_last;
            if constexpr (R == NCH) q_last = load_offsets(next, (NCH - 1) * 8);
            else {
                // ring: the refills the previous node's last row left to this one (for a brick's
                // first node they reload what the initial fill put there)
                constexpr int NCHP = ExactPlan<J, S>::NCHP;
#pragma unroll
                for (int d = NCH - 1; d < NCHP; ++d) q[d % R] = load_offsets(cur, (d + R - NCHP) * 8);
            }
            double acc[J], even[RB * J], odd[RB * J];
            xissue<J, S, 0>(even, q, cur, next, lane_addr);
            if (pending)
                xbatch<J, VOLUME, TAIL, S, true, 0>(acc, even, odd, q, cur, next, lane_addr, epi,
                                                    run, a, t_first, lane);
            else
                xbatch<J, VOLUME, TAIL, S, false, 0>(acc, even, odd, q, cur, next, lane_addr, epi,
                                                     run, a, t_first, lane);
#pragma unroll
            for (int j = 0; j < J; ++j) {                  // z: log2 of the coalescence (rounded
#pragma clang fp contract(off)                             // product: see finish_node)
                epi.x[j] = acc[j] * a.z_scale;
            }
            epi.node = node;
            if (VOLUME && TAIL == 3)
                epi.row = a.marginal + ((int64_t)tile * a.n_nodes + node);
            else if (VOLUME) epi.row = a.volume + ((int64_t)node * a.vol_stride + t_first);
            if constexpr (R == NCH) q[NCH - 1] = q_last;
            pending = true;
        }
        if (pending)                                   // the brick's last node: not overlapped
            xepi_steps<J, VOLUME, TAIL, 0, XEpiSteps<VOLUME>::value>(epi, run, a, t_first, lane);
        run.merge_brick();
    }
    if (a.want_scan) publish<J>(a, run, win, wave, nwaves, lane, t_first, a.set0 + group);
}

// the marginalised map of a locate window instead of its volume (a.marginal: [ntiles][n_nodes])
template <int J, int S>
__global__ __launch_bounds__(1024) void stack_exact_marginal_kernel(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    int tile, group;
    stack_tile_group(a, tile, group);
    if (group >= a.ngroups) return;                   // grid is padded to a multiple of 8 groups
    if (a.run_if != nullptr && *a.run_if == 0) return;
    stack_exact_body<J, true, 3, S>(a, win);
}

template <int J, bool VOLUME, int S>
__global__ __launch_bounds__(1024) void stack_exact_kernel(StackArgs a_launch) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    constexpr int KT = kWave * J;
    const StackArgs a = step_view(a_launch);
    int tile, group;
    stack_tile_group(a, tile, group);
    if (group >= a.ngroups) return;                   // grid is padded to a multiple of 8 groups
    if (a.run_if != nullptr && *a.run_if == 0) return;
    // only the volume-writing variant cares where its tile lies in the scan
    if (VOLUME && a.n_chunk < KT) stack_exact_body<J, VOLUME, 2, S>(a, win);
    else if (VOLUME && (tile + 1) * KT > a.n_chunk) stack_exact_body<J, VOLUME, 1, S>(a, win);
    else stack_exact_body<J, VOLUME, 0, S>(a, win);
}

// ---------------------------------------------------------------------------------------
// Direct kernel: same decomposition, operands straight from global memory (the log-onset
// array is a few MB and lives in L2 / Infinity Cache).  Handles bricks whose windows do not
// fit the LDS budget (arbitrary tables stay correct), and serves as an on-device cross-check.
// ---------------------------------------------------------------------------------------
template <int J, bool VOLUME>
__global__ __launch_bounds__(1024) void stack_direct_kernel(StackArgs a_launch) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    const StackArgs a = step_view(a_launch);
    constexpr int KT = kWave * J;
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    // XCD-aware workgroup -> (time tile, brick group) map: workgroup b is observed to run on XCD
    // b % 8, each XCD has its own L2; all time tiles of one brick group read the same slice of the
    // offset table, so whole groups are dealt to XCDs (group = xcd + 8 * k) and a slice is fetched
    // into one L2 instead of eight.  Purely a placement hint: any mapping gives the same result.
    int tile, group;
    stack_tile_group(a, tile, group);
    if (group >= a.ngroups) return;                   // grid is padded to a multiple of 8 groups
    if (a.run_if != nullptr && *a.run_if == 0) return;
    const int t_first = tile * KT;
    const int S = g.n_rows;
    const int n_list = a.brick_list ? a.n_list : g.nbricks;

    Running<J> run;
    run.reset();

    int tcl[J];                                         // clamped sample index per lane
#pragma unroll
    for (int j = 0; j < J; ++j) {
        int t = t_first + lane + kWave * j;
        t = t < a.n_chunk ? t : a.n_chunk - 1;          // stay inside the rows
        tcl[j] = t + a.sample0 + a.fsmp;
    }

    for (int i = group; i < n_list; i += a.ngroups) {
        const int b = a.brick_list ? a.brick_list[i] : i;
        int x0, y0, z0, vx, vy, vz;
        brick_extents(g, b, x0, y0, z0, vx, vy, vz);
        const int nvalid = vx * vy * vz;
        for (int m = wave; m < nvalid; m += nwaves) {
            const int node = brick_walk_node(g, x0, y0, z0, vy, vz, m);
            const int32_t *row = a.lut + (int64_t)node * S;
            double acc[J];
            start_node<J, VOLUME>(a, acc, node, t_first, lane);
            for (int r = 0; r < S; ++r) {
                int d = row[r];
                d = d < 0 ? 0 : d;
                const double *p = a.onsets + (int64_t)r * a.T + d;
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] += p[tcl[j]];
            }
            finish_node<J, VOLUME>(a, run, acc, node, t_first, lane);
        }
        run.merge_brick();
    }
    if (a.want_scan) publish<J>(a, run, win, wave, nwaves, lane, t_first, a.set0 + group);
}

// ---------------------------------------------------------------------------------------
// Scan of a materialised volume (find_max_coa).  Workgroup = (up to 16 adjacent 64-sample tiles,
// node chunk); wavefront w owns tile w of the group (lanes <-> samples) and walks EVERY node of the
// chunk in ascending order (strict '>' keeps the first maximum, migratelib.c:102) with 8 loads in
// flight, so the wavefronts of a workgroup together read whole contiguous stretches of each volume
// row, row after row -- one sequential stream per workgroup -- and need no cross-wave combine.
// Values are compared as stored (already exponentiated).  HBM-read bound: 8 bytes per node-sample.
// ---------------------------------------------------------------------------------------
constexpr int kScanWaves = 16;                         // tiles per workgroup, at most
#ifdef QM_TU_STEPS
__global__ __launch_bounds__(kScanWaves * kWave) void scan_volume_kernel(
    const double *__restrict__ vol, int64_t vol_stride, int n_chunk, int64_t n_nodes,
    int64_t nodes_per_set, double *__restrict__ part_max, int64_t *__restrict__ part_idx,
    double *__restrict__ part_sum) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int t = (blockIdx.x * (blockDim.x >> 6) + wave) * kWave + lane;
    if (t - lane >= n_chunk) return;                    // whole wave past the end (no barriers here)
    const int tc = t < n_chunk ? t : n_chunk - 1;       // clamp: keep every lane's loads in range
    const int set = blockIdx.y;
    const int64_t n0 = (int64_t)set * nodes_per_set;
    int64_t n1 = n0 + nodes_per_set;
    n1 = n1 < n_nodes ? n1 : n_nodes;
    double best = -__builtin_inf(), total = 0.0;
    int64_t bi = kNoIndex;
    const double *col = vol + tc;
    int64_t n = n0;
    for (; n + 7 < n1; n += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(col + (n + k) * vol_stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            total += v[k];
            if (v[k] > best) {
                best = v[k];
                bi = n + k;
            }
        }
    }
    for (; n < n1; ++n) {
        const double v = __builtin_nontemporal_load(col + n * vol_stride);
        total += v;
        if (v > best) {
            best = v;
            bi = n;
        }
    }
    if (t < n_chunk) {
        const int64_t o = (int64_t)set * n_chunk + t;
        part_max[o] = best;
        part_idx[o] = bi;
        part_sum[o] = total;
    }
}
#endif  // QM_TU_STEPS

// per-tile marginal sums -> marginal map (fixed tile order: deterministic)
#ifdef QM_TU_STEPS
__global__ void marginal_reduce_kernel(const double *__restrict__ part, int ntiles, int64_t n,
                                       double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int t = 0; t < ntiles; ++t) s += part[(int64_t)t * n + i];
    out[i] = s;
}
#endif  // QM_TU_STEPS

// ---------------------------------------------------------------------------------------
// Combine partial sets ([n_sets][n]): lanes <-> samples, the 4 wavefronts of a workgroup split
// the sets, LDS combine.  mode 0: emit one combined partial (index + node_offset, still in the
// log2 domain); mode 1: final series from log2-domain partials; mode 2: final series from
// partials that already hold coalescence values (volume scan).  Ties -> lowest node index.
// Set s of each array starts at element s * set_stride (n for [n_sets][n] arrays; 3 n for the
// packed [n_sets][3][n] layout of the cross-GPU all-gather).
// ---------------------------------------------------------------------------------------
#ifdef QM_TU_STEPS
constexpr int kCombineWaves = 16;
__global__ __launch_bounds__(kCombineWaves * kWave) void combine_kernel(
    const double *__restrict__ part_max, const int64_t *__restrict__ part_idx,
    const double *__restrict__ part_sum, int n_sets, int n, int64_t set_stride, int mode,
    int64_t node_offset, double n_nodes_total, double *__restrict__ out_max,
    double *__restrict__ out_norm_or_sum, int64_t *__restrict__ out_idx,
    const int32_t *__restrict__ run_if, double *__restrict__ out_z) {
    // (out_z, optional: the largest log2-domain maximum itself -- what tie_rule = 1 measures its slack from)
    // 16 wavefronts split the sets (wave w: sets w, w + 16, ...), four sets' loads in flight per
    // wave: a scan of a few hundred samples has only a handful of 64-sample columns, so the sets
    // are where the parallelism is (Icequake-sized step, 625 samples x 576 sets: 48 -> ~9 us,
    // i.e. the whole step 0.51 -> 0.45 ms).
    __shared__ double smax[kCombineWaves][kWave], ssum[kCombineWaves][kWave];
    __shared__ int64_t sidx[kCombineWaves][kWave];
    if (run_if != nullptr && *run_if == 0) return;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int t = blockIdx.x * kWave + lane;
    const int tc = t < n ? t : n - 1;
    double best = -__builtin_inf(), total = 0.0;
    int64_t bi = kNoIndex;
    constexpr int U = 4;
    for (int s0 = wave; s0 < n_sets; s0 += kCombineWaves * U) {
        double v[U], p[U];
        int64_t i[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int s = s0 + kCombineWaves * k;
            const int64_t o = (int64_t)(s < n_sets ? s : s0) * set_stride + tc;
            v[k] = part_max[o];
            i[k] = part_idx[o];
            p[k] = part_sum[o];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {                  // ascending set order per wave
            if (s0 + kCombineWaves * k < n_sets) {
                total += p[k];
                if (better(v[k], i[k], best, bi)) {
                    best = v[k];
                    bi = i[k];
                }
            }
        }
    }
    smax[wave][lane] = best;
    ssum[wave][lane] = total;
    sidx[wave][lane] = bi;
    __syncthreads();
    if (wave != 0 || t >= n) return;
    for (int w = 1; w < kCombineWaves; ++w) {
        total += ssum[w][lane];
        if (better(smax[w][lane], sidx[w][lane], best, bi)) {
            best = smax[w][lane];
            bi = sidx[w][lane];
        }
    }
    if (bi != kNoIndex) bi += node_offset;
    if (out_z) out_z[t] = best;
    if (mode == 0) {
        out_max[t] = best;
        out_norm_or_sum[t] = total;
        out_idx[t] = bi;
    } else {
        // (no finite sum at this sample -- only possible with -inf onsets, i.e. log(0), which
        // core/lib.py:93 clips away: the reference's exp(-inf) = 0, not rint's NaN)
        const double peak = (mode != 1) ? best : best == -__builtin_inf() ? 0.0 : qm_exp2_peak(best);
        out_max[t] = peak;
        out_norm_or_sum[t] = peak * n_nodes_total / total;     // migratelib.c:108
        out_idx[t] = (bi == kNoIndex) ? 0 : bi;
    }
}
#endif  // QM_TU_STEPS

}  // namespace qm
