// qm_screen.hpp -- the opt-in screened detect: an EXACT-INTEGER (fixed-point) sweep over every
// node-sample, then an exact float64 re-evaluation of the few (brick, sample) cells that can hold
// the maximum.
//
// Why: the fused float64 kernel (qm_kernels.hpp) is bound by LDS operand bandwidth (8 bytes per
// add).  With 4-byte operands a lane fetches a PAIR of consecutive samples with one ds_read_b64:
// half the LDS bytes per node-sample.  The detect outputs do not need every node-sample in
// float64 -- but what replaces it must come with a bound, not with a statistical argument.
// Float32 stacks cannot give one (worst case S * 2^-24 * sum|L| / available ~ 4e-6 > the 1e-6
// contract); integer stacks can, because integer adds are exact:
//
//   Per step, with c = log2(e) / available and R = max_r max_t |L_r(t)|, every log-onset is
//   quantised ONCE to q_r(t) = rint(L_r(t) * c * 2^k), k the largest integer with
//   S * (R c 2^k + 1) < 2^31 (no stack can overflow int32).  |q - L c 2^k| <= 1/2, so the integer
//   stack Q(n,t) = sum_r q_r(...) -- exact in any order -- satisfies
//       | Q 2^-k  -  z(n,t) | <= S 2^-(k+1) =: dz,     z = stack64 * c  (log2 of the coalescence).
//   (C3: S = 30, R c S ~ 3.2  ->  k = 29, dz = 2.8e-8.)
//
//   * max_coa, max_coa_idx (must be exact): the node(s) holding the float64 maximum have
//     Q >= Qmax - S (two stacks, each within S/2 units of its own truth).  The sweep keeps the
//     integer maximum of every (brick, sample) cell; every cell with max >= Qmax - S - 2 (two
//     units for the float64 roundings on the reference's side) is re-evaluated node by node in
//     float64, in the reference's operation order, and the maximum / lowest index is taken over
//     those cells.  Typically that is one cell per sample.
//   * max_norm_coa = max * N / sum_n 2^z: the sweep's term for node n is exp2f(float(Q) * 2^-k),
//     summed in float32 with compensation over all the nodes a wave visits, then in float64.
//     Relative error, worst case:
//         ln2 * dz                      quantisation           (required <= 1.0e-7 per step)
//       + ln2 * |z| * 2^-24             int32 -> float32       (3.3e-7 at |z| <= 8, required)
//       + 2^-23                         v_exp_f32, <= 1 ulp    (1.2e-7; checked exhaustively on
//                                                               the device, qm_exp2f_max_error)
//       + 2 * 2^-24                     compensated float32 sum (1.2e-7, any number of terms)
//       (the scaling by 2^-k is exact; the float64 adds contribute ~1e-15)
//     <= 6.7e-7.  Every term is positive, so the sum -- and with it max_norm_coa, whose numerator
//     is the exact float64 maximum -- inherits at most that relative error: inside the 1e-6
//     contract BY CONSTRUCTION, whatever the data (correlated errors included).
//   The per-step preconditions (finite onsets, ln2 * dz <= 1e-7, sum_r max_t |L_r| c <= 8, at most
//   16 candidate cells per sample) are evaluated ON THE DEVICE; a step that fails one is redone by
//   the float64 kernel, which is enqueued behind every screened step and returns at once
//   otherwise: a device-side flag decides, the host never waits.
//
// LDS layout (32-bit words): row r of a brick owns two staggered copies of its window,
//   A_r[u] = q[first_r + u], u < span2_r + KT;   B_r[u] = q[first_r + u + 1], u < span2_r + KT - 2
// (span2 = delay span rounded up to even) so that a pair starting at ANY delay d is an 8-byte
// aligned ds_read_b64: even d reads A at word d, odd d reads B at word d-1 (a 4-byte-aligned
// ds_read_b64 is ~28x slower on gfx950).  The 16-bit table holds, per node and row, the byte offset
//   8*P_r + 4*(d & ~1) + (d & 1) * 4 * (span2_r + KT)   (P_r = sum of span2 over the rows before r)
// and the row's r*(8*KT - 8) goes into the read's immediate offset.  Rows are added two at a
// time: q_a + q_b + acc is one v_add3_u32 per sample of the pair.
#pragma once

#include "qm_kernels.hpp"

namespace qm {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int qm_v2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) qm_v2i lds_v2i;
constexpr int kScreenSlots = 16;        // candidate cells kept per sample
constexpr double kScreenMaxZ = 8.0;     // bound on |z| the int32 -> float32 budget is stated for
constexpr double kScreenMaxQuant = 1.0e-7;   // budget of ln2 * dz

// per-step parameters of the sweep, computed on the device (screen_quantise_kernel)
struct ScreenParams {
    float unit;                    // 2^-k: float(Q) * unit = z
    int32_t k;                     // fractional bits of the fixed-point log2-coalescence
    int32_t slack;                 // S + 2: a cell within `slack` units of the maximum is a candidate
    int32_t pad;
};

struct ScreenArgs {
    GridDesc g;
    const int32_t *onsets_q;       // [S][T] quantised log-onsets (units of 2^-k in z)
    const uint16_t *rel;           // [nbricks][brick_nodes][row_pad] byte offsets (see above)
    const int32_t *brick_meta;     // [nbricks][S] int4 (min delay, span2, P_r, 0)
    const int32_t *brick_total;    // [nbricks] P_S
    int T, fsmp, n_samples;
    int ntiles, ngroups;
    int window_bytes;              // LDS bytes available to the windows
    const ScreenParams *params;    // device: this step's 2^-k
    int32_t *cell_max;             // [nbricks][ns_pad] integer stack maxima
    int32_t *group_max;            // [ngroups][ns_pad] maxima over the cells of a workgroup
    int64_t ns_pad;                // ntiles * KT
    double *part_sum;              // [ngroups][n_samples]
};

// a brick can be screened iff its windows fit the LDS budget and its offsets fit 16 bits
__host__ __device__ __forceinline__ bool screen_fits(int64_t p_total, int n_rows, int kt,
                                                     int window_bytes) {
    return 8 * p_total + (int64_t)n_rows * (8 * kt - 8) <= window_bytes &&
           12 * p_total + 4 * kt <= kMaxSpanBytes;
}

#ifdef QM_TU_SCREEN
// ---- per step: max |L| per row, then the quantised copy of the log-onsets -----------------------
__global__ __launch_bounds__(256) void screen_rowmax_kernel(const double *__restrict__ onsets,
                                                            int T, double *__restrict__ row_absmax) {
    __shared__ double red[256];
    const int r = blockIdx.x;
    double m = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const double av = __builtin_fabs(onsets[(int64_t)r * T + t]);
        m = (av > m || av != av) ? av : m;            // a NaN sticks: the step is then not screened
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double o = red[threadIdx.x + s];
            if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) row_absmax[r] = red[0];
}

// flags[0]: bit 0 = redo the step in float64 (a precondition of the bound fails, or some sample
// has more candidate cells than slots), bit 1 = non-finite onsets; flags[1]: work-list entries.
__global__ __launch_bounds__(256) void screen_quantise_kernel(
    const double *__restrict__ onsets, int T, int n_rows, int available,
    const double *__restrict__ row_absmax, int32_t *__restrict__ out,
    ScreenParams *__restrict__ params, int32_t *__restrict__ flags) {
    const int r = blockIdx.x;
    const double c = QM_LOG2E / (double)available;
    double rmax = 0.0, rsum = 0.0;
    bool finite = true;
    for (int i = 0; i < n_rows; ++i) {
        const double v = row_absmax[i];
        finite = finite && (v < 1e300);                 // false for NaN and Inf
        rmax = v > rmax ? v : rmax;
        rsum += v;
    }
    // largest k <= 30 with n_rows * (rmax c 2^k + 1) < 2^31
    int k = 0;
    if (finite) {
        k = 30;
        while (k > 0 && (double)n_rows * (rmax * c * __builtin_amdgcn_ldexp(1.0, k) + 1.0) >=
                            2147483647.0)
            --k;
    }
    const double scale = c * __builtin_amdgcn_ldexp(1.0, k);
    for (int t = threadIdx.x; t < T; t += 256) {
        const double v = onsets[(int64_t)r * T + t];
        out[(int64_t)r * T + t] = finite ? (int32_t)__builtin_rint(v * scale) : 0;
    }
    if (r == 0 && threadIdx.x == 0) {
        params->unit = (float)__builtin_amdgcn_ldexp(1.0, -k);
        params->k = k;
        params->slack = n_rows + 2;
        params->pad = 0;
        const double dz = (double)n_rows * __builtin_amdgcn_ldexp(1.0, -(k + 1));
        int bad = 0;
        if (!finite) bad |= 2;
        if (!(0.6931471805599453 * dz <= kScreenMaxQuant)) bad |= 1;   // quantisation budget
        if (!(rsum * c <= kScreenMaxZ)) bad |= 1;                        // |z| <= 8 for every node-sample
        if (bad) atomicOr(flags, bad);
    }
}
#endif  // QM_TU_SCREEN

#ifdef QM_TU_TABLES
// ---- per table: span2 prefixes and the staggered-copy offset table -----------------------------
__global__ void screen_prefix_kernel(GridDesc g, const int4 *__restrict__ meta,
                                     int4 *__restrict__ smeta, int32_t *__restrict__ stotal) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.nbricks) return;
    int64_t run = 0;
    for (int r = 0; r < g.n_rows; ++r) {
        const int4 m = meta[(int64_t)b * g.n_rows + r];
        const int span2 = (m.y + 1) & ~1;
        smeta[(int64_t)b * g.n_rows + r] =
            make_int4(m.x, span2, (int32_t)(run > INT32_MAX ? INT32_MAX : run), 0);
        run += span2;
    }
    stotal[b] = (int32_t)(run > INT32_MAX ? INT32_MAX : run);
}

__global__ void screen_rel_kernel(GridDesc g, const int32_t *__restrict__ lut,
                                  const int4 *__restrict__ smeta,
                                  const int32_t *__restrict__ stotal, int kt, int window_bytes,
                                  uint16_t *__restrict__ rel) {
    const int b = blockIdx.x;
    const bool fits = screen_fits(stotal[b], g.n_rows, kt, window_bytes);
    const int per = g.brick_nodes * g.row_pad;
    int x0, y0, z0, vx, vy, vz;
    brick_extents(g, b, x0, y0, z0, vx, vy, vz);
    const int nvalid = vx * vy * vz;
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const int m = i / g.row_pad, r = i % g.row_pad;
        uint16_t v = 0;
        if (fits && r < g.n_rows && m < nvalid) {
            const int node = brick_walk_node(g, x0, y0, z0, vy, vz, m);
            int d = lut[(int64_t)node * g.n_rows + r];
            d = d < 0 ? 0 : d;
            const int4 rec = smeta[(int64_t)b * g.n_rows + r];
            d -= rec.x;
            v = (uint16_t)(8 * rec.z + 4 * (d & ~1) + (d & 1) * 4 * (rec.y + kt));
        }
        rel[(int64_t)b * per + i] = v;
    }
}
#endif  // QM_TU_TABLES

// ---- the integer sweep ------------------------------------------------------------------------
template <int JP>
__device__ __forceinline__ void stage_windows32(const ScreenArgs &a, int32_t *win, int b, int wave,
                                                int nwaves, int lane, int t_first) {
    constexpr int KT = 128 * JP;
    constexpr int U = 2 * JP + 1;                       // loads in flight per pass
    const int S = a.g.n_rows;
    for (int r0 = 0; r0 < S; r0 += kWave) {
        int4 rec = make_int4(0, 0, 0, 0);
        if (r0 + lane < S)
            rec = reinterpret_cast<const int4 *>(a.brick_meta)[(int64_t)b * S + r0 + lane];
        const int rend = (S - r0 < kWave) ? S - r0 : kWave;
        for (int k = wave; k < rend; k += nwaves) {
            const int r = r0 + k;
            const int lo = __builtin_amdgcn_readlane(rec.x, k);
            const int len = __builtin_amdgcn_readlane(rec.y, k) + KT;          // copy A
            const int dstA = 2 * __builtin_amdgcn_readlane(rec.z, k) + r * (2 * KT - 2);
            const int dstB = dstA + len;                                       // copy B: len - 2
            const int first = lo + a.fsmp + t_first;
            const int room = a.T - first;
            const int32_t *src = a.onsets_q + (int64_t)r * a.T + first;
            for (int u0 = 0; u0 < len; u0 += kWave * U) {
                int32_t v[U];
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int u = u0 + kWave * i + lane;
                    v[i] = (u < len && u < room) ? src[u] : 0;
                }
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int u = u0 + kWave * i + lane;
                    if (u < len) {
                        win[dstA + u] = v[i];
                        if (u > 0 && u < len - 1) win[dstB + u - 1] = v[i];
                    }
                }
            }
        }
    }
}

// One 8-row chunk (ROWS of them used): rows are taken two at a time -- acc + q_a + q_b is one
// v_add3_u32 per sample -- in batches of G pair-slots (G = 2: four ds_read_b64 per batch, two
// batches in flight: the register budget of the 16-wave workgroup).  Batch index B = P * NG + h:
// row pair P, slot group h.
template <int JP> struct SweepGroup { static constexpr int value = JP >= 2 ? 2 : 1; };

template <int JP, int ROWS, int B>
__device__ __forceinline__ void sweep_issue(qm_v2i (&buf)[2 * SweepGroup<JP>::value],
                                            const unsigned (&addr)[8]) {
    constexpr int KT = 128 * JP;
    constexpr unsigned ROWB = 8 * KT - 8;               // bytes between consecutive rows
    constexpr int G = SweepGroup<JP>::value, NG = JP / G;
    constexpr int P = B / NG, h = B % NG;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int r = 2 * P + half;                     // compile-time after unrolling
        if (r < ROWS) {
            // the row's r * ROWB and the pair's 512 * j go into the read's immediate offset
            const volatile lds_v2i *p = (const volatile lds_v2i *)(uintptr_t)addr[r];
#pragma unroll
            for (int j = 0; j < G; ++j) buf[half * G + j] = p[r * (ROWB / 8) + 64 * (h * G + j)];
        }
    }
}

template <int JP, int ROWS, int B>
__device__ __forceinline__ void sweep_retire(qm_v2i (&acc)[JP],
                                             const qm_v2i (&buf)[2 * SweepGroup<JP>::value]) {
    constexpr int G = SweepGroup<JP>::value, NG = JP / G;
    constexpr int P = B / NG, h = B % NG;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        if constexpr (2 * P + 1 < ROWS) acc[h * G + j] += buf[j] + buf[G + j];
        else acc[h * G + j] += buf[j];
    }
}

template <int JP, int ROWS, int B>
__device__ __forceinline__ void sweep_batches(qm_v2i (&acc)[JP],
                                              qm_v2i (&even)[2 * SweepGroup<JP>::value],
                                              qm_v2i (&odd)[2 * SweepGroup<JP>::value],
                                              const unsigned (&addr)[8]) {
    constexpr int NB = ((ROWS + 1) / 2) * (JP / SweepGroup<JP>::value);
    if constexpr (B < NB) {
        if constexpr (B + 1 < NB) sweep_issue<JP, ROWS, B + 1>((B & 1) ? even : odd, addr);
        __builtin_amdgcn_sched_barrier(0);
        sweep_retire<JP, ROWS, B>(acc, (B & 1) ? odd : even);
        __builtin_amdgcn_sched_barrier(0);
        sweep_batches<JP, ROWS, B + 1>(acc, even, odd, addr);
    }
}

template <int JP, int ROWS>
__device__ __forceinline__ void sweep_chunk(qm_v2i (&acc)[JP], const unsigned (&addr)[8]) {
    qm_v2i even[2 * SweepGroup<JP>::value], odd[2 * SweepGroup<JP>::value];
    sweep_issue<JP, ROWS, 0>(even, addr);
    sweep_batches<JP, ROWS, 0>(acc, even, odd, addr);
}

// The nodes of one brick for one wave, LAST = rows in the last offset chunk (compile time: a
// switch on the row count INSIDE the node loop is control flow, and LLVM then sinks the adds of
// all rows below it -- the operands would live in scratch; the kernel switches once per brick).
template <int JP, int NCH, int LAST>
__device__ __forceinline__ void sweep_brick(const ScreenArgs &a, const uint16_t *brick_rel,
                                            int nvalid, int wave, int nwaves, unsigned lane_addr,
                                            float unit, qm_v2i (&best)[JP], v2f (&fsum)[JP],
                                            v2f (&comp)[JP]) {
    constexpr int KT = 128 * JP;
    constexpr unsigned ROWB = 8 * KT - 8;               // bytes between consecutive rows
    const GridDesc &g = a.g;
    // fsum / comp: float32 sum of this wave's terms over ALL its bricks, COMPENSATED (Kahan): the
    // computed sum is within 2 * 2^-24 (+ O(n 2^-48)) of the exact sum of the float32 terms
    // whatever their number -- no float64 registers in the node loop, nothing to flush.  (Plain
    // HIP -O3 does not reassociate floating point, so the compensation survives; every term is
    // positive.)
    uint4 qn[NCH];
    {
        const uint16_t *p = brick_rel + (int64_t)(wave < nvalid ? wave : 0) * g.row_pad;
#pragma unroll
        for (int c = 0; c < NCH; ++c) qn[c] = load_offsets(p, c * 8);
    }
    for (int m = wave; m < nvalid; m += nwaves) {
        // the node after this one (or a harmless reload of this one at the end): a chunk of
        // qn is refilled with its offsets as soon as this node's copy has been unpacked
        const uint16_t *next =
            brick_rel + (int64_t)(m + nwaves < nvalid ? m + nwaves : m) * g.row_pad;
        qm_v2i acc[JP];
#pragma unroll
        for (int j = 0; j < JP; ++j) acc[j] = qm_v2i{0, 0};
        unsigned addr[8];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unpack8(qn[c], lane_addr + (unsigned)c * 8u * ROWB, addr);
            qn[c] = load_offsets(next, c * 8);
            if (c + 1 < NCH) sweep_chunk<JP, 8>(acc, addr);
            else sweep_chunk<JP, LAST>(acc, addr);
        }
#pragma unroll
        for (int j = 0; j < JP; ++j) {
            best[j].x = acc[j].x > best[j].x ? acc[j].x : best[j].x;
            best[j].y = acc[j].y > best[j].y ? acc[j].y : best[j].y;
            // z = float(Q) * 2^-k: one rounding (the conversion); the scaling is exact
            const v2f z = v2f{(float)acc[j].x, (float)acc[j].y} * v2f{unit, unit};
            const v2f term = v2f{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
            const v2f y = term - comp[j];
            const v2f t = fsum[j] + y;
            comp[j] = (t - fsum[j]) - y;
            fsum[j] = t;
        }
    }
}

template <int JP, int NCH>
__global__ __launch_bounds__(1024) void screen_lds_kernel(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) int32_t swin[];
    constexpr int KT = 128 * JP;
    constexpr unsigned ROWB = 8 * KT - 8;               // bytes between consecutive rows
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int slot = blockIdx.x >> 3;                   // XCD-aware map, as stack_lds_kernel
    const int tile = slot % a.ntiles;
    const int group = (int)(blockIdx.x & 7) + 8 * (slot / a.ntiles);
    if (group >= a.ngroups) return;
    const int t_first = tile * KT;
    const int S = g.n_rows;
    const int last_rows = S - 8 * (NCH - 1);
    const float unit = a.params->unit;                  // 2^-k of this step
    const unsigned lane_addr =
        (unsigned)(uintptr_t)((__attribute__((address_space(3))) int32_t *)swin) + (unsigned)lane * 8u;
    // the cell-maximum row [KT] behind the windows: the waves merge into it at the end of a brick;
    // it is written out and reset between the next brick's two barriers (nobody merges there)
    int32_t *cellbuf = swin + a.window_bytes / 4;
    for (int k = threadIdx.x; k < KT; k += blockDim.x) cellbuf[k] = INT32_MIN;

    v2f fsum[JP], comp[JP];                             // this wave's compensated float32 sums
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        fsum[j] = v2f{0.f, 0.f};
        comp[j] = v2f{0.f, 0.f};
    }
    int prev_b = -1;
    int32_t group_best = INT32_MIN;                     // threads k < KT: maximum of sample k

    for (int b = group; b < g.nbricks; b += a.ngroups) {
        if (!screen_fits(a.brick_total[b], S, KT, a.window_bytes)) {
            // the float64 direct kernel covers this brick; its cells never become candidates
            for (int k = threadIdx.x; k < KT; k += blockDim.x)
                a.cell_max[(int64_t)b * a.ns_pad + t_first + k] = INT32_MIN;
            continue;
        }
        __syncthreads();                                // previous brick consumed and merged
        if (prev_b >= 0) {
            if ((int)threadIdx.x < KT) {
                const int32_t v = cellbuf[threadIdx.x];
                a.cell_max[(int64_t)prev_b * a.ns_pad + t_first + threadIdx.x] = v;
                group_best = v > group_best ? v : group_best;
                cellbuf[threadIdx.x] = INT32_MIN;
            }
        }
        stage_windows32<JP>(a, swin, b, wave, nwaves, lane, t_first);
        __syncthreads();

        int x0, y0, z0, vx, vy, vz;
        brick_extents(g, b, x0, y0, z0, vx, vy, vz);
        const int nvalid = vx * vy * vz;
        const uint16_t *brick_rel = a.rel + (int64_t)b * g.brick_nodes * g.row_pad;

        qm_v2i best[JP];
#pragma unroll
        for (int j = 0; j < JP; ++j) best[j] = qm_v2i{INT32_MIN, INT32_MIN};
        switch (last_rows) {                            // once per brick, outside the node loop
#define QM_SWEEP_CASE(L)                                                                       \
    case L:                                                                                    \
        sweep_brick<JP, NCH, L>(a, brick_rel, nvalid, wave, nwaves, lane_addr, unit, best, fsum,  \
                                comp);                                                         \
        break;
            QM_SWEEP_CASE(1) QM_SWEEP_CASE(2) QM_SWEEP_CASE(3) QM_SWEEP_CASE(4)
            QM_SWEEP_CASE(5) QM_SWEEP_CASE(6) QM_SWEEP_CASE(7) QM_SWEEP_CASE(8)
#undef QM_SWEEP_CASE
            default: break;
        }
        {   // merge this wave's cell maxima: LDS integer max, no return value
            unsigned cell = (unsigned)(uintptr_t)((__attribute__((address_space(3))) int32_t *)cellbuf) +
                            (unsigned)lane * 8u;
#pragma unroll
            for (int j = 0; j < JP; ++j) {
                asm volatile("ds_max_i32 %0, %1\n\tds_max_i32 %0, %2 offset:4"
                             :: "v"(cell), "v"(best[j].x), "v"(best[j].y) : "memory");
                cell += 512u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        prev_b = b;
    }
    __syncthreads();
    if ((int)threadIdx.x < KT) {
        if (prev_b >= 0) {
            const int32_t v = cellbuf[threadIdx.x];
            a.cell_max[(int64_t)prev_b * a.ns_pad + t_first + threadIdx.x] = v;
            group_best = v > group_best ? v : group_best;
        }
        a.group_max[(int64_t)group * a.ns_pad + t_first + threadIdx.x] = group_best;
    }
    // partial sums of this workgroup: cross-wave through LDS (the windows are dead)
    double *red = reinterpret_cast<double *>(swin);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        red[wave * KT + 128 * j + 2 * lane] = (double)fsum[j].x - (double)comp[j].x;
        red[wave * KT + 128 * j + 2 * lane + 1] = (double)fsum[j].y - (double)comp[j].y;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < KT; k += blockDim.x) {
        double total = 0.0;
        for (int w = 0; w < nwaves; ++w) total += red[w * KT + k];
        if (t_first + k < a.n_samples) a.part_sum[(int64_t)group * a.n_samples + t_first + k] = total;
    }
}

#ifdef QM_TU_SCREEN
// ---- candidates --------------------------------------------------------------------------------
// integer maximum of every sample over the workgroups: peak[t]
__global__ __launch_bounds__(256) void screen_peak_kernel(const int32_t *__restrict__ group_max,
                                                          int64_t ns_pad, int n_samples,
                                                          int ngroups, int32_t *__restrict__ peak) {
    __shared__ int32_t red[4][kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * kWave + lane;
    const int tc = t < n_samples ? t : n_samples - 1;
    int32_t m = INT32_MIN;
    for (int gsel = wave; gsel < ngroups; gsel += 4) {
        const int32_t v = group_max[(int64_t)gsel * ns_pad + tc];
        m = v > m ? v : m;
    }
    red[wave][lane] = m;
    __syncthreads();
    if (wave == 0 && t < n_samples) {
        for (int w = 1; w < 4; ++w) m = red[w][lane] > m ? red[w][lane] : m;
        peak[t] = m;
    }
}

// every cell within `slack` units of the sample's integer maximum becomes a candidate: a slot in
// the sample's list (for the final pick) and an entry in the flat work list (for the refinement).
// Only the cells of workgroups whose own maximum reaches the bar are looked at (workgroup g owns
// bricks g, g + ngroups, ...).
__global__ __launch_bounds__(256) void screen_candidates_kernel(
    const int32_t *__restrict__ cell_max, const int32_t *__restrict__ group_max, int64_t ns_pad,
    int n_samples, int nbricks, int ngroups, int groups_per_block,
    const int32_t *__restrict__ peak, const ScreenParams *__restrict__ params,
    int32_t *__restrict__ counts, int32_t *__restrict__ cells, int32_t *__restrict__ work,
    int32_t *__restrict__ flags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * kWave + lane;
    if (t >= n_samples) return;
    if (flags[0] & 2) return;                           // non-finite onsets: nothing was swept
    const int64_t bar = (int64_t)peak[t] - params->slack;
    const int g0 = blockIdx.y * groups_per_block;
    const int g1 = min(ngroups, g0 + groups_per_block);
    for (int gsel = g0 + wave; gsel < g1; gsel += 4) {
        if ((int64_t)group_max[(int64_t)gsel * ns_pad + t] < bar) continue;
        for (int b = gsel; b < nbricks; b += ngroups) {
            if ((int64_t)cell_max[(int64_t)b * ns_pad + t] < bar) continue;
            const int k = atomicAdd(&counts[t], 1);
            if (k < kScreenSlots) {
                cells[(int64_t)t * kScreenSlots + k] = b;
                work[atomicAdd(&flags[1], 1)] = t * kScreenSlots + k;
            } else {
                atomicOr(flags, 1);
            }
        }
    }
}
#endif  // QM_TU_SCREEN

// ---- exact re-evaluation of the candidate cells: one workgroup per work-list entry -------------
struct RefineArgs {
    GridDesc g;
    const double *onsets;          // [S][T] float64 log-onsets
    const int32_t *lut;            // [N][S]
    int T, fsmp, n_samples;
    double z_scale;
    const int32_t *cells;          // [n_samples][kScreenSlots] brick of each slot
    const int32_t *work;           // slot ids to evaluate
    const int32_t *flags;          // flags[1] = number of entries in `work`
    double *cand_z;                // [n_samples][kScreenSlots]
    int64_t *cand_idx;
};

#ifdef QM_TU_SCREEN
__global__ __launch_bounds__(256) void screen_refine_kernel(RefineArgs a) {
    __shared__ double sz[4];
    __shared__ int64_t si[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const GridDesc &g = a.g;
    const int total = a.flags[1];
    for (int c = blockIdx.x; c < total; c += gridDim.x) {
        const int slot = a.work[c];
        const int t = slot / kScreenSlots;
        const int b = a.cells[slot];
        int x0, y0, z0, vx, vy, vz;
        brick_extents(g, b, x0, y0, z0, vx, vy, vz);
        const int nvalid = vx * vy * vz;
        double best = -__builtin_inf();
        int64_t bi = kNoIndex;
        for (int m = threadIdx.x; m < nvalid; m += 256) {   // ascending node index per thread
            const int node = brick_walk_node(g, x0, y0, z0, vy, vz, m);
            const int32_t *row = a.lut + (int64_t)node * g.n_rows;
            const double *col = a.onsets + a.fsmp + t;
            double s = 0.0;
            int r = 0;
            for (; r + 8 <= g.n_rows; r += 8) {             // 8 delays, then 8 gathers, in flight
                int d[8];
                double v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = row[r + k];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    v[k] = col[(int64_t)(r + k) * a.T + (d[k] < 0 ? 0 : d[k])];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];      // ascending rows: migratelib.c:54-59
            }
            for (; r < g.n_rows; ++r) {
                const int d = row[r];
                s += col[(int64_t)r * a.T + (d < 0 ? 0 : d)];
            }
            const double z = s * a.z_scale;
            if (z > best) {                                 // strict: first node wins
                best = z;
                bi = node;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off, kWave);
            const int64_t oi = __shfl_xor(bi, off, kWave);
            if (better(ov, oi, best, bi)) {
                best = ov;
                bi = oi;
            }
        }
        __syncthreads();                                    // previous entry's sz/si consumed
        if (lane == 0) {
            sz[wave] = best;
            si[wave] = bi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (better(sz[w], si[w], best, bi)) {
                    best = sz[w];
                    bi = si[w];
                }
            a.cand_z[slot] = best;
            a.cand_idx[slot] = bi;
        }
    }
}
#endif  // QM_TU_SCREEN

#ifdef QM_TU_SCREEN
// one partial set (log2-domain maximum, local node index, sum) from the candidates and the
// workgroups' sums -- the same form stack_lds_kernel publishes, so combine_kernel finishes it.
__global__ __launch_bounds__(256) void screen_collect_kernel(
    const int32_t *__restrict__ counts, const double *__restrict__ cand_z,
    const int64_t *__restrict__ cand_idx, const double *__restrict__ part_sum, int ngroups,
    int n_samples, double *__restrict__ out_max, int64_t *__restrict__ out_idx,
    double *__restrict__ out_sum) {
    __shared__ double ssum[4][kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * kWave + lane;
    const int tc = t < n_samples ? t : n_samples - 1;
    double total = 0.0;
    for (int gsel = wave; gsel < ngroups; gsel += 4) total += part_sum[(int64_t)gsel * n_samples + tc];
    ssum[wave][lane] = total;
    __syncthreads();
    if (wave != 0 || t >= n_samples) return;
    total = ((ssum[0][lane] + ssum[1][lane]) + ssum[2][lane]) + ssum[3][lane];
    double best = -__builtin_inf();
    int64_t bi = kNoIndex;
    const int n = min(counts[t], kScreenSlots);
    for (int k = 0; k < n; ++k) {
        const double v = cand_z[(int64_t)t * kScreenSlots + k];
        const int64_t i = cand_idx[(int64_t)t * kScreenSlots + k];
        if (better(v, i, best, bi)) {
            best = v;
            bi = i;
        }
    }
    out_max[t] = best;
    out_idx[t] = bi;
    out_sum[t] = total;
}

// ---- exhaustive accuracy check of v_exp_f32 (the bound above quotes <= 1 ulp for it) ------------
// every float32 in [lo, hi]: max relative deviation of exp2f from the float64 exp2, per block
__global__ __launch_bounds__(256) void exp2f_error_kernel(float lo, float hi,
                                                          double *__restrict__ block_max) {
    __shared__ double red[256];
    const unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);   // same sign: monotone bits
    const unsigned first = a < b ? a : b, last = a < b ? b : a;
    double worst = 0.0;
    for (uint64_t u = (uint64_t)first + (uint64_t)blockIdx.x * 256 + threadIdx.x; u <= last;
         u += (uint64_t)gridDim.x * 256) {
        const float x = __uint_as_float((unsigned)u);
        const double want = exp2((double)x);
        const double got = (double)__builtin_amdgcn_exp2f(x);
        const double rel = __builtin_fabs(got - want) / want;
        worst = rel > worst ? rel : worst;
    }
    red[threadIdx.x] = worst;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_max[blockIdx.x] = red[0];
}
#endif  // QM_TU_SCREEN

}  // namespace qm
