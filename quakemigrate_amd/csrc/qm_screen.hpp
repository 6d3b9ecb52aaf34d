// qm_screen.hpp -- detect in two precisions: a float32 screening sweep over every node-sample,
// then an exact float64 re-evaluation of the few (brick, sample) cells that can hold the maximum.
//
// Why: the fused float64 kernel (qm_kernels.hpp) is bound by LDS operand bandwidth (8 bytes per
// add) and FP64 VALU issue together.  In float32 a lane fetches a PAIR of consecutive samples
// with one ds_read_b64 and adds both with one v_pk_add_f32: half the LDS bytes and half the VALU
// issue per node-sample.  The detect outputs do not need every node-sample in float64:
//   * the sum over nodes behind max_norm_coa is an average of ~N values: float32 terms
//     (relative error ~1e-7 each, independent) leave it accurate to ~1e-8, contract 1e-6 --
//     unless the terms' errors are all alike (a numerically flat sample) or one term carries the
//     sum; both are detected per sample (screen_collect_kernel) and such a step is redone;
//   * the maximum and its node index must be exact.  With A = sum_r max_t |L_r(t)| the float32
//     stack of any node differs from its float64 stack by at most D = 1.001 * S * 2^-24 * A
//     (one rounding per stored operand, one per add, any order).  So the node(s)
//     holding the true maximum have a float32 stack >= (float32 maximum) - 2 D.  The sweep keeps
//     the float32 maximum of every (brick, sample) cell; every cell within 2 D of the sample's
//     maximum is re-evaluated node by node in float64, in the reference's operation order, and the
//     exact maximum / lowest node index is taken over those cells.  Typically that is one cell
//     (512 nodes) per sample.  If a sample has more candidate cells than slots, or the onsets
//     are not finite, the step is redone too.  "Redone" = the float64 kernel, enqueued behind
//     every screened step, runs instead of returning at once: a device-side flag decides, the
//     host never waits.
// max_coa and max_coa_idx are therefore identical to the float64 path's; max_norm_coa agrees to
// ~1e-8 relative.
//
// LDS layout (float words): row r of a brick owns two staggered copies of its window,
//   A_r[u] = L[first_r + u], u < span2_r + KT;   B_r[u] = L[first_r + u + 1], u < span2_r + KT - 2
// (span2 = delay span rounded up to even) so that a pair starting at ANY delay d is an 8-byte
// aligned ds_read_b64: even d reads A at word d, odd d reads B at word d-1 (a 4-byte-aligned
// ds_read_b64 is ~28x slower on gfx950).  The 16-bit table holds, per node and row, the byte offset
//   8*P_r + 4*(d & ~1) + (d & 1) * 4 * (span2_r + KT)   (P_r = sum of span2 over the rows before r)
// and the row's r*(8*KT - 8) goes into the read's immediate offset.
#pragma once

#include "qm_kernels.hpp"

namespace qm {

typedef qm_v2f v2f;
constexpr int kScreenSlots = 16;        // candidate cells kept per sample

struct ScreenArgs {
    GridDesc g;
    const float *onsets32;         // [S][T] log-onsets rounded to float32
    const uint16_t *rel;           // [nbricks][brick_nodes][row_pad] byte offsets (see above)
    const int32_t *brick_meta;     // [nbricks][S] int4 (min delay, span2, P_r, 0)
    const int32_t *brick_total;    // [nbricks] P_S
    int T, fsmp, n_samples;
    int ntiles, ngroups;
    int window_bytes;              // LDS bytes available to the windows
    float z_scale;                 // log2(e) / available
    float *cell_max;               // [nbricks][ns_pad] float32 stack maxima (not scaled)
    float *group_max;              // [ngroups][ns_pad] maxima over the cells of a workgroup
    int64_t ns_pad;                // ntiles * KT
    double *part_sum;              // [ngroups][n_samples]
};

// a brick can be screened iff its windows fit the LDS budget and its offsets fit 16 bits
__host__ __device__ __forceinline__ bool screen_fits(int64_t p_total, int n_rows, int kt,
                                                     int window_bytes) {
    return 8 * p_total + (int64_t)n_rows * (8 * kt - 8) <= window_bytes &&
           12 * p_total + 4 * kt <= kMaxSpanBytes;
}

// ---- per step: float32 copy of the log-onsets and max |L| per row ------------------------------
__global__ __launch_bounds__(256) void screen_prepare_kernel(const double *__restrict__ onsets,
                                                             int T, float *__restrict__ out,
                                                             double *__restrict__ row_absmax) {
    __shared__ double red[256];
    const int r = blockIdx.x;
    double m = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const double v = onsets[(int64_t)r * T + t];
        out[(int64_t)r * T + t] = (float)v;
        const double av = __builtin_fabs(v);
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

// ---- the float32 sweep ------------------------------------------------------------------------
__device__ __forceinline__ float max_keep32(float best, float x) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(best), "v"(x));
    return r;
}

template <int JP>
__device__ __forceinline__ void stage_windows32(const ScreenArgs &a, float *win, int b, int wave,
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
            const float *src = a.onsets32 + (int64_t)r * a.T + first;
            for (int u0 = 0; u0 < len; u0 += kWave * U) {
                float v[U];
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int u = u0 + kWave * i + lane;
                    v[i] = (u < len && u < room) ? src[u] : 0.0f;
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

template <int JP, int NCH>
__global__ __launch_bounds__(1024) void screen_lds_kernel(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) float swin[];
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
    const unsigned lane_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)swin) +
                               (unsigned)lane * 8u;
    // the cell-maximum row [KT] behind the windows: the waves merge into it at the end of a brick;
    // it is written out and reset between the next brick's two barriers (nobody merges there)
    float *cellbuf = swin + a.window_bytes / 4;
    for (int k = threadIdx.x; k < KT; k += blockDim.x) cellbuf[k] = -__builtin_inff();

    double vsum[2 * JP];
#pragma unroll
    for (int i = 0; i < 2 * JP; ++i) vsum[i] = 0.0;
    int prev_b = -1;
    float group_best = -__builtin_inff();              // threads k < KT: maximum of sample k

    for (int b = group; b < g.nbricks; b += a.ngroups) {
        if (!screen_fits(a.brick_total[b], S, KT, a.window_bytes)) {
            // the float64 direct kernel covers this brick; its cells never become candidates
            for (int k = threadIdx.x; k < KT; k += blockDim.x)
                a.cell_max[(int64_t)b * a.ns_pad + t_first + k] = -__builtin_inff();
            continue;
        }
        __syncthreads();                                // previous brick consumed and merged
        if (prev_b >= 0) {
            if ((int)threadIdx.x < KT) {
                const float v = cellbuf[threadIdx.x];
                a.cell_max[(int64_t)prev_b * a.ns_pad + t_first + threadIdx.x] = v;
                group_best = fmaxf(group_best, v);
                cellbuf[threadIdx.x] = -__builtin_inff();
            }
        }
        stage_windows32<JP>(a, swin, b, wave, nwaves, lane, t_first);
        __syncthreads();

        int x0, y0, z0, vx, vy, vz;
        brick_extents(g, b, x0, y0, z0, vx, vy, vz);
        const int nvalid = vx * vy * vz;
        const uint16_t *brick_rel = a.rel + (int64_t)b * g.brick_nodes * g.row_pad;

        v2f best[JP], fsum[JP];
#pragma unroll
        for (int j = 0; j < JP; ++j) {
            best[j] = v2f{-__builtin_inff(), -__builtin_inff()};
            fsum[j] = v2f{0.f, 0.f};
        }
        uint4 qn[NCH];
        {
            const uint16_t *p = brick_rel + (int64_t)(wave < nvalid ? wave : 0) * g.row_pad;
#pragma unroll
            for (int c = 0; c < NCH; ++c) qn[c] = load_offsets(p, c * 8);
        }
        int count = 0;
        for (int m = wave; m < nvalid; m += nwaves) {
            uint4 qc[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) qc[c] = qn[c];
            {
                const uint16_t *p =
                    brick_rel + (int64_t)(m + nwaves < nvalid ? m + nwaves : m) * g.row_pad;
#pragma unroll
                for (int c = 0; c < NCH; ++c) qn[c] = load_offsets(p, c * 8);
            }
            v2f acc[JP];
#pragma unroll
            for (int j = 0; j < JP; ++j) acc[j] = v2f{0.f, 0.f};
            unsigned addr[8];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                unpack8(qc[c], lane_addr + (unsigned)c * 8u * ROWB, addr);
                if (c + 1 < NCH || last_rows == 8) ring32_full<JP>(acc, addr);
                else ring32_tail<JP>(acc, addr, last_rows);
            }
#pragma unroll
            for (int j = 0; j < JP; ++j) {
                best[j].x = max_keep32(best[j].x, acc[j].x);
                best[j].y = max_keep32(best[j].y, acc[j].y);
                const v2f z = acc[j] * v2f{a.z_scale, a.z_scale};
                fsum[j] += v2f{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
            }
            if ((++count & 7) == 0) {                   // keep float32 partial sums short
#pragma unroll
                for (int j = 0; j < JP; ++j) {
                    vsum[2 * j] += (double)fsum[j].x;
                    vsum[2 * j + 1] += (double)fsum[j].y;
                    fsum[j] = v2f{0.f, 0.f};
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JP; ++j) {
            vsum[2 * j] += (double)fsum[j].x;
            vsum[2 * j + 1] += (double)fsum[j].y;
        }
        {   // merge this wave's cell maxima: LDS float max, no return value
            unsigned cell = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)cellbuf) +
                            (unsigned)lane * 8u;
#pragma unroll
            for (int j = 0; j < JP; ++j) {
                asm volatile("ds_max_f32 %0, %1\n\tds_max_f32 %0, %2 offset:4"
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
            const float v = cellbuf[threadIdx.x];
            a.cell_max[(int64_t)prev_b * a.ns_pad + t_first + threadIdx.x] = v;
            group_best = fmaxf(group_best, v);
        }
        a.group_max[(int64_t)group * a.ns_pad + t_first + threadIdx.x] = group_best;
    }
    // partial sums of this workgroup: cross-wave through LDS (the windows are dead)
    double *red = reinterpret_cast<double *>(swin);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        red[wave * KT + 128 * j + 2 * lane] = vsum[2 * j];
        red[wave * KT + 128 * j + 2 * lane + 1] = vsum[2 * j + 1];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < KT; k += blockDim.x) {
        double total = 0.0;
        for (int w = 0; w < nwaves; ++w) total += red[w * KT + k];
        if (t_first + k < a.n_samples) a.part_sum[(int64_t)group * a.n_samples + t_first + k] = total;
    }
}

// ---- candidates --------------------------------------------------------------------------------
// float32 maximum of every sample over the workgroups: peak[t]
__global__ __launch_bounds__(256) void screen_peak_kernel(const float *__restrict__ group_max,
                                                          int64_t ns_pad, int n_samples,
                                                          int ngroups, float *__restrict__ peak) {
    __shared__ float red[4][kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * kWave + lane;
    const int tc = t < n_samples ? t : n_samples - 1;
    float m = -__builtin_inff();
    for (int gsel = wave; gsel < ngroups; gsel += 4) m = fmaxf(m, group_max[(int64_t)gsel * ns_pad + tc]);
    red[wave][lane] = m;
    __syncthreads();
    if (wave == 0 && t < n_samples)
        peak[t] = fmaxf(fmaxf(red[0][lane], red[1][lane]), fmaxf(red[2][lane], red[3][lane]));
}

// every cell within 2 D of the sample's float32 maximum becomes a candidate: a slot in the
// sample's list (for the final pick) and an entry in the flat work list (for the refinement).
// Only the cells of workgroups whose own maximum reaches the bar are looked at (workgroup g owns
// bricks g, g + ngroups, ...).
// flags[0]: bit 0 = some sample overflowed its slots, bit 1 = non-finite onsets; flags[1]: entries.
__global__ __launch_bounds__(256) void screen_candidates_kernel(
    const float *__restrict__ cell_max, const float *__restrict__ group_max, int64_t ns_pad,
    int n_samples, int nbricks, int ngroups, int groups_per_block, const float *__restrict__ peak,
    const double *__restrict__ row_absmax, int n_rows, int32_t *__restrict__ counts,
    int32_t *__restrict__ cells, int32_t *__restrict__ work, int32_t *__restrict__ flags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * kWave + lane;
    if (t >= n_samples) return;
    double A = 0.0;
    for (int r = 0; r < n_rows; ++r) A += row_absmax[r];
    const double D = 1.001 * (double)n_rows * 5.9604644775390625e-08 * A;     // S * 2^-24 * A
    if (!(D < 1e300)) {                                 // NaN / Inf in the onsets: do not screen
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) atomicOr(flags, 2);
        return;
    }
    const double bar = (double)peak[t] - 2.0 * D;
    const int g0 = blockIdx.y * groups_per_block;
    const int g1 = min(ngroups, g0 + groups_per_block);
    for (int gsel = g0 + wave; gsel < g1; gsel += 4) {
        if (!((double)group_max[(int64_t)gsel * ns_pad + t] >= bar)) continue;
        for (int b = gsel; b < nbricks; b += ngroups) {
            if (!((double)cell_max[(int64_t)b * ns_pad + t] >= bar)) continue;
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

// one partial set (log2-domain maximum, local node index, sum) from the candidates and the
// workgroups' sums -- the same form stack_lds_kernel publishes, so combine_kernel finishes it.
__global__ __launch_bounds__(256) void screen_collect_kernel(
    const int32_t *__restrict__ counts, const double *__restrict__ cand_z,
    const int64_t *__restrict__ cand_idx, const double *__restrict__ part_sum, int ngroups,
    int n_samples, int n_cells, const double *__restrict__ row_absmax, int n_rows, int available,
    int32_t *__restrict__ flags, double *__restrict__ out_max, int64_t *__restrict__ out_idx,
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
    // every cell of the grid a candidate: the sample is (numerically) flat, the float32 errors of
    // its terms are then all alike and do not average out of the sum -> redo the step in float64
    if (counts[t] >= n_cells) atomicOr(flags, 1);
    const int n = min(counts[t], kScreenSlots);
    for (int k = 0; k < n; ++k) {
        const double v = cand_z[(int64_t)t * kScreenSlots + k];
        const int64_t i = cand_idx[(int64_t)t * kScreenSlots + k];
        if (better(v, i, best, bi)) {
            best = v;
            bi = i;
        }
    }
    // Accuracy of the sum: a float32 term is off by about E relative -- the add roundings of its
    // stack (a random walk: ~sqrt(S)/sqrt(3) half-ulps of a partial sum <= A, taken twice over),
    // the rounding of its operands and of z (<= 2 u A / available), v_exp_f32 -- independently from
    // term to term, so the sum is off by about E * sqrt(sum of squared shares) <= E * sqrt(largest
    // share).  With ~N comparable terms that is E / sqrt(N); if one node dominates the sum
    // (extreme dynamic range on a small grid) it approaches E itself: then the step is redone in
    // float64 (budget 3e-7 of the contract's 1e-6).
    double A = 0.0;
    for (int r = 0; r < n_rows; ++r) A += row_absmax[r];
    const double E = (2.0 * __builtin_sqrt((double)n_rows / 3.0) + 2.0) * 5.9604644775390625e-08 * A /
                         (double)available + 2e-7;
    const double share = bi == kNoIndex ? 0.0 : qm_exp2_peak(best) / total;
    if (!(E * __builtin_sqrt(share) <= 3e-7)) atomicOr(flags, 1);
    out_max[t] = best;
    out_idx[t] = bi;
    out_sum[t] = total;
}

}  // namespace qm
