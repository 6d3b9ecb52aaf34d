// qm_pair.hpp -- the float64 stacking kernel with 16-byte LDS operands.
//
// Same arithmetic as stack_lds_kernel / stack_exact_kernel (qm_kernels.hpp): float64 sums in
// ascending row order per (node, sample), 2^z, running maximum / first index / sum, optional
// volume store.  What changes is how the operands leave LDS.  The kernel is bound by the LDS
// pipeline, and on gfx950 a wavefront gets its operands markedly cheaper as 16-byte reads
// (tools/micro/f64_lds.hip: ds_read_b128 beside its two dependent v_add_f64 sustains ~2.0-2.2 clk
// per 8-byte operand and CU, ds_read_b64 beside one ~2.55; the b128 form also halves the LDS
// instructions a wave has to issue).  So a lane owns PAIRS of consecutive samples,
//     t = t_first + 2*lane + 128*jp + {0, 1},   jp = 0 .. JP-1     (time tile KT = 128*JP)
// and fetches both with one ds_read_b128.  A 16-byte read must be 16-byte aligned (an 8-byte
// aligned ds_read_b128 is 10x slower, same micro-benchmark), but a node's delay d is any integer:
// every row window is therefore staged TWICE, staggered by one sample,
//     A_r[u] = L[first_r + u],   B_r[u] = L[first_r + u + 1],   u < span2_r + KT
// (span2 = delay span rounded up to even); an even d - min_r reads A at d - min_r, an odd one B at
// d - min_r - 1.  Layout in LDS doubles: row r starts at 2*(r*KT + P_r), A then B (P_r = sum of
// span2 over the rows before r).  The 16-bit table holds per (node, row) the byte offset
//     8 * (2*P_r + (e & ~1) + (e & 1) * (span2_r + KT)),   e = d - min_r
// and the row's 2*r*KT doubles go into a per-chunk base and the read's immediate offset, so the
// address is still one v_add_u32_sdwa per row.  Two copies of float64 windows need all 160 KB of
// a CU's LDS: one 16-wave workgroup per CU.  In the volume-writing variant a lane's pair is 16
// contiguous bytes of the node's volume row: one global_store_dwordx4 per pair.
//
// The per-brick running maximum is merged ACROSS the workgroup at the end of every brick (through
// the window area, which is dead by then): thread k then carries the running (maximum, index) of
// sample k of the tile, instead of every wave carrying a private copy for all its sample slots --
// 9 fewer VGPRs per lane, which is what lets the volume-writing variant keep its software
// pipeline in registers.
#pragma once

#include "qm_kernels.hpp"

namespace qm {

typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_v2d;

// a brick fits the paired layout iff both copies of its windows fit LDS and its byte offsets
// fit 16 bits (p_total = sum of span2 over the rows)
__host__ __device__ __forceinline__ bool pair_fits(int64_t p_total, int n_rows, int kt,
                                                   int lds_bytes) {
    return 16 * ((int64_t)n_rows * kt + p_total) <= lds_bytes &&
           16 * p_total + 8 * (int64_t)kt <= kMaxSpanBytes;
}

// offsets of the paired layout (one launch per (table, tile length)); smeta = (min, span2, P, 0)
// per (brick, row) as screen_prefix_kernel builds it
#ifdef QM_TU_TABLES
__global__ void pair_rel_kernel(GridDesc g, const int32_t *__restrict__ lut,
                                const int4 *__restrict__ smeta,
                                const int32_t *__restrict__ stotal, int kt, int lds_bytes,
                                uint16_t *__restrict__ rel) {
    const int b = blockIdx.x;
    const bool fits = pair_fits(stotal[b], g.n_rows, kt, lds_bytes);
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
            d = d < 0 ? 0 : d;                           // migratelib.c:55
            const int4 rec = smeta[(int64_t)b * g.n_rows + r];
            const int e = d - rec.x;
            v = (uint16_t)(8 * (2 * rec.z + (e & ~1) + (e & 1) * (rec.y + kt)));
        }
        rel[(int64_t)b * per + i] = v;
    }
}
#endif  // QM_TU_TABLES

// Stage both copies of every row window of brick b (a.brick_meta = smeta).  One global load per
// element, two LDS stores.
template <int JP>
__device__ __forceinline__ void stage_pair_windows(const StackArgs &a, double *win, int b,
                                                   int wave, int nwaves, int lane, int t_first) {
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
            const int len = __builtin_amdgcn_readlane(rec.y, k) + KT;
            const int dstA = 2 * (__builtin_amdgcn_readlane(rec.z, k) + r * KT);
            const int dstB = dstA + len;
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
                    if (u < len) {
                        win[dstA + u] = v[i];
                        if (u > 0) win[dstB + u - 1] = v[i];
                    }
                }
            }
        }
    }
}

// ---- pipeline pieces: one table row per batch, JP 16-byte reads per row --------------------
template <int JP, int S, int I>
__device__ __forceinline__ void pissue(v2d (&buf)[JP], uint4 (&q)[exact_nch(S)],
                                       const uint16_t *next, unsigned lane_addr) {
    constexpr int KT = 128 * JP;
    if constexpr (I < S) {
        constexpr int ci = I >> 3, e = I & 7;
        // row I: base 2*I*KT doubles = chunk base (2*8*KT doubles per chunk) + immediate
        const volatile lds_v2d *p = (const volatile lds_v2d *)(uintptr_t)(
            lane_addr + (unsigned)(ci * 8 * 2 * KT * 8) + chunk_entry(q[ci], e));
#pragma unroll
        for (int jp = 0; jp < JP; ++jp) buf[jp] = p[e * KT + 64 * jp];   // v2d units: 16 bytes
        // (the last chunk is refilled by the node loop, a whole node ahead of its use)
        if constexpr (e == 7 && ci + 1 < exact_nch(S)) q[ci] = load_offsets(next, ci * 8);
    }
}

template <int JP, int S, int I>
__device__ __forceinline__ void pretire(double (&acc)[2 * JP], const v2d (&buf)[JP]) {
    if constexpr (I < S) {
#pragma unroll
        for (int jp = 0; jp < JP; ++jp) {               // ascending row order per sample
            if constexpr (I == 0) {                     // 0.0 + x, without the add
                acc[2 * jp] = buf[jp].x;
                acc[2 * jp + 1] = buf[jp].y;
            } else {
                acc[2 * jp] += buf[jp].x;
                acc[2 * jp + 1] += buf[jp].y;
            }
        }
    }
}

// Stores of the lanes with `on` set, without a branch: EXEC is narrowed and restored inside one
// asm statement.  `row` is wave-uniform (scalar base), `u` the lane's offset in doubles.  The
// compiler's s_waitcnt bookkeeping does not see these stores; its waits can only come out
// stricter than needed for that (the counter is in-order), never too weak.  Its hazard
// recogniser does not see them either: gfx940+ needs 2 wait states between a VMEM store of more
// than 8 bytes and a VALU write to its data registers -- the s_nop supplies them.
__device__ __forceinline__ void store_pair_masked(double *row, int u, v2d val, bool on) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(on);
    const unsigned off = (unsigned)u * 8u;
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\t"
                 "global_store_dwordx4 %2, %3, %4 nt\n\ts_mov_b64 exec, %0\n\ts_nop 1"
                 : "=&s"(save)
                 : "s"(mask), "v"(off), "v"(val), "s"(row)
                 : "memory");
}

// epilogue of the previous node, sample slots j = 2*jp + h  <->  t = t_first + 2*lane + 128*jp + h.
// steps: 0 k | 1 f | 2..D+1 Horner | ldexp | sum | track | (store)
template <int JP, bool VOLUME, int TAIL, int STEP>
__device__ __forceinline__ void pepi_step(Epilogue<2 * JP> &s, double (&vsum)[2 * JP],
                                          double (&bmax)[2 * JP], int (&bidx)[2 * JP],
                                          const StackArgs &a, int t_first, int lane) {
    constexpr int J = 2 * JP;
    constexpr int D = Exp2Degree<VOLUME>::value;
    constexpr int H0 = 2, H1 = 2 + D;                  // Horner steps [H0, H1)
    if constexpr (VOLUME && STEP == H1 + 3) {
        // The stores come last: every offset load of the node being stacked has been issued by
        // now (loads and stores share one in-order counter on gfx9).  No control flow here: a
        // branch in the node body makes LLVM sink the adds of all rows below it, and the row
        // operands then live in scratch.  TAIL 0, a full tile: unconditional 16-byte stores
        // (s.row is wave-uniform: scalar base + 32-bit lane offset).  TAIL 1, the scan's last
        // tile, pulled back over its predecessor: the pairs that lie wholly in the overlap were
        // already written by that tile, their lanes are masked out of the store (EXEC narrowed
        // inside one asm statement: no branch, no duplicate HBM traffic).  TAIL 2, a scan shorter
        // than one tile: the lanes past the end are masked out; the lane that owns the last
        // sample of an odd-length scan stores 8 bytes.
        typedef v2d __attribute__((aligned(8))) v2d_a8;   // rows of an odd length start anywhere
#pragma unroll
        for (int jp = 0; jp < JP; ++jp) {
            const int u = 2 * lane + 128 * jp;
            v2d pair;
            pair.x = s.p[2 * jp];
            pair.y = s.p[2 * jp + 1];
            if constexpr (TAIL == 0) {
                __builtin_nontemporal_store(pair, reinterpret_cast<v2d_a8 *>(s.row + u));
            } else if constexpr (TAIL == 1) {
                const int overlap = 128 * JP - a.n_chunk % (128 * JP);   // tile samples [0, overlap)
                store_pair_masked(s.row, u, pair, u + 1 >= overlap);
            } else {
                store_pair_masked(s.row, u, pair, t_first + u + 1 < a.n_chunk);
                store_one_masked(s.row, u, pair.x, t_first + u + 1 == a.n_chunk);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if constexpr (STEP == 0) s.p[j] = __builtin_rint(s.x[j]);      // k (as double)
            else if constexpr (STEP == 1) {
                s.f[j] = s.x[j] - s.p[j];                                  // f = z - k
                s.k[j] = (int)s.p[j];
            } else if constexpr (STEP >= H0 && STEP < H1)
                s.p[j] = exp2_horner<D, STEP - H0>(s.p[j], s.f[j]);
            else if constexpr (STEP == H1) s.p[j] = __builtin_amdgcn_ldexp(s.p[j], s.k[j]);
            else if constexpr (STEP == H1 + 1) vsum[j] += s.p[j];
            else if constexpr (STEP == H1 + 2) {
                const bool gt = s.x[j] > bmax[j];                          // strict: first node wins
                bidx[j] = gt ? s.node : bidx[j];
                bmax[j] = max_keep(bmax[j], s.x[j]);
            }
        }
    }
}

template <int JP, bool VOLUME, int TAIL, int FIRST, int LAST>
__device__ __forceinline__ void pepi_steps(Epilogue<2 * JP> &s, double (&vsum)[2 * JP],
                                           double (&bmax)[2 * JP], int (&bidx)[2 * JP],
                                           const StackArgs &a, int t_first, int lane) {
    if constexpr (FIRST < LAST) {
        pepi_step<JP, VOLUME, TAIL, FIRST>(s, vsum, bmax, bidx, a, t_first, lane);
        pepi_steps<JP, VOLUME, TAIL, FIRST + 1, LAST>(s, vsum, bmax, bidx, a, t_first, lane);
    }
}

template <int JP, bool VOLUME, int TAIL, int S, bool WITH_EPI, int I>
__device__ __forceinline__ void pbatch(double (&acc)[2 * JP], v2d (&even)[JP], v2d (&odd)[JP],
                                       uint4 (&q)[exact_nch(S)], const uint16_t *next,
                                       unsigned lane_addr, Epilogue<2 * JP> &epi,
                                       double (&vsum)[2 * JP], double (&bmax)[2 * JP],
                                       int (&bidx)[2 * JP], const StackArgs &a, int t_first,
                                       int lane) {
    if constexpr (I < S) {
        if constexpr (I + 1 < S) pissue<JP, S, I + 1>((I & 1) ? even : odd, q, next, lane_addr);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WITH_EPI) {
            constexpr int E = XEpiSteps<VOLUME>::value;
            pepi_steps<JP, VOLUME, TAIL, I * E / S, (I + 1) * E / S>(epi, vsum, bmax, bidx, a,
                                                                       t_first, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        pretire<JP, S, I>(acc, (I & 1) ? odd : even);
        __builtin_amdgcn_sched_barrier(0);
        pbatch<JP, VOLUME, TAIL, S, WITH_EPI, I + 1>(acc, even, odd, q, next, lane_addr, epi,
                                                       vsum, bmax, bidx, a, t_first, lane);
    }
}

// sample index (inside the tile) of slot j of a lane
template <int JP>
__device__ __forceinline__ int pair_slot_sample(int lane, int j) {
    return 2 * lane + 128 * (j >> 1) + (j & 1);
}

template <int JP, bool VOLUME, int TAIL, int S>
__device__ __forceinline__ void stack_pair_body(const StackArgs &a, double *win) {
    constexpr int KT = 128 * JP;
    constexpr int J = 2 * JP;
    constexpr int NCH = exact_nch(S);
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    // XCD-aware workgroup -> (time tile, brick group) map, as stack_lds_kernel
    int tile, group;
    stack_tile_group(a, tile, group);
    // The last tile of a scan that is not a multiple of the tile length is pulled back so that it
    // ends with the scan: it then overlaps its predecessor, both compute the overlap with the same
    // arithmetic and write the same bits to the partial sets -- no lane is ever past the end.  Only
    // a scan shorter than one tile is ragged.
    const int t_first = (a.n_chunk >= KT && (tile + 1) * KT > a.n_chunk) ? a.n_chunk - KT : tile * KT;
    const unsigned lane_addr = (unsigned)(uintptr_t)((lds_f64 *)win) + (unsigned)lane * 16u;

    double vsum[J];                                     // per wave, per sample slot
#pragma unroll
    for (int j = 0; j < J; ++j) vsum[j] = 0.0;
    // thread k < KT: running (maximum, lowest index) of sample k of the tile over the bricks
    double tmax = -__builtin_inf();
    int tidx = INT32_MAX;

    for (int b = group; b < g.nbricks; b += a.ngroups) {
        if (!pair_fits(a.brick_total[b], S, KT, a.cap_doubles * 8)) continue;   // direct kernel's job
        __syncthreads();                              // previous brick's merge has read LDS
        stage_pair_windows<JP>(a, win, b, wave, nwaves, lane, t_first);
        __syncthreads();

        int x0, y0, z0, vx, vy, vz;
        brick_extents(g, b, x0, y0, z0, vx, vy, vz);
        const int nvalid = vx * vy * vz;
        int lz = wave % vz, ly = (wave / vz) % vy, lx = wave / (vz * vy);
        const uint16_t *brick_rel = a.rel + (int64_t)b * g.brick_nodes * g.row_pad;

        double bmax[J];
        int bidx[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            bmax[j] = -__builtin_inf();
            bidx[j] = INT32_MAX;
        }
        uint4 q[NCH];                                  // offsets of the node about to be stacked
        {
            const uint16_t *p = brick_rel + (int64_t)(wave < nvalid ? wave : 0) * g.row_pad;
#pragma unroll
            for (int c = 0; c < NCH; ++c) q[c] = load_offsets(p, c * 8);
        }
        Epilogue<J> epi;
        bool pending = false;                          // wave-uniform: epi holds a node
        for (int m = wave; m < nvalid; m += nwaves) {
            const int node = ((x0 + lx) * g.ny + (y0 + ly)) * g.nz + (z0 + lz);
            lz += nwaves;
            while (lz >= vz) { lz -= vz; ++ly; }
            while (ly >= vy) { ly -= vy; ++lx; }
            // the node after this one (or a harmless reload of this one at the end)
            const uint16_t *next =
                brick_rel + (int64_t)(m + nwaves < nvalid ? m + nwaves : m) * g.row_pad;

            // the next node's last offset chunk: a load issued at the end of the node would be
            // waited for at once, by the register copies at the loop's back edge
            const uint4 q_last = load_offsets(next, (NCH - 1) * 8);
            double acc[J];
            v2d even[JP], odd[JP];
            pissue<JP, S, 0>(even, q, next, lane_addr);
            if (pending)
                pbatch<JP, VOLUME, TAIL, S, true, 0>(acc, even, odd, q, next, lane_addr, epi, vsum,
                                                       bmax, bidx, a, t_first, lane);
            else
                pbatch<JP, VOLUME, TAIL, S, false, 0>(acc, even, odd, q, next, lane_addr, epi,
                                                        vsum, bmax, bidx, a, t_first, lane);
#pragma unroll
            for (int j = 0; j < J; ++j) {                  // z: log2 of the coalescence (rounded
#pragma clang fp contract(off)                             // product: see finish_node)
                epi.x[j] = acc[j] * a.z_scale;
            }
            epi.node = node;
            if (VOLUME) epi.row = a.volume + ((int64_t)node * a.vol_stride + t_first);
            q[NCH - 1] = q_last;
            pending = true;
        }
        if (pending)                                   // the brick's last node: not overlapped
            pepi_steps<JP, VOLUME, TAIL, 0, XEpiSteps<VOLUME>::value>(epi, vsum, bmax, bidx, a,
                                                                        t_first, lane);

        // ---- merge the brick's maxima across the waves (the windows are dead now)
        if (a.want_scan) {
            double *smax = win;
            int *sidx = reinterpret_cast<int *>(win + (size_t)nwaves * KT);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int k = wave * KT + pair_slot_sample<JP>(lane, j);
                smax[k] = bmax[j];
                sidx[k] = bidx[j];
            }
            __syncthreads();
            if ((int)threadIdx.x < KT) {
                const int k = threadIdx.x;
                for (int w = 0; w < nwaves; ++w) {
                    const double v = smax[w * KT + k];
                    const int i = sidx[w * KT + k];
                    if (better(v, i, tmax, tidx)) {
                        tmax = v;
                        tidx = i;
                    }
                }
            }
        }
    }
    if (!a.want_scan) return;
    // ---- this workgroup's partial set: sums across the waves, (max, idx) from the threads
    {
        double *ssum = win;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < J; ++j) ssum[wave * KT + pair_slot_sample<JP>(lane, j)] = vsum[j];
        __syncthreads();
        if ((int)threadIdx.x < KT) {
            const int k = threadIdx.x;
            double total = 0.0;
            for (int w = 0; w < nwaves; ++w) total += ssum[w * KT + k];
            const int t = t_first + k;
            if (t < a.n_chunk) {
                const int64_t o = (int64_t)(a.set0 + group) * (a.part_stride ? a.part_stride : a.n_chunk) + t;
                a.part_max[o] = tmax;
                a.part_idx[o] = tidx == INT32_MAX ? kNoIndex : (int64_t)tidx;
                a.part_sum[o] = total;
            }
        }
    }
}

template <int JP, bool VOLUME, int S>
__global__ __launch_bounds__(1024) void stack_pair_kernel(StackArgs a_launch) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    constexpr int KT = 128 * JP;
    // (several timesteps per launch -- pair = 2 puts the fused detect here: this workgroup's step's
    // onsets and its columns of the partial sets, as in the other stacking kernels)
    const StackArgs a = step_view(a_launch);
    int tile, group;
    stack_tile_group(a, tile, group);
    if (group >= a.ngroups) return;                   // grid is padded to a multiple of 8 groups
    if (a.run_if != nullptr && *a.run_if == 0) return;
    // only the volume-writing variant cares where its tile lies in the scan (see t_first in the
    // body and the store step of pepi_step)
    if (VOLUME && a.n_chunk < KT) stack_pair_body<JP, VOLUME, 2, S>(a, win);
    else if (VOLUME && (tile + 1) * KT > a.n_chunk) stack_pair_body<JP, VOLUME, 1, S>(a, win);
    else stack_pair_body<JP, VOLUME, 0, S>(a, win);
}

}  // namespace qm
