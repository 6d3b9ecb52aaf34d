// qm_shift.hpp -- the shift-reuse float64 stacking kernel (fused detect), round 3.
//
// Same arithmetic as the other stacking kernels (qm_kernels.hpp): float64 sums in ascending row
// order per (node, sample) -- migratelib.c:54-59 --, 2^z, running maximum / first index / sum.
// What changes is how many operands leave LDS.  The round-2 kernels read one 8-byte operand per
// add; on gfx950 the data returning from LDS and the float64 adds share the SIMD's cycles
// (2 cycles per 8-byte operand + 4 per add), so operands are the lever.
//
//   * a lane owns FOUR CONSECUTIVE samples, t = t_first + 4*lane + k (time tile = 256 samples);
//   * a wavefront stacks a 2x2x2 GROUP of nodes at a time: for one table row their delays differ
//     by a few samples, so all eight nodes' operands are one WINDOW of 4 + (max - min delay)
//     consecutive samples per lane, read once into registers (2.2 operands per node-row at C3
//     instead of 4);
//   * node g's four adds take window registers [idx_g, idx_g + 4): a wave-uniform, data-dependent
//     register index -- gfx9 VGPR-index mode (gen_shift_asm.py, where the schedule is described).
//
// LDS: per brick the S row windows, each de-interleaved into two planes of 16-byte slots (plane A
// slot s = window samples 4s, 4s+1; plane B = 4s+2, 4s+3) so that a window quad of every lane is
// one aligned, conflict-free ds_read_b128 per plane.  Two 4-wave workgroups per CU, 80 KB each.
// The per-(group, row) schedule (register indices, window address, quad count) is a stream of
// 32-byte records built once per table (shift_stream_kernel) and read with scalar loads.
#pragma once

#include "qm_kernels.hpp"

namespace qm {

// Row blocks: the rows of the workgroup's NEXT block that a wavefront's row loop stages itself while it adds
// the current one (gen_shift_asm.py, stage_step).  rows = 0: nothing (no block follows, or the kernel has
// staged it: a tile whose windows reach past the onsets' last samples, a wavefront without a group).
struct ShiftStageNext {
    const void *meta;        // the block's row-window records (smeta: 16 bytes per row)
    int rows;                // rows of the block
    int first_row, stride;   // this wavefront's rows: first_row, first_row + stride, ...
    const void *src;         // onsets + (the block's first table row) * T + fsmp + sample0 + t_first
    unsigned row_bytes;      // 8 T
    unsigned lds;            // LDS byte address of the half the block goes to
    unsigned lane32;         // lane * 32
};

#ifndef QM_SHIFT_ASM_INC                     // (development: tools/shift_variants.sh swaps the loop)
#define QM_SHIFT_ASM_INC "qm_shift_asm.inc"
#endif
#include QM_SHIFT_ASM_INC

// Tables of more than 64 rows: the row-block kernels at the end of this file (one group per
// wavefront, its accumulators in registers while the rows pass through LDS block by block).
// Up to 64 rows, three workgroup shapes: two 4-wave workgroups per CU with 80 KB each (up to ~32 rows;
// running state in registers, two wavefronts per SIMD); ONE 8-wave workgroup that owns all 160 KB
// (33-64 rows: twice the rows' windows; plane B lies beyond a DS instruction's 16-bit offset and
// gets its own address register); or ONE 12-wave workgroup per CU (opt-in: three per SIMD, the
// running state of every wavefront in LDS behind the windows so that the loop fits 168 VGPRs).
constexpr int kShiftWaves = 4;                          // wavefronts per workgroup (default shape)
constexpr int kShiftWaves8 = 8;
constexpr int kShiftWaves3 = 12;
constexpr int kShiftKT = 256;                           // samples per time tile
// Round 6, WIDE tiles: six samples per lane, time tile 384.  A register window then feeds 48 adds instead
// of 32 (the row loop's cost is its LDS reads and per-row bookkeeping), and with a
// lane stride of 48 bytes 16-byte reads of a CONTIGUOUS row window are conflict-free, so the window may start
// at any EVEN sample (two-plane layout: a multiple of four).  384-sample windows of ~30 rows need more than
// 80 KB: the 8-wave workgroup that owns a CU's LDS.
constexpr int kShiftWideKT = kWave * kShiftWideSpl;
constexpr int kShiftLdsBytes = 2 * kShiftPlane;         // both planes
constexpr int kShiftStateBytes = 5 * kShiftStateChunk;  // per wavefront (12-wave shape)
constexpr int kShiftLdsBytes3 = 2 * kShiftPlane3 + kShiftWaves3 * kShiftStateBytes;
static_assert(kShiftLdsBytes3 <= 160 * 1024, "the 12-wave workgroup's LDS exceeds a CU's");
constexpr int kShiftLdsBytes8 = 2 * kShiftPlane8;
static_assert(kShiftLdsBytes8 <= 160 * 1024, "the 8-wave workgroup's LDS exceeds a CU's");
__host__ __device__ constexpr int shift_plane(int nw) {
    return nw == kShiftWaves3 ? kShiftPlane3 : nw == kShiftWaves8 ? kShiftPlane8 : kShiftPlane;
}
__host__ __device__ constexpr int shift_lds_bytes(int nw) {
    return nw == kShiftWaves3 ? kShiftLdsBytes3 : nw == kShiftWaves8 ? kShiftLdsBytes8 : kShiftLdsBytes;
}
constexpr int kShiftMaxRows = 64;                       // table rows the stream builder handles
constexpr int kShiftHalfBytes = 80 * 1024;              // row blocks, double-buffered: one half of a CU's LDS
static_assert(2 * kShiftPlane <= kShiftHalfBytes, "a half holds both planes");
static_assert(QM_EXP2_DEGREE_SUM == 8 && QM_EXP2_DEGREE_VOLUME == kShiftVolumeDegree,
              "the generated loops carry the degree-8 (detect) 2^f and the stored values' degree");

// Record geometry: 64 bytes with the eight register indices as dwords, or 32 bytes with them as bytes
// of two dwords ("packed": gen_shift_asm.py says which loops take which).  Row blocks and the other
// loops may differ; a table's stream is built in the format of the kernels that will read it.
__host__ __device__ constexpr bool shift_packed(bool blocks) { return blocks ? kShiftPackedBlocks : kShiftPackedGroups; }
__host__ __device__ constexpr int shift_rec_bytes(bool blocks) { return shift_packed(blocks) ? 32 : 64; }
constexpr int kShiftRec = shift_rec_bytes(false);       // ... of the loops that take all groups of a brick per call
constexpr int kShiftRecBlocks = shift_rec_bytes(true);  // ... of the row-block loops
// dwords of the next row's header (LDS offset, quad count) / of the group's first node and valid mask
__host__ __device__ constexpr int shift_rec_hdr(bool packed) { return packed ? 2 : 8; }
__host__ __device__ constexpr int shift_rec_base(bool packed) { return packed ? 4 : 10; }

// records per (brick, wave): lead-in + groups * rows2 + trailing pad; every (brick, wave) owns a
// fixed-size run (a brick at the grid's edge uses a prefix of it)
__host__ __device__ __forceinline__ int shift_groups_per_brick(const GridDesc &g) {
    return (g.bx / 2) * (g.by / 2) * (g.bz / 2);
}
__host__ __device__ __forceinline__ int64_t shift_recs_per_wave(const GridDesc &g, int rows2, int nw) {
    return (int64_t)((shift_groups_per_brick(g) + nw - 1) / nw) * rows2 + 2;
}
// first record of the run of (brick b, wave w, row block k): a wavefront's blocks of one brick are
// contiguous (the loop's L2 prefetch runs from one into the next)
__host__ __device__ __forceinline__ int64_t shift_run_record(int64_t b, int w, int k, int nw, int nblk,
                                                             int64_t rpw) {
    return ((b * nw + w) * nblk + k) * rpw;
}

struct ShiftArgs {
    StackArgs a;                 // grid (shift bricks), onsets, scan geometry, partial sets
    const int4 *smeta;           // [nbricks][S] (min delay, span, first slot, slots) per row window
    const int32_t *stotal;       // [nbricks] slots of all rows (the zero row of an odd S follows)
    const int32_t *sfit;         // [nbricks] 1: the brick runs here, 0: direct kernel
    const char *stream;          // records, [nbricks][nw][shift_recs_per_wave]
    int rows2;                   // stream rows per group: S rounded up to even
    int nw;                      // wavefronts per workgroup the stream was dealt for
    int lazy;                    // detect: the loop flavour that recovers the arg-max lazily
    int nblk;                    // row blocks per brick (1: all rows of a brick are staged at once)
    int sb;                      // rows per block (even; the last block may hold fewer)
    int stage_slots;             // row blocks: the largest slot count of a row window (the loop's staging: <= 127)
    int stage_reach;             // ... and the furthest sample past a tile's first that a window holds
    // wide tiles (a.wide_tiles > 0: the launch's first tiles hold 384 samples, six per lane): their own row-window
    // records and stream on the same brick grid and deal; sfit is the verdict for BOTH tile kinds
    const int4 *wmeta;
    const int32_t *wtotal;
    const char *wstream;
};

// a wavefront that sees at least this many 2x2x2 groups between two resets of its running
// maximum takes the lazy flavour (gen_shift_asm.py, epilogue_node)
constexpr int kShiftLazyGroups = 160;

// What a launch produces besides the partial sets of the scan
constexpr int kShiftDetect = 0;     // nothing: the fused detect
constexpr int kShiftVolume = 1;     // the 4-D volume (whole tiles: two 16-byte stores per node and lane, a
                                    // pulled-back last tile masks the lanes its predecessor stores; tail
                                    // tiles: one masked 8-byte store per sample slot)
constexpr int kShiftMarginal = 2;   // the marginalised map: per (tile, node) the sum over the tile's own
                                    // samples inside [m0, m1) -> a.marginal[tile][node]

struct LaunchShape;
const char *shift_unit_defines();                                           // qm_launch_shift.hip
hipError_t launch_shift_detect(const ShiftArgs &a, const LaunchShape &s);
hipError_t launch_shift_volume(const ShiftArgs &a, const LaunchShape &s);
hipError_t launch_shift_marginal(const ShiftArgs &a, const LaunchShape &s);
hipError_t launch_shift_marginal8(const ShiftArgs &a, const LaunchShape &s);
hipError_t launch_shift_detect3(const ShiftArgs &a, const LaunchShape &s);  // the 12-wave shape
hipError_t launch_shift_detect8(const ShiftArgs &a, const LaunchShape &s);  // the 8-wave shape
hipError_t launch_shift_volume8(const ShiftArgs &a, const LaunchShape &s);
hipError_t launch_shift_rows8(const ShiftArgs &a, const LaunchShape &s);    // row blocks (> 64 rows)
hipError_t launch_shift_rows2(const ShiftArgs &a, const LaunchShape &s);    // ... double-buffered, LDS-direct
hipError_t launch_shift_rows2_volume(const ShiftArgs &a, const LaunchShape &s);
hipError_t launch_shift_rows4(const ShiftArgs &a, const LaunchShape &s);    // ... two 4-wave workgroups per CU
hipError_t launch_shift_rows4_volume(const ShiftArgs &a, const LaunchShape &s);
hipError_t launch_shift_wide_rows(const ShiftArgs &a, const LaunchShape &s);    // row blocks on wide tiles
hipError_t launch_shift_detect_sets(const ShiftArgs &a, const LaunchShape &s);  // qm_launch_shift_sets.hip: a partial
hipError_t launch_shift_detect8_sets(const ShiftArgs &a, const LaunchShape &s); // set per brick (tie_rule = 1)

// valid 2x2x2 groups of a brick form a box [0,cx) x [0,cy) x [0,cz) in group coordinates
__device__ __forceinline__ void shift_group_box(const GridDesc &g, int b, int &x0, int &y0, int &z0,
                                                int &vx, int &vy, int &vz, int &cx, int &cy,
                                                int &cz) {
    brick_extents(g, b, x0, y0, z0, vx, vy, vz);
    cx = (vx + 1) / 2; cy = (vy + 1) / 2; cz = (vz + 1) / 2;
}

// delays of the (up to) eight nodes of group (gx, gy, gz) for row r, relative to the row's brick
// minimum; invalid nodes (outside the grid) copy node 0.  Returns the valid-node mask.
__device__ __forceinline__ unsigned shift_group_delays(const GridDesc &g, const int32_t *lut, int x0,
                                                       int y0, int z0, int vx, int vy, int vz,
                                                       int gx, int gy, int gz, int r, int lo,
                                                       int (&d)[8]) {
    unsigned mask = 0;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int lx = 2 * gx + (n >> 2), ly = 2 * gy + ((n >> 1) & 1), lz = 2 * gz + (n & 1);
        if (lx < vx && ly < vy && lz < vz) {
            const int64_t node = ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + (z0 + lz);
            int t = lut[node * g.n_rows + r];
            t = t < 0 ? 0 : t;                            // migratelib.c:55
            d[n] = t - lo;
            mask |= 1u << n;
        } else {
            d[n] = d[0];                                  // node 0 of a valid group is valid
        }
    }
    return mask;
}

// register window of a (group, row): first sample e0 (relative to the brick's row minimum) and quads of four
// doubles it spans.  wide: six samples per lane from an even sample (kShiftWideSpl), else four from a multiple
// of four.
__device__ __forceinline__ void shift_window(const int (&d)[8], int wide, int &e0, int &nq) {
    int dmin = d[0], dmax = d[0];
#pragma unroll
    for (int n = 1; n < 8; ++n) {
        dmin = d[n] < dmin ? d[n] : dmin;
        dmax = d[n] > dmax ? d[n] : dmax;
    }
    e0 = wide ? dmin & ~1 : dmin & ~3;
    nq = (dmax - e0 + (wide ? kShiftWideSpl : 4) + 3) / 4;
    nq = nq < 2 ? 2 : nq;
}
// samples a lane is ahead of its predecessor / the furthest 4-double slot (exclusive) a lane may touch in a
// window that starts at sample e0 and fetches `fetched` quads
__host__ __device__ constexpr int shift_lane_step(int wide) { return wide ? kShiftWideSpl : 4; }
__host__ __device__ constexpr int shift_slots_touched(int wide, int e0, int fetched) {
    return (e0 + shift_lane_step(wide) * (kWave - 1) + 4 * fetched + 3) / 4;
}
// slots of the all-zero window that the padding row of an odd row count reads
__host__ __device__ constexpr int shift_nq_min(int wide) { return wide ? kShiftNqMinWide : kShiftNqMin; }
__host__ __device__ constexpr int shift_zero_slots(int wide) {
    return wide ? shift_slots_touched(1, 0, kShiftNqMinWide) : kWave + kShiftNqMin;
}

#ifdef QM_TU_TABLES
// Pass 1, one workgroup per brick: slots every row window needs (the furthest slot a lane may
// touch: e0/4 + 63 + the quads fetched), their prefix, and whether the brick fits.
// meta_raw = (min, span, ., .) per (brick, row) from brick_minmax_kernel.
__global__ __launch_bounds__(256) void shift_need_kernel(GridDesc g, const int32_t *__restrict__ lut,
                                                         const int4 *__restrict__ meta_raw,
                                                         int4 *__restrict__ smeta,
                                                         int32_t *__restrict__ stotal,
                                                         int32_t *__restrict__ sfit,
                                                         unsigned long long *__restrict__ tally,
                                                         int plane_bytes, int nblk, int sb, int wide) {
    // tally[0] += quads the loop fetches, tally[1] += (group, row) pairs: operands per add; tally[2] = the
    // largest slot count of a row window, tally[3] = the furthest window sample past a tile's first one (what
    // the row-block kernels need to know before they let the row loop stage: ShiftArgs::stage_slots / _reach)
    // Row blocks (nblk > 1): workgroup vb = (brick, block) handles rows [k sb, k sb + S) of brick b;
    // smeta / stotal / sfit are per (brick, block), sb rows apart.
    __shared__ int need[kShiftMaxRows];
    __shared__ int overflow;
    const int vb = blockIdx.x, b = vb / nblk, r0 = (vb % nblk) * sb;
    const int S = g.n_rows - r0 < sb ? g.n_rows - r0 : sb;
    int x0, y0, z0, vx, vy, vz, cx, cy, cz;
    shift_group_box(g, b, x0, y0, z0, vx, vy, vz, cx, cy, cz);
    for (int r = threadIdx.x; r < S; r += blockDim.x) need[r] = 0;
    if (threadIdx.x == 0) overflow = 0;
    __syncthreads();
    const int nvg = cx * cy * cz;
    unsigned quads = 0;
    for (int i = threadIdx.x; i < nvg * S; i += blockDim.x) {
        const int j = i / S, r = i % S;
        const int gz = j % cz, gy = (j / cz) % cy, gx = j / (cz * cy);
        int d[8], e0, nq;
        shift_group_delays(g, lut, x0, y0, z0, vx, vy, vz, gx, gy, gz, r0 + r,
                           meta_raw[(int64_t)b * g.n_rows + r0 + r].x, d);
        shift_window(d, wide, e0, nq);
        if (nq > kShiftNqMax) atomicOr(&overflow, 1);
        const int fetched = nq > shift_nq_min(wide) ? nq : shift_nq_min(wide);
        quads += (unsigned)fetched;
        atomicMax(&need[r], shift_slots_touched(wide, e0, fetched));
    }
    atomicAdd(&tally[0], (unsigned long long)quads);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&tally[1], (unsigned long long)nvg * S);
        int run = 0;
        for (int r = 0; r < S; ++r) {
            const int4 raw = meta_raw[(int64_t)b * g.n_rows + r0 + r];
            smeta[(int64_t)vb * sb + r] = make_int4(raw.x, raw.y, run, need[r]);
            run += need[r];
            atomicMax(&tally[2], (unsigned long long)need[r]);
            const int reach = raw.x + 4 * need[r];
            atomicMax(&tally[3], (unsigned long long)(reach > 0 ? reach : 0));
        }
        stotal[vb] = run;
        const int zero_row = (S & 1) ? shift_zero_slots(wide) : 0;   // all-zero window of the padding row
        // (two planes of 16-byte slots; wide tiles: ONE contiguous region of 32-byte slots, plane_bytes = half of it)
        sfit[vb] = (!overflow && (int64_t)(run + zero_row) * 16 <= plane_bytes) ? 1 : 0;
    }
}

// Pass 2, one workgroup per brick: the record stream (format: gen_shift_asm.py).
__global__ __launch_bounds__(256) void shift_stream_kernel(GridDesc g, const int32_t *__restrict__ lut,
                                                           const int4 *__restrict__ smeta,
                                                           const int32_t *__restrict__ stotal,
                                                           const int32_t *__restrict__ sfit, int rows2max,
                                                           int nw, int nblk, int sb, int packed, int wide,
                                                           uint32_t *__restrict__ stream) {
    extern __shared__ uint2 hdr[];                      // [group j][row] (LDS address, quads)
    // (row blocks: workgroup vb = (brick, block); sfit is per brick -- all of its blocks fit)
    const int vb = blockIdx.x, b = vb / nblk, k = vb % nblk, r0 = k * sb;
    const int S = g.n_rows - r0 < sb ? g.n_rows - r0 : sb;
    const int rows2 = S + (S & 1);
    if (!sfit[b]) return;
    int x0, y0, z0, vx, vy, vz, cx, cy, cz;
    shift_group_box(g, b, x0, y0, z0, vx, vy, vz, cx, cy, cz);
    const int nvg = cx * cy * cz;
    const int64_t rpw = shift_recs_per_wave(g, rows2max, nw);
    auto record = [&](int w, int64_t i) {
        return stream + (shift_run_record(b, w, k, nw, nblk, rpw) + i) * (packed ? 8 : 16);
    };
    for (int i = threadIdx.x; i < nvg * rows2; i += blockDim.x) {
        const int j = i / rows2, r = i % rows2;
        const int w = j % nw, pos = j / nw;
        uint32_t *rec = record(w, 1 + (int64_t)pos * rows2 + r);
        if (r >= S) {                                   // padding row of an odd S: adds 0.0
            for (int n = 0; n < (packed ? 2 : 8); ++n) rec[n] = 0;
            hdr[j * rows2 + r] = make_uint2(16u * (unsigned)stotal[vb], 2u);
            continue;
        }
        const int gz = j % cz, gy = (j / cz) % cy, gx = j / (cz * cy);
        const int4 m = smeta[(int64_t)vb * sb + r];
        int d[8], e0, nq;
        const unsigned mask = shift_group_delays(g, lut, x0, y0, z0, vx, vy, vz, gx, gy, gz, r0 + r, m.x, d);
        shift_window(d, wide, e0, nq);
        if (packed) {                                   // the eight register indices (< 48) as bytes
            uint32_t lo = 0, hi = 0;
            for (int n = 0; n < 4; ++n) {
                lo |= (2u * (unsigned)(d[n] - e0)) << (8 * n);
                hi |= (2u * (unsigned)(d[4 + n] - e0)) << (8 * n);
            }
            rec[0] = lo;
            rec[1] = hi;
        } else {
            for (int n = 0; n < 8; ++n) rec[n] = 2u * (unsigned)(d[n] - e0);
        }
        // (the loops of contiguous windows double this: byte 32 z + 8 e0 -- wide tiles: e0 is even, not a
        // multiple of four)
        hdr[j * rows2 + r] = make_uint2(16u * (unsigned)m.z + 4u * (unsigned)e0, (unsigned)nq);
        if (r == 0) {
            rec[shift_rec_base(packed)] = (uint32_t)(((int64_t)(x0 + 2 * gx) * g.ny + (y0 + 2 * gy)) * g.nz + (z0 + 2 * gz));
            rec[shift_rec_base(packed) + 1] = mask;
        }
    }
    __syncthreads();
    // headers travel one record ahead of their row
    for (int i = threadIdx.x; i < nvg * rows2 + nw; i += blockDim.x) {
        if (i >= nvg * rows2) {                          // lead-in record of wave w
            const int w = i - nvg * rows2;
            if (w < nvg) {
                const uint2 h = hdr[w * rows2];
                uint32_t *rec = record(w, 0);
                rec[shift_rec_hdr(packed)] = h.x;
                rec[shift_rec_hdr(packed) + 1] = h.y;
            }
            continue;
        }
        const int j = i / rows2, r = i % rows2;
        const int w = j % nw, pos = j / nw;
        uint32_t *rec = record(w, 1 + (int64_t)pos * rows2 + r);
        uint2 h = make_uint2(0u, 2u);                    // after the wave's last row: harmless
        if (r + 1 < rows2) h = hdr[j * rows2 + r + 1];
        else if (j + nw < nvg) h = hdr[(j + nw) * rows2];
        rec[shift_rec_hdr(packed)] = h.x;
        rec[shift_rec_hdr(packed) + 1] = h.y;
    }
}
#endif  // QM_TU_TABLES

#ifdef QM_SHIFT_TU
// Stage the row windows of brick b for the tile starting at t_first: window sample u of row r goes
// to plane (u & 2) / 2, slot first_r + u / 4, half u & 1.  Every slot of the row is written (zero
// past the data the brick can touch and past the rows' end): a lane may fetch whole quads.
// Row blocks: vb = (brick, block), the block's S rows start at table row row0, metadata sb rows apart.
// CONTIG (tail tiles of KT = 64, 128 or 192 samples, gen_shift_asm.py): one plane, window sample u of
// row r at byte 8 u of the row's region (which starts at twice the two-plane layout's slot offset).
template <int NW, int RB = (NW == 12 ? 3 : 8), int U = 6, bool CONTIG = false, int KT = kShiftKT>
__device__ __forceinline__ void stage_shift_windows(const ShiftArgs &s, double *win, int vb, int row0,
                                                    int S, int sb, int wave, int lane, int t_first) {
    const StackArgs &a = s.a;
    constexpr bool kWide = KT == kShiftWideKT;
    const int4 *const smeta = kWide ? s.wmeta : s.smeta;
    const int32_t *const stotal = kWide ? s.wtotal : s.stotal;
    // A wavefront stages rows wave, wave + NW, ...: the loads of ALL its rows (up to RB x U x 64
    // samples) are issued before the first LDS store, so the brick's staging costs one round trip
    // to L2 instead of one per row (while a workgroup stages, its SIMDs' other wavefronts run at
    // half rate: a wavefront alone issues one float64 instruction per 8 cycles).
    // (RB rows per wavefront and pass, U 64-sample chunks per row and pass)
    constexpr int kPlane = shift_plane(NW);
    // LDS position (in doubles) of window sample u of a row whose first slot is z
    auto where = [&](int z, int u) {
        if constexpr (CONTIG) return 4 * z + u;
        else return ((u & 2) ? kPlane / 8 : 0) + 2 * (z + (u >> 2)) + (u & 1);
    };
    for (int r0 = wave; r0 < S; r0 += NW * RB) {
        double v[RB][U];
        int4 m[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int r = r0 + k * NW;
            m[k] = r < S ? smeta[(int64_t)vb * sb + r] : make_int4(0, 0, 0, 0);
            const int len = m[k].y + KT;                           // samples the brick can touch
            const int first = m[k].x + a.fsmp + a.sample0 + t_first;   // index inside the row
            const int room = a.T - first;
            const double *src = a.onsets + (int64_t)(r < S ? row0 + r : 0) * a.T + first;
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = kWave * i + lane;
                v[k][i] = (r < S && u < len && u < room) ? src[u] : 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int total = 4 * m[k].w;                          // every slot of the row is written
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = kWave * i + lane;
                if (u < total) win[where(m[k].z, u)] = v[k][i];
            }
            // (a row window longer than U x 64 samples: the rest, one load at a time)
            if (total > kWave * U) {
                const int r = r0 + k * NW;
                const int len = m[k].y + KT;
                const int first = m[k].x + a.fsmp + a.sample0 + t_first;
                const int room = a.T - first;
                const double *src = a.onsets + (int64_t)(row0 + r) * a.T + first;
                for (int u = kWave * U + lane; u < total; u += kWave)
                    win[where(m[k].z, u)] = (u < len && u < room) ? src[u] : 0.0;
            }
        }
    }
    if ((S & 1) && wave == 0) {                                    // the padding row's zero window
        const int z = stotal[vb];
        for (int u = lane; u < 4 * shift_zero_slots(kWide); u += kWave) win[where(z, u)] = 0.0;
    }
}

// Wide tiles: the contiguous row windows of brick b by LDS-DIRECT loads (global_load_lds_dwordx4: 16 bytes per
// lane from global memory to LDS at M0 + 16 lane, no registers -- the group loop leaves the compiler 48 VGPRs,
// and a staging through registers spilled the wavefront's running state around every brick).  A wavefront
// stages rows wave, wave + NW, ...; everything about a row is wave-uniform except the lane's 16 bytes: M0 = the
// chunk's LDS address, the chunk's global address as the instruction's scalar base, EXEC = the chunk's pairs.
// Samples behind what the brick can touch are whatever follows them in memory (fetched with a window, never
// added); a row whose window reaches past the onsets' last sample -- only the scan's last tiles at the largest
// delays -- goes through registers, zero-filled.  The caller waits (vmcnt) before its barrier.
// (row blocks: vb = (brick, block), the block's S rows start at table row row0, metadata sb rows apart)
template <int NW>
__device__ __forceinline__ void stage_shift_wide(const ShiftArgs &s, double *win, int vb, int S, int wave,
                                                 int lane, int t_first, int row0 = 0, int sb = 0) {
    const StackArgs &a = s.a;
    using int4s = int __attribute__((ext_vector_type(4)));
    const int4s *meta = reinterpret_cast<const int4s *>(s.wmeta + (int64_t)vb * (sb ? sb : S));
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_f64 *)win);
    const unsigned lane16 = (unsigned)lane * 16u;
    for (int r = wave; r < S; r += NW) {
        const int4s m = meta[__builtin_amdgcn_readfirstlane(r)];          // (min delay, span, first slot, slots)
        const int first = __builtin_amdgcn_readfirstlane(m.x) + a.fsmp + a.sample0 + t_first;
        const int total = 4 * __builtin_amdgcn_readfirstlane(m.w);         // doubles, all of them written
        const int z4 = 4 * __builtin_amdgcn_readfirstlane(m.z);
        const int room = a.T - first;
        const double *src = a.onsets + (int64_t)(row0 + r) * a.T + first;
        if (room >= total) {
            const unsigned long long sp = (unsigned long long)src;
            const double *usrc = (const double *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sp >> 32)) << 32) |
                                                  (unsigned)__builtin_amdgcn_readfirstlane((int)sp));
            for (int c = 0; c < total; c += 2 * kWave) {
                const int pairs = (total - c) / 2 < kWave ? (total - c) / 2 : kWave;
                unsigned long long saved;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_lshr_b64 exec, -1, %[sh]\n\t"
                             "s_mov_b32 m0, %[ma]\n\t"
                             "s_nop 0\n\t"
                             "global_load_lds_dwordx4 %[vo], %[src]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved)
                             : [sh] "s"(__builtin_amdgcn_readfirstlane(kWave - pairs)),
                               [ma] "s"(__builtin_amdgcn_readfirstlane((int)(lds_base + 8u * (unsigned)(z4 + c)))),
                               [vo] "v"(lane16), [src] "s"(usrc + c)
                             : "memory", "m0");
            }
        } else {
            const int len = __builtin_amdgcn_readfirstlane(m.y) + kShiftWideKT;
            for (int u = lane; u < total; u += kWave) win[z4 + u] = (u < len && u < room) ? src[u] : 0.0;
        }
    }
    if ((S & 1) && wave == 0) {                                    // the padding row's zero window
        const int z4 = 4 * s.wtotal[vb];
        for (int u = lane; u < 4 * shift_zero_slots(1); u += kWave) win[z4 + u] = 0.0;
    }
}

// What a workgroup of the shift-reuse kernels works on: XCD-aware (time tile, brick group) map as
// stack_lds_kernel -- group g runs on XCD g mod 8 --, the grid is padded to a multiple of 8 groups.
// Tiles are 256 samples; what a scan leaves beyond its whole tiles is either (a.tail_spl = 1, 2, 3)
// ONE tail tile of 64 / 128 / 192 samples starting where they end -- `spl` samples per lane, its own
// loop flavour (gen_shift_asm.py) -- or (a.tail_spl = 0: a remainder of more than 192 samples, and the
// kernels without tail flavours) a last whole tile pulled back so that it ends with the scan (it
// overlaps its predecessor: same arithmetic, same bits).
// Round 6: the launch's first a.wide_tiles tiles are WIDE (384 samples, six per lane; the last of them pulled
// back if the scan ends inside it), the 256-sample tiles and the tail tile follow behind them.
struct ShiftWork {
    int tile, group, spl;
    int t_own;                   // first sample the tile is there for
    int t_first;                 // first sample it computes (< t_own: pulled back over its predecessor)
    bool run;
};
__device__ __forceinline__ ShiftWork shift_work(const StackArgs &a) {
    ShiftWork w;
    stack_tile_group(a, w.tile, w.group);
    w.run = w.group < a.ngroups && !(a.run_if != nullptr && *a.run_if == 0);
    if (w.tile < a.wide_tiles) {
        w.spl = kShiftWideSpl;
        w.t_own = w.tile * kShiftWideKT;
        w.t_first = w.t_own + kShiftWideKT > a.n_chunk ? a.n_chunk - kShiftWideKT : w.t_own;
        return w;
    }
    w.spl = (a.tail_spl > 0 && w.tile == a.ntiles - 1) ? a.tail_spl : 4;
    w.t_own = a.wide_tiles * kShiftWideKT + (w.tile - a.wide_tiles) * kShiftKT;
    w.t_first = (w.spl == 4 && w.t_own + kShiftKT > a.n_chunk && a.n_chunk >= kShiftKT)
                    ? a.n_chunk - kShiftKT : w.t_own;
    return w;
}
// a wavefront's running (max z, sum of 2^z, first index) of its J samples per lane
template <int J>
__device__ __forceinline__ void shift_reset(double (&vmax)[J], double (&vsum)[J], int (&vidx)[J]) {
#pragma unroll
    for (int k = 0; k < J; ++k) {
        vmax[k] = -__builtin_inf();
        vsum[k] = 0.0;
        vidx[k] = INT32_MAX;
    }
}

// The workgroup's wavefronts hold (max, sum, index) of the tile's 64 J samples, J per lane:
// combine them through LDS (thread k owns sample k) and publish the workgroup's partial set.
// Call after a barrier behind the last use of `win`.
// (t_own: the first sample the tile is there for.  A tile pulled back over its predecessor computes that one's last
// samples a second time; it publishes its OWN only -- round 6: the two used to write the same entries, which is
// harmless while both run the same loop flavour and a last-bits race in max_norm_coa when they do not: with
// tie_rule = 1 a wide tile runs the lazy loop and the 256-sample tile behind it the eager one, whose sums' terms
// differ in the last bits; found by the fuzz campaign as a batch that differed from its steps in one sample.)
template <int NW, int J = 4>
__device__ __forceinline__ void shift_publish(const StackArgs &a, double *win, const double (&vmax)[J],
                                              const double (&vsum)[J], const int (&vidx)[J], int wave,
                                              int lane, int group, int t_first, int t_own = 0) {
    constexpr int KT = kWave * J;
    double *smax = win, *ssum = win + NW * KT;
    int *sidx = reinterpret_cast<int *>(win + 2 * NW * KT);
#pragma unroll
    for (int k = 0; k < J; ++k) {
        const int o = wave * KT + J * lane + k;
        smax[o] = vmax[k];
        ssum[o] = vsum[k];
        sidx[o] = vidx[k];
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k >= KT) return;                              // (more threads than samples: 8 / 12 waves, tail tiles)
    double best = smax[k], total = ssum[k];
    int bi = sidx[k];
    for (int w = 1; w < NW; ++w) {
        const double v = smax[w * KT + k];
        const int i = sidx[w * KT + k];
        total += ssum[w * KT + k];
        if (better(v, i, best, bi)) {
            best = v;
            bi = i;
        }
    }
    const int t = t_first + k;
    if (t < a.n_chunk && t >= t_own) {
        const int64_t o = (int64_t)(a.set0 + group) * (a.part_stride ? a.part_stride : a.n_chunk) + t;
        a.part_max[o] = best;
        a.part_idx[o] = bi == INT32_MAX ? kNoIndex : (int64_t)bi;
        a.part_sum[o] = total;
    }
}

// One (time tile, brick group) of a launch: J samples per lane (4: a whole tile on the two-plane
// layout; 1..3: the scan's tail tile on the contiguous layout; 6: a wide tile, contiguous layout, fused
// detect in the 8-wave shape only).
// (WIDE_LAZY: the wide tile's loop flavour, chosen outside the brick loop -- with both flavours' asm statements
// in one loop the compiler shuffled the 30 registers of running state between their operand assignments and
// spilled them around every brick)
// (SETS: the kernel flavours that also leave the largest z per BRICK and sample (a.brick_max), below -- flavours
// of their own because the extra code costs the plain kernels registers: nine spilled in the wide tile's loop.
// 1: the wavefronts' running maxima are folded and reset after every brick; 2, wide tiles: the generated loop
// raises the brick's row itself, the running state is the plain kernel's)
template <int MODE, int NW, int J, bool WIDE_LAZY = false, int SETS = 0>
__device__ __forceinline__ void shift_tile(const ShiftArgs &s, double *win, const ShiftWork &work,
                                           int lane, int wave) {
    constexpr bool kLdsState = NW == kShiftWaves3;
    constexpr bool kWide = J == kShiftWideSpl;
    constexpr bool kTail = J != 4;                      // (contiguous row windows)
    static_assert(!(kTail && kLdsState), "the 12-wave shape has no tail flavours");
    static_assert(!kWide || (MODE == kShiftDetect && NW == kShiftWaves8), "wide tiles: fused detect, 8 waves");
    static_assert(!SETS || (MODE == kShiftDetect && !kLdsState), "brick maxima: fused detect, 4 or 8 waves");
    static_assert(SETS != 2 || kWide, "brick maxima from the generated loop: wide tiles");
    const StackArgs &a = s.a;
    const GridDesc &g = a.g;
    const int tile = work.tile, group = work.group, t_first = work.t_first;
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_f64 *)win);
    const unsigned lane_addr = lds_base + (unsigned)lane * (kTail ? 8u * J : 16u);
    // volume, whole tiles: lanes of a pulled-back tile whose four samples its predecessor stores are
    // masked off at the stores (a lane that straddles the seam stores its four: same bits)
    const int seam = work.t_own - t_first;                         // samples of overlap, 0 .. 64 J - 1
    const unsigned long long store_lanes = ~0ull << (seam / 4);
    // volume, tail tiles: per sample slot the lanes whose sample lies inside the scan
    unsigned long long slot_lanes[J];
    // marginal: 1.0 for the lane's samples that are the tile's own (not its predecessor's) and lie
    // inside the window, 0.0 for the others
    double weight[J];
#pragma unroll
    for (int k = 0; k < J; ++k) {
        const int t = t_first + J * lane + k;
        slot_lanes[k] = __builtin_amdgcn_ballot_w64(t < a.n_chunk);
        weight[k] = (t >= work.t_own && t >= a.m0 && t < a.m1 && t < a.n_chunk) ? 1.0 : 0.0;
    }
    // marginal map, whole groups: the eight nodes' shares are summed over the wavefront together
    // (gen_shift_asm.py, marginal_butterfly); lane l ends with the total of node 4 b2 + 2 b0 + b1 of
    // its index -- its element's byte offset from the group's first node -- and fetches from the
    // lanes 16 and 32 away on the way
    const int lane_node = 4 * ((lane >> 2) & 1) + 2 * (lane & 1) + ((lane >> 1) & 1);
    const unsigned node_off = 8u * (unsigned)((lane_node & 4 ? g.ny * g.nz : 0) + (lane_node & 2 ? g.nz : 0) +
                                             (lane_node & 1));
    const unsigned lane_x16 = 4u * (unsigned)(lane ^ 16), lane_x32 = 4u * (unsigned)(lane ^ 32);
    (void)store_lanes; (void)slot_lanes; (void)weight; (void)node_off; (void)lane_x16; (void)lane_x32;

    double vmax[J], vsum[J];
    int vidx[J];
    shift_reset<J>(vmax, vsum, vidx);
    // 12-wave shape: the running state lives in LDS behind the windows, 5 chunks of 64 lanes x 16
    // bytes per wavefront (maxima 0-1 / 2-3, sums 0-1 / 2-3, indices)
    double *state = win + (2 * shift_plane(NW) + wave * kShiftStateBytes) / 8 + 2 * lane;
    const unsigned state_addr = (unsigned)(uintptr_t)((lds_f64 *)state);
    if constexpr (kLdsState) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            state[(k / 2) * (kShiftStateChunk / 8) + (k & 1)] = vmax[k];
            state[(2 + k / 2) * (kShiftStateChunk / 8) + (k & 1)] = vsum[k];
            reinterpret_cast<int *>(state + 4 * (kShiftStateChunk / 8))[k] = vidx[k];
        }
    }
    (void)state_addr;
    // 2^f: the stored values' polynomial where values are stored; the running sums' for the fused
    // detect and for the marginalised map (sums of positive terms: they inherit its 7.8e-13)
    constexpr int D = MODE == kShiftVolume ? Exp2Degree<true>::value
                      : MODE == kShiftMarginal ? kShiftMarginalDegree : Exp2Degree<false>::value;
    double c[D + 1];
#pragma unroll
    for (int i = 0; i <= D; ++i) c[i] = exp2_coeff<D>(i);

    const int64_t rpw = shift_recs_per_wave(g, s.rows2, NW);
    const int npairs = s.rows2 / 2, nz = g.nz, nynz = g.ny * g.nz;
    double *const marg_tile = a.marginal + (int64_t)tile * a.n_nodes;
    double *const vol_tile = a.volume + t_first;
    const unsigned vol_stride_bytes = (unsigned)(a.vol_stride * 8);
    (void)marg_tile; (void)vol_tile; (void)vol_stride_bytes;
    // Brick maxima (SETS; tie_rule = 1, qm_ties.hpp): besides its partial set the workgroup leaves, per brick of its
    // walk and sample of the tile, the largest z -- a.brick_max[brick][sample] -- so that the refinement stacks ONE
    // brick again per sample instead of everything a workgroup owns.
    //   SETS = 1: the wavefronts' (max, index) are folded after every brick -- across the wavefronts through LDS
    //   into the brick's row, and into the walk's own pair (wmax, widx: touched once per brick, the compiler may
    //   keep them in scratch) -- and start the next brick from (-inf, none); the sums run on.
    //   SETS = 2 (wide tiles): the generated loop itself raises the row, by atomic maxima from the groups that come
    //   within the tie slack of the wavefront's running maximum (gen_shift_asm.py, brick_max: the others cannot
    //   hold a candidate); rows start at -inf (the engine fills them); the running state is the plain kernel's.
    double wmax[J];
    int widx[J];
    (void)wmax; (void)widx;
    const int64_t brow_stride = a.part_stride ? a.part_stride : a.n_chunk;
    if constexpr (SETS == 1) {
#pragma unroll
        for (int k = 0; k < J; ++k) {
            wmax[k] = -__builtin_inf();
            widx[k] = INT32_MAX;
        }
    }
    // SETS = 1: the row of brick b from the wavefronts' registers (call behind a barrier that follows the brick)
    auto brick_row = [&](int b) {
        constexpr int KT = kWave * J;
#pragma unroll
        for (int k = 0; k < J; ++k) win[wave * KT + J * lane + k] = vmax[k];
        __syncthreads();
        const int j = threadIdx.x;
        if (j < KT && t_first + j < a.n_chunk && t_first + j >= work.t_own) {     // (its own samples, as shift_publish)
            double best = win[j];
            for (int w = 1; w < NW; ++w) best = win[w * KT + j] > best ? win[w * KT + j] : best;
            a.brick_max[(int64_t)b * brow_stride + t_first + j] = best;
        }
#pragma unroll
        for (int k = 0; k < J; ++k) {
            if (better(vmax[k], vidx[k], wmax[k], widx[k])) {
                wmax[k] = vmax[k];
                widx[k] = vidx[k];
            }
            vmax[k] = -__builtin_inf();
            vidx[k] = INT32_MAX;
        }
    };
    int prev = -1;                                                 // the brick whose row is still to be written
    for (int b = group; b < g.nbricks; b += a.ngroups) {
        if constexpr (SETS == 1) {
            if (prev >= 0) {
                __syncthreads();                                   // (the brick is done in every wavefront)
                brick_row(prev);
            }
            prev = b;
        }
        if (!s.sfit[b]) continue;                     // direct kernel's job
#ifdef QM_SHIFT_EXP_NOSTAGE                            // timing experiment (wrong results): the first
                                                      // brick's windows for all, no barriers
        if (b == group) {
            if constexpr (kWide) {
                stage_shift_wide<NW>(s, win, b, g.n_rows, wave, lane, t_first);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                stage_shift_windows<NW>(s, win, b, 0, g.n_rows, g.n_rows, wave, lane, t_first);
            }
            __syncthreads();
        }
#else
        __syncthreads();                              // previous brick fully consumed
        if constexpr (kWide) {
            stage_shift_wide<NW>(s, win, b, g.n_rows, wave, lane, t_first);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            stage_shift_windows<NW, (NW == 12 ? 3 : 8), 6, kTail, kWave * J>(s, win, b, 0, g.n_rows, g.n_rows, wave,
                                                                             lane, t_first);
        }
        __syncthreads();
#endif
        int x0, y0, z0, vx, vy, vz, cx, cy, cz;
        shift_group_box(g, b, x0, y0, z0, vx, vy, vz, cx, cy, cz);
        const int nvg = cx * cy * cz;
        const int mine = (nvg - wave + NW - 1) / NW;                       // groups of this wave
        if (mine <= 0) continue;
        const char *const stream = kWide ? s.wstream : s.stream;
        const char *run = stream + shift_run_record(b, wave, 0, NW, 1, rpw) * kShiftRec;
        // this wavefront's next run (its next brick that runs here; none: this one again, harmless):
        // the loop pulls its head into L2 (gen_shift_asm.py: NEXT_RUN)
        int nb = b + a.ngroups;
        while (nb < g.nbricks && !s.sfit[nb]) nb += a.ngroups;
        const char *next_run = stream + shift_run_record(nb < g.nbricks ? nb : b, wave, 0, NW, 1, rpw) * kShiftRec;
        const unsigned next_off = (unsigned)lane * 64u;
        const void *next_meta = s.smeta + (int64_t)(nb < g.nbricks ? nb : b) * g.n_rows;
        (void)next_meta;
        const unsigned lane_addr_b = lane_addr + (unsigned)kShiftPlane8;   // (8-wave shape: plane B)
#ifdef QM_SHIFT_DEPHASE          // (experiment: the second wavefront of every SIMD starts its rows late)
        if (NW == kShiftWaves8 && (wave & 4)) __builtin_amdgcn_s_sleep(QM_SHIFT_DEPHASE);
#endif
        (void)lane_addr_b; (void)next_run; (void)next_off;
#define QM_TAIL_CALL(JJ)                                                                              \
        if constexpr (MODE == kShiftMarginal)                                                         \
            shift_tail##JJ##_marginal(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz,       \
                                      a.z_scale, c, marg_tile, weight, node_off, lane_x16, lane_x32); \
        else if constexpr (MODE == kShiftVolume)                                                      \
            shift_tail##JJ##_volume(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz,         \
                                    a.z_scale, c, vol_tile, vol_stride_bytes, (unsigned)lane * (8u * JJ), \
                                    slot_lanes);                                                      \
        else                                                                                          \
            shift_tail##JJ##_detect(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz,         \
                                    a.z_scale, c)
        // (SETS = 2: the brick's row of this tile, such that row + the lane's LDS window address = its six samples)
        const char *const brow = reinterpret_cast<const char *>(a.brick_max + (int64_t)b * brow_stride + t_first) -
                                 lds_base;
        (void)brow;
        if constexpr (kWide && WIDE_LAZY && SETS == 2)
            shift_wide_detect_bmax_lazy(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr,
                                        brow, nz, nynz, a.z_scale, c);
        else if constexpr (kWide && SETS == 2)
            shift_wide_detect_bmax(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr,
                                   brow, nz, nynz, a.z_scale, c);
        else if constexpr (kWide && WIDE_LAZY)
            shift_wide_detect_lazy(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz,
                                   a.z_scale, c);
        else if constexpr (kWide)
            shift_wide_detect(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz,
                              a.z_scale, c);
        else if constexpr (J == 1) { QM_TAIL_CALL(1); }
        else if constexpr (J == 2) { QM_TAIL_CALL(2); }
        else if constexpr (J == 3) { QM_TAIL_CALL(3); }
#undef QM_TAIL_CALL
        else if constexpr (kLdsState)
            shift_groups_detect3(run, mine, next_run, next_off, next_meta, npairs, lane_addr, state_addr, nz, nynz, a.z_scale, c);
        else if constexpr (NW == kShiftWaves8 && MODE == kShiftMarginal)
            shift_groups_marginal8(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, lane_addr_b, nz, nynz,
                                   a.z_scale, c, marg_tile, weight, node_off, lane_x16, lane_x32);
        else if constexpr (MODE == kShiftMarginal)
            shift_groups_marginal(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz, a.z_scale, c,
                                  marg_tile, weight, node_off, lane_x16, lane_x32);
        else if constexpr (NW == kShiftWaves8 && MODE == kShiftVolume)
            shift_groups_volume8(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, lane_addr_b, nz, nynz,
                                 a.z_scale, c, vol_tile, vol_stride_bytes, (unsigned)lane * 32u, store_lanes);
        else if constexpr (NW == kShiftWaves8) {
            if (s.lazy && SETS != 1)                  // (a running maximum reset after every brick: eager)
                shift_groups_detect8_lazy(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, lane_addr_b, nz,
                                          nynz, a.z_scale, c);
            else
                shift_groups_detect8(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, lane_addr_b, nz, nynz,
                                     a.z_scale, c);
        }
        else if constexpr (MODE == kShiftVolume)
            shift_groups_volume(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz, a.z_scale, c,
                                vol_tile, vol_stride_bytes, (unsigned)lane * 32u, store_lanes);
        else if (s.lazy && SETS != 1)
            shift_groups_detect_lazy(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz, a.z_scale, c);
        else
            shift_groups_detect(vmax, vsum, vidx, run, mine, next_run, next_off, next_meta, npairs, lane_addr, nz, nynz, a.z_scale, c);
    }
    if (!a.want_scan) return;
    if constexpr (kLdsState) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vmax[k] = state[(k / 2) * (kShiftStateChunk / 8) + (k & 1)];
            vsum[k] = state[(2 + k / 2) * (kShiftStateChunk / 8) + (k & 1)];
            vidx[k] = reinterpret_cast<int *>(state + 4 * (kShiftStateChunk / 8))[k];
        }
    }
    // cross-wave combine through LDS: thread k of the workgroup owns sample k of the tile
    __syncthreads();
    if constexpr (SETS == 1) {
        brick_row(prev);                                           // the walk's last brick
#pragma unroll
        for (int k = 0; k < J; ++k) {
            vmax[k] = wmax[k];
            vidx[k] = widx[k];
        }
        __syncthreads();
    }
    shift_publish<NW, J>(a, win, vmax, vsum, vidx, wave, lane, group, t_first, work.t_own);
}

template <int MODE, int NW, bool SETS>           // (SETS: also a.brick_max, see shift_tile)
__device__ __forceinline__ void stack_shift_body(ShiftArgs &s, double *win) {
    static_assert(NW == kShiftWaves || NW == kShiftWaves8 || (NW == kShiftWaves3 && MODE == kShiftDetect),
                  "workgroup shapes: 4 or 8 waves, or 12 (detect only)");
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (MODE == kShiftDetect) s.a = step_view(s.a);   // (several timesteps per launch: this workgroup's)
    const ShiftWork work = shift_work(s.a);
    if (!work.run) return;
    if constexpr (MODE == kShiftDetect && NW == kShiftWaves8) {
        if (work.spl == kShiftWideSpl) {
            if constexpr (SETS) {
                if (s.lazy) return shift_tile<MODE, NW, kShiftWideSpl, true, 2>(s, win, work, lane, wave);
                return shift_tile<MODE, NW, kShiftWideSpl, false, 2>(s, win, work, lane, wave);
            }
            if (s.lazy) return shift_tile<MODE, NW, kShiftWideSpl, true, 0>(s, win, work, lane, wave);
            return shift_tile<MODE, NW, kShiftWideSpl, false, 0>(s, win, work, lane, wave);
        }
    }
    if constexpr (NW != kShiftWaves3) {
        if (work.spl == 3) return shift_tile<MODE, NW, 3, false, SETS ? 1 : 0>(s, win, work, lane, wave);
        if (work.spl == 2) return shift_tile<MODE, NW, 2, false, SETS ? 1 : 0>(s, win, work, lane, wave);
        if (work.spl == 1) return shift_tile<MODE, NW, 1, false, SETS ? 1 : 0>(s, win, work, lane, wave);
    }
    shift_tile<MODE, NW, 4, false, SETS ? 1 : 0>(s, win, work, lane, wave);
}

template <int MODE, int NW>
__global__ __launch_bounds__(NW * kWave, NW == kShiftWaves3 ? 3 : 2) void stack_shift_kernel(ShiftArgs s) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    stack_shift_body<MODE, NW, false>(s, win);
}
// the fused detect that also leaves a row of maxima per BRICK (StackArgs::brick_max; tie_rule = 1, qm_ties.hpp)
template <int NW>
__global__ __launch_bounds__(NW * kWave, 2) void stack_shift_bricks_kernel(ShiftArgs s) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    stack_shift_body<kShiftDetect, NW, true>(s, win);
}

// Tables of more rows than a CU's LDS holds windows for (> 64): ROW BLOCKS.  A brick is as many
// 2x2x2 groups as the workgroup has wavefronts (4x4x4 nodes for 8), each wavefront owns ONE group
// and keeps its 64 accumulators in registers while the workgroup stages the brick's rows block by
// block (<= 64 rows each): rows are still added one at a time in ascending order, so every sum has
// the reference's bits.  The accumulators live in hard registers of the generated loop
// (shift_group_rows8: v80 up) ACROSS its calls -- the compiler does not know, so its own code
// between two calls must stay below kShiftBlockVgprs: see the kernel's attributes; the staging is
// kept lean for that (half the rows in flight per pass) and tests/test_host.py checks the ISA.
// (amdgpu_waves_per_eu(6, 6) is what keeps the compiler below v80 = kShiftBlockVgprs: it plans for
// six wavefronts per SIMD, i.e. 80 registers, and treats the rest as reserved; the generated loop's
// hard registers above bring the kernel to 248, two wavefronts per SIMD)
// Row blocks: the metadata (row-window records, 16 bytes per row) of the block whose staging FOLLOWS the
// next one's -- two steps ahead in the workgroup's sequence (brick b: blocks 0 .. nblk-1, then its next
// brick nb) --, for the generated loop's prefetch.  A hint: past the end it points at this brick's own.
__device__ __forceinline__ const void *shift_meta_ahead(const ShiftArgs &s, const GridDesc &g, int b, int nb,
                                                        int k, const int4 *smeta = nullptr) {
    if (smeta == nullptr) smeta = s.smeta;
    int pb = b, pk = k + 2;
    if (pk >= s.nblk) {
        pb = nb;
        pk -= s.nblk;
        if (pk >= s.nblk) {                       // (a single block per brick: the brick after the next)
            pb = nb + s.a.ngroups;
            pk = 0;
        }
    }
    if (pb >= g.nbricks) pb = b, pk = 0;
    return smeta + ((int64_t)pb * s.nblk + pk) * s.sb;
}

#ifndef QM_ROWS_RB               // rows in flight per wavefront and staging pass, 64-sample chunks per row
#define QM_ROWS_RB 3             // (3 x 5: the most that leaves the compiler without spills below v80)
#define QM_ROWS_U 5
#endif
template <int NW>
__global__ __attribute__((amdgpu_flat_work_group_size(NW * kWave, NW * kWave), amdgpu_waves_per_eu(6, 6)))
void stack_shift_rows_kernel(ShiftArgs s) {
    static_assert(NW == kShiftWaves8, "row blocks: the 8-wave workgroup");
    extern __shared__ __attribute__((aligned(16))) double win[];
    const StackArgs &a = s.a;
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const ShiftWork work = shift_work(a);
    if (!work.run) return;
    const int tile = work.tile, group = work.group, t_first = work.t_first;
    const unsigned lane_addr = (unsigned)(uintptr_t)((lds_f64 *)win) + (unsigned)lane * 16u;

    double vmax[4], vsum[4];
    int vidx[4];
    shift_reset(vmax, vsum, vidx);
    constexpr int D = Exp2Degree<false>::value;
    double c[D + 1];
#pragma unroll
    for (int i = 0; i <= D; ++i) c[i] = exp2_coeff<D>(i);

    const int rows2max = s.sb + (s.sb & 1);
    const int64_t rpw = shift_recs_per_wave(g, rows2max, NW);
    for (int b = group; b < g.nbricks; b += a.ngroups) {
        if (!s.sfit[b]) continue;                     // direct kernel's job
        int x0, y0, z0, vx, vy, vz, cx, cy, cz;
        shift_group_box(g, b, x0, y0, z0, vx, vy, vz, cx, cy, cz);
        const bool mine = wave < cx * cy * cz;        // one group per wavefront
        // this wavefront's next run: the first block of its next brick
        int nb = b + a.ngroups;
        while (nb < g.nbricks && !s.sfit[nb]) nb += a.ngroups;
        const char *next_run = s.stream + shift_run_record(nb < g.nbricks ? nb : b, wave, 0, NW, s.nblk, rpw) * kShiftRecBlocks;
        for (int k = 0; k < s.nblk; ++k) {
            const void *next_meta = shift_meta_ahead(s, g, b, nb, k);
            const int row0 = k * s.sb;
            const int rows = g.n_rows - row0 < s.sb ? g.n_rows - row0 : s.sb;
            __syncthreads();                          // previous block fully consumed
            stage_shift_windows<NW, QM_ROWS_RB, QM_ROWS_U>(s, win, b * s.nblk + k, row0, rows, s.sb, wave,
                                                           lane, t_first);
            __syncthreads();
            if (mine) {
                const char *run = s.stream + shift_run_record(b, wave, k, NW, s.nblk, rpw) * kShiftRecBlocks;
                const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)((k == 0 ? 1u : 0u) | (k == s.nblk - 1 ? 2u : 0u)));
                shift_group_rows8(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                  (rows + 1) / 2, lane_addr, lane_addr + (unsigned)kShiftPlane8, g.nz,
                                  g.ny * g.nz, a.z_scale, c);
            }
        }
    }
    if (!a.want_scan) return;
    // cross-wave combine through LDS: thread k of the workgroup owns sample k of the tile
    __syncthreads();
    shift_publish<NW>(a, win, vmax, vsum, vidx, wave, lane, group, t_first);
}

// Row blocks, second form: the CU's LDS holds TWO halves of 80 KB (each laid out like the 4-wave
// shape: plane B kShiftPlane bytes above plane A), the rows are staged in blocks of <= 34, and the
// NEXT block is written into the idle half while the wavefronts add the current one -- by
// LDS-direct loads (global_load_lds_dwordx4, gfx950: 16 bytes per lane from global memory straight
// into LDS at M0 + 16 lane, no registers, counted by vmcnt).  A 16-byte slot of plane A is window
// samples (4s, 4s+1), of plane B (4s+2, 4s+3): one instruction per plane and 64 slots.  One
// barrier per block.  (Window slots past the row's end are left as they are: the loop fetches them
// with the quads but never adds them.)

// the all-zero window of the padding row of a block with an odd row count
__device__ __forceinline__ void stage_shift_zero_row(const ShiftArgs &s, double *half, int vb, int S, int wave,
                                                     int lane) {
    if ((S & 1) && wave == 0) {
        const int z = s.stotal[vb];
        for (int u = lane; u < 4 * (64 + kShiftNqMin); u += kWave)
            half[((u & 2) ? kShiftPlane / 8 : 0) + 2 * (z + (u >> 2)) + (u & 1)] = 0.0;
    }
}

template <int NW>
__device__ __forceinline__ void stage_shift_block_direct(const ShiftArgs &s, double *half, int vb,
                                                         int row0, int S, int wave, int lane,
                                                         int t_first) {
    const StackArgs &a = s.a;
    using lds_ptr = __attribute__((address_space(3))) void *;
    using glb_ptr = const __attribute__((address_space(1))) void *;
    // (the rows' metadata first, all loads in flight at once: a wavefront stages up to five rows of
    // a block, and five dependent round trips to L2 were as long as the block's adds -- PMC of the
    // first version: VALU busy 44 %, profiles/r03_pmc_rows128_*)
    // They come by SCALAR loads (their own counter: a vector load's wait would also wait for the
    // stream prefetches the loop of the block before has left in flight).
    constexpr int R = (34 + NW - 1) / NW;
    using int4s = int __attribute__((ext_vector_type(4)));
    int4s ms[R];
    if constexpr (NW == kShiftWaves8) {
        static_assert(R == 5, "five metadata loads per wavefront and block");
        const int4 *base = s.smeta + (int64_t)vb * s.sb;
        unsigned off[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = wave + j * NW;
            off[j] = (unsigned)__builtin_amdgcn_readfirstlane((r < S ? r : 0) * 16);
        }
        asm volatile("s_load_dwordx4 %0, %5, %6\n\ts_load_dwordx4 %1, %5, %7\n\ts_load_dwordx4 %2, %5, %8\n\t"
                     "s_load_dwordx4 %3, %5, %9\n\ts_load_dwordx4 %4, %5, %10\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ms[0]), "=&s"(ms[1]), "=&s"(ms[2]), "=&s"(ms[3]), "=&s"(ms[4])
                     : "s"(base), "s"(off[0]), "s"(off[1]), "s"(off[2]), "s"(off[3]), "s"(off[4])
                     : "memory");
    } else {
        // (the 4-wave form: nine rows per wavefront; the wave-uniform index makes these scalar loads)
        const int4s *base = reinterpret_cast<const int4s *>(s.smeta + (int64_t)vb * s.sb);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = wave + j * NW;
            ms[j] = base[__builtin_amdgcn_readfirstlane(r < S ? r : 0)];
        }
    }
    int4 meta[R];
#pragma unroll
    for (int j = 0; j < R; ++j) meta[j] = make_int4(ms[j].x, ms[j].y, ms[j].z, ms[j].w);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int r = wave + j * NW;
        if (r >= S) break;
        const int4 m = meta[j];
        const int first = m.x + a.fsmp + a.sample0 + t_first;      // index inside the row
        const int room = a.T - first;
        const double *src = a.onsets + (int64_t)(row0 + r) * a.T + first;
#ifndef QM_ROWS_STAGE_SLOW
        if (room >= 4 * m.w) {
            // Every slot of the window lies inside the row (all rows but those that reach the onsets' last
            // samples): everything about the loads is wave-uniform except the lane's 32 bytes, so the pair
            // of loads per 64 slots is issued with scalar bookkeeping only -- EXEC = the chunk's slots, M0 =
            // the plane's LDS address, the row pointer as the instruction's scalar base (the instruction
            // offset moves the global AND the LDS address: plane B's M0 is 16 short).  Round 5: the
            // compiler's per-lane form of this loop (the code below) was 18 % of a 128-row step.
            // (wave-uniform by construction; said again for the 4-wave form, whose metadata arrive in VGPRs)
            const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)((unsigned)(uintptr_t)((lds_f64 *)half) + 16u * (unsigned)m.z));
            const unsigned long long sp = (unsigned long long)src;
            src = (const double *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sp >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)sp));
            const int slots = __builtin_amdgcn_readfirstlane(m.w);
            const unsigned lane32 = (unsigned)lane * 32u;
            for (int c = 0; c < slots; c += kWave) {
                const int n = slots - c < kWave ? slots - c : kWave;
                unsigned long long saved;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_lshr_b64 exec, -1, %[sh]\n\t"
                             "s_mov_b32 m0, %[ma]\n\t"
                             "s_nop 0\n\t"
                             "global_load_lds_dwordx4 %[vo], %[src]\n\t"
                             "s_mov_b32 m0, %[mb]\n\t"
                             "s_nop 0\n\t"
                             "global_load_lds_dwordx4 %[vo], %[src] offset:16\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved)
                             : [sh] "s"(kWave - n), [ma] "s"(la + 16u * (unsigned)c),
                               [mb] "s"(la + 16u * (unsigned)c + (unsigned)kShiftPlane - 16u), [vo] "v"(lane32),
                               [src] "s"(src + 4 * c)
                             : "memory", "m0");
            }
            continue;
        }
#endif
        for (int c = 0; c * kWave < m.w; ++c) {
            const int slot = c * kWave + lane;
            if (slot < m.w) {
                // samples of the slot's pair that lie inside the row: 2 -> one 16-byte load; 1 (the
                // row's last sample, e.g. the scan's last sample at the largest delay) -> that
                // sample alone, through a register; 0 -> nothing (never added, whatever is there)
                const int u = 4 * slot;
                double *la = half + 2 * (m.z + c * kWave);         // (+ 16 bytes x lane by the hardware)
                double *lb = la + kShiftPlane / 8;
                if (room - u >= 2)
                    __builtin_amdgcn_global_load_lds((glb_ptr)(src + u), (lds_ptr)(lds_f64 *)la, 16, 0, 0);
                else if (room - u == 1)
                    la[2 * lane] = src[u];
                if (room - u >= 4)
                    __builtin_amdgcn_global_load_lds((glb_ptr)(src + u + 2), (lds_ptr)(lds_f64 *)lb, 16, 0, 0);
                else if (room - u == 3)
                    lb[2 * lane] = src[u + 2];
            }
        }
    }
    stage_shift_zero_row(s, half, vb, S, wave, lane);
}

template <bool VOLUME, int NW>
__global__ __attribute__((amdgpu_flat_work_group_size(NW * kWave, NW * kWave), amdgpu_waves_per_eu(6, 6)))
void stack_shift_rows2_kernel(ShiftArgs s) {
    static_assert(NW == kShiftWaves8, "row blocks: the 8-wave workgroup");
    extern __shared__ __attribute__((aligned(16))) double win[];
    const StackArgs &a = s.a;
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const ShiftWork work = shift_work(a);
    if (!work.run) return;
    const int tile = work.tile, group = work.group, t_first = work.t_first;
    const unsigned lane_addr = (unsigned)(uintptr_t)((lds_f64 *)win) + (unsigned)lane * 16u;
    // volume: lanes of a pulled-back tile whose four samples its predecessor stores are masked off
    const int seam = tile * kShiftKT - t_first;
    const unsigned long long store_lanes = ~0ull << (seam / 4);

    double vmax[4], vsum[4];
    int vidx[4];
    shift_reset(vmax, vsum, vidx);
    constexpr int D = Exp2Degree<VOLUME>::value;
    double c[D + 1];
#pragma unroll
    for (int i = 0; i <= D; ++i) c[i] = exp2_coeff<D>(i);

    const int rows2max = s.sb + (s.sb & 1);
    const int64_t rpw = shift_recs_per_wave(g, rows2max, NW);
    auto rows_of = [&](int k) { return g.n_rows - k * s.sb < s.sb ? g.n_rows - k * s.sb : s.sb; };
    // The row loop stages the next block itself (gen_shift_asm.py, stage_step) where its scalar-only form
    // holds: every window of the table inside the onsets' rows for this tile, at most 127 slots each.
    const int tile_room = a.T - (a.fsmp + a.sample0 + t_first);
    const bool in_loop = kShiftStageInLoop && s.stage_slots <= 127 && tile_room >= s.stage_reach &&
                         a.T < (1 << 28);
    int b = group;
    while (b < g.nbricks && !s.sfit[b]) b += a.ngroups;            // (others: the direct kernel's job)
    int cur = 0;                                                   // half the current block lies in
    if (b < g.nbricks)
        stage_shift_block_direct<NW>(s, win, b * s.nblk, 0, rows_of(0), wave, lane, t_first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    while (b < g.nbricks) {
        int nb = b + a.ngroups;
        while (nb < g.nbricks && !s.sfit[nb]) nb += a.ngroups;
        int x0, y0, z0, vx, vy, vz, cx, cy, cz;
        shift_group_box(g, b, x0, y0, z0, vx, vy, vz, cx, cy, cz);
        const bool mine = wave < cx * cy * cz;                     // one group per wavefront
        const char *next_run =
            s.stream + shift_run_record(nb < g.nbricks ? nb : b, wave, 0, NW, s.nblk, rpw) * kShiftRecBlocks;
        for (int k = 0; k < s.nblk; ++k) {
            const void *next_meta = shift_meta_ahead(s, g, b, nb, k);
            // the next block (of this brick, or the first of the next) into the idle half
            double *idle = win + (cur ^ 1) * (kShiftHalfBytes / 8);
            const bool follows = k + 1 < s.nblk || nb < g.nbricks;
            const int vbn = k + 1 < s.nblk ? b * s.nblk + k + 1 : nb * s.nblk;
            const int kn = k + 1 < s.nblk ? k + 1 : 0;
            ShiftStageNext stage{};
            stage.meta = s.smeta;
#ifndef QM_ROWS_EXP_NOSTAGE      // (timing experiment: wrong results)
            if (follows && in_loop && mine) {
                stage_shift_zero_row(s, idle, vbn, rows_of(kn), wave, lane);
                stage.meta = s.smeta + (int64_t)vbn * s.sb;
                stage.rows = rows_of(kn);
                stage.first_row = wave;
                stage.stride = NW;
                stage.src = a.onsets + (int64_t)kn * s.sb * a.T + a.fsmp + a.sample0 + t_first;
                stage.row_bytes = (unsigned)a.T * 8u;
                stage.lds = (unsigned)(uintptr_t)((lds_f64 *)idle);
                stage.lane32 = (unsigned)lane * 32u;
            } else if (follows) {
                stage_shift_block_direct<NW>(s, idle, vbn, kn * s.sb, rows_of(kn), wave, lane, t_first);
            }
#endif
            if (mine) {
                const char *run = s.stream + shift_run_record(b, wave, k, NW, s.nblk, rpw) * kShiftRecBlocks;
                const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)((k == 0 ? 1u : 0u) | (k == s.nblk - 1 ? 2u : 0u)));
                if constexpr (VOLUME)
                    shift_group_rows_volume(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                            (rows_of(k) + 1) / 2, lane_addr + (unsigned)(cur * kShiftHalfBytes),
                                            stage, g.nz, g.ny * g.nz, a.z_scale, c, a.volume + t_first,
                                            (unsigned)(a.vol_stride * 8), (unsigned)lane * 32u, store_lanes);
                else if (s.lazy)
                    shift_group_rows_lazy(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                          (rows_of(k) + 1) / 2, lane_addr + (unsigned)(cur * kShiftHalfBytes),
                                          stage, g.nz, g.ny * g.nz, a.z_scale, c);
                else
                    shift_group_rows(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                     (rows_of(k) + 1) / 2, lane_addr + (unsigned)(cur * kShiftHalfBytes), stage,
                                     g.nz, g.ny * g.nz, a.z_scale, c);
            }
            // this wavefront's staging loads are in LDS (the loop has waited for them, leaving only
            // its own stream prefetches in flight; a wavefront without a group waits here), then
            // everyone's; the current half is free
            if (!mine || VOLUME) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef QM_ROWS_EXP_NOBARRIER    // (timing experiment: wrong results)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            cur ^= 1;
        }
        b = nb;
    }
    if (!a.want_scan) return;
    // cross-wave combine through LDS: thread k of the workgroup owns sample k of the tile
    shift_publish<NW>(a, win, vmax, vsum, vidx, wave, lane, group, t_first);
}
// Row blocks on WIDE tiles (round 6): the double-buffered form above with six samples per lane -- for tables
// whose 384-sample row windows do not fit a CU's LDS all at once (from ~36 rows of C3's geometry on; BASELINE
// configs[3]: 60 rows).  Bricks of 4x4x4 nodes, one 2x2x2 group per wavefront with its 48 accumulators (96
// VGPRs) in the generated loop's hard registers across the calls; blocks of <= 20 rows in halves of 80 KB,
// contiguous row windows, the next block staged by the row loop itself (one LDS address per row, four loads:
// gen_shift_asm.py, stage_step) or -- where a window reaches past the onsets' rows -- by stage_shift_wide.
// The compiler's own code between two calls stays below kShiftWideBlockVgprs (amdgpu_waves_per_eu(9, 9): 56
// registers; csrc/check_shift_isa.py walks the ISA at every build).  Fused detect only; whole wide tiles, the
// last one pulled back over its predecessor.
// (LAZY: the loop flavour, chosen outside the block loop -- both flavours' asm statements in one loop make the
// compiler shuffle and spill the running state around every call, see shift_tile)
template <bool LAZY>
__device__ __forceinline__ void shift_wide_rows_body(const ShiftArgs &s, double *win) {
    constexpr int NW = kShiftWaves8, J = kShiftWideSpl;
    const StackArgs &a = s.a;
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const ShiftWork work = shift_work(a);
    if (!work.run) return;
    const int group = work.group, t_first = work.t_first;
    const unsigned lane_addr = (unsigned)(uintptr_t)((lds_f64 *)win) + (unsigned)lane * (8u * J);

    double vmax[J], vsum[J];
    int vidx[J];
    shift_reset<J>(vmax, vsum, vidx);
    constexpr int D = Exp2Degree<false>::value;
    double c[D + 1];
#pragma unroll
    for (int i = 0; i <= D; ++i) c[i] = exp2_coeff<D>(i);

    const int rows2max = s.sb + (s.sb & 1);
    const int64_t rpw = shift_recs_per_wave(g, rows2max, NW);
    auto rows_of = [&](int k) { return g.n_rows - k * s.sb < s.sb ? g.n_rows - k * s.sb : s.sb; };
    const int tile_room = a.T - (a.fsmp + a.sample0 + t_first);
    const bool in_loop = s.stage_slots <= 127 && tile_room >= s.stage_reach && a.T < (1 << 28);
    int b = group;
    while (b < g.nbricks && !s.sfit[b]) b += a.ngroups;            // (others: the direct kernel's job)
    int cur = 0;                                                   // half the current block lies in
    if (b < g.nbricks) stage_shift_wide<NW>(s, win, b * s.nblk, rows_of(0), wave, lane, t_first, 0, s.sb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    while (b < g.nbricks) {
        int nb = b + a.ngroups;
        while (nb < g.nbricks && !s.sfit[nb]) nb += a.ngroups;
        int x0, y0, z0, vx, vy, vz, cx, cy, cz;
        shift_group_box(g, b, x0, y0, z0, vx, vy, vz, cx, cy, cz);
        const bool mine = wave < cx * cy * cz;                     // one group per wavefront
        const char *next_run =
            s.wstream + shift_run_record(nb < g.nbricks ? nb : b, wave, 0, NW, s.nblk, rpw) * kShiftRecBlocks;
        for (int k = 0; k < s.nblk; ++k) {
            const void *next_meta = shift_meta_ahead(s, g, b, nb, k, s.wmeta);
            double *idle = win + (cur ^ 1) * (kShiftHalfBytes / 8);
            const bool follows = k + 1 < s.nblk || nb < g.nbricks;
            const int vbn = k + 1 < s.nblk ? b * s.nblk + k + 1 : nb * s.nblk;
            const int kn = k + 1 < s.nblk ? k + 1 : 0;
            ShiftStageNext stage{};
            stage.meta = s.wmeta;
            if (follows && in_loop && mine) {
                if ((rows_of(kn) & 1) && wave == 0) {              // the padding row's zero window
                    const int z4 = 4 * s.wtotal[vbn];
                    for (int u = lane; u < 4 * shift_zero_slots(1); u += kWave) idle[z4 + u] = 0.0;
                }
                stage.meta = s.wmeta + (int64_t)vbn * s.sb;
                stage.rows = rows_of(kn);
                stage.first_row = wave;
                stage.stride = NW;
                stage.src = a.onsets + (int64_t)kn * s.sb * a.T + a.fsmp + a.sample0 + t_first;
                stage.row_bytes = (unsigned)a.T * 8u;
                stage.lds = (unsigned)(uintptr_t)((lds_f64 *)idle);
                stage.lane32 = (unsigned)lane * 16u;               // (16 bytes per lane and load)
            } else if (follows) {
                stage_shift_wide<NW>(s, idle, vbn, rows_of(kn), wave, lane, t_first, kn * s.sb, s.sb);
            }
            if (mine) {
                const char *run = s.wstream + shift_run_record(b, wave, k, NW, s.nblk, rpw) * kShiftRecBlocks;
                const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)((k == 0 ? 1u : 0u) | (k == s.nblk - 1 ? 2u : 0u)));
                if constexpr (LAZY)
                    shift_wide_rows_lazy(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                         (rows_of(k) + 1) / 2, lane_addr + (unsigned)(cur * kShiftHalfBytes), stage,
                                         g.nz, g.ny * g.nz, a.z_scale, c);
                else
                    shift_wide_rows(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                    (rows_of(k) + 1) / 2, lane_addr + (unsigned)(cur * kShiftHalfBytes), stage,
                                    g.nz, g.ny * g.nz, a.z_scale, c);
            }
            if (!mine) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
        }
        b = nb;
    }
    if (!a.want_scan) return;
    // cross-wave combine through LDS: thread k of the workgroup owns sample k of the tile
    shift_publish<NW, J>(a, win, vmax, vsum, vidx, wave, lane, group, t_first);
}

#if QM_SHIFT_TU == 1                                    // (the one kernel here that is no template: one unit's)
__global__ __attribute__((amdgpu_flat_work_group_size(kShiftWaves8 * kWave, kShiftWaves8 * kWave),
                          amdgpu_waves_per_eu(9, 9)))
void stack_shift_wide_rows_kernel(ShiftArgs s) {
    extern __shared__ __attribute__((aligned(16))) double win[];
    if (s.lazy) shift_wide_rows_body<true>(s, win);
    else shift_wide_rows_body<false>(s, win);
}
#endif

// Row blocks, third form (round 4): TWO 4-wave workgroups per CU, 80 KB each -- the shape of the
// tables of up to ~32 rows, for the same reason: the two wavefronts of a SIMD then belong to
// different workgroups and are in different phases, so one's block boundary (the barrier, the
// generated loop's lead-in: two records and the first window from L2) is covered by the other's adds.
// In the 8-wave forms above both wavefronts of every SIMD enter every block together and the
// lead-in is exposed on all SIMDs at once (PMC: VALU busy 48 %).  A brick is 4x4x2 nodes = one
// 2x2x2 group per wavefront; a block of <= 34 rows is staged by LDS-direct loads into the
// workgroup's single buffer between two barriers (no double buffering: the other workgroup is what
// runs meanwhile).  Same generated loop, same accumulators in its hard registers across the calls,
// same bits.
template <bool VOLUME>
__global__ __attribute__((amdgpu_flat_work_group_size(kShiftWaves * kWave, kShiftWaves * kWave),
                          amdgpu_waves_per_eu(6, 6)))
void stack_shift_rows4_kernel(ShiftArgs s) {
    constexpr int NW = kShiftWaves;
    extern __shared__ __attribute__((aligned(16))) double win[];
    const StackArgs &a = s.a;
    const GridDesc &g = a.g;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const ShiftWork work = shift_work(a);
    if (!work.run) return;
    const int tile = work.tile, group = work.group, t_first = work.t_first;
    const unsigned lane_addr = (unsigned)(uintptr_t)((lds_f64 *)win) + (unsigned)lane * 16u;
    // volume: lanes of a pulled-back tile whose four samples its predecessor stores are masked off
    const int seam = tile * kShiftKT - t_first;
    const unsigned long long store_lanes = ~0ull << (seam / 4);
    (void)store_lanes;

    double vmax[4], vsum[4];
    int vidx[4];
    shift_reset(vmax, vsum, vidx);
    constexpr int D = Exp2Degree<VOLUME>::value;
    double c[D + 1];
#pragma unroll
    for (int i = 0; i <= D; ++i) c[i] = exp2_coeff<D>(i);

    const int rows2max = s.sb + (s.sb & 1);
    const int64_t rpw = shift_recs_per_wave(g, rows2max, NW);
    auto rows_of = [&](int k) { return g.n_rows - k * s.sb < s.sb ? g.n_rows - k * s.sb : s.sb; };
    int b = group;
    while (b < g.nbricks && !s.sfit[b]) b += a.ngroups;            // (others: the direct kernel's job)
    while (b < g.nbricks) {
        int nb = b + a.ngroups;
        while (nb < g.nbricks && !s.sfit[nb]) nb += a.ngroups;
        int x0, y0, z0, vx, vy, vz, cx, cy, cz;
        shift_group_box(g, b, x0, y0, z0, vx, vy, vz, cx, cy, cz);
        const bool mine = wave < cx * cy * cz;                     // one group per wavefront
        const char *next_run =
            s.stream + shift_run_record(nb < g.nbricks ? nb : b, wave, 0, NW, s.nblk, rpw) * kShiftRecBlocks;
        for (int k = 0; k < s.nblk; ++k) {
            const void *next_meta = shift_meta_ahead(s, g, b, nb, k);
            // everyone is done with the block in LDS; this block's rows in, by LDS-direct loads
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stage_shift_block_direct<NW>(s, win, b * s.nblk + k, k * s.sb, rows_of(k), wave, lane, t_first);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (mine) {
                ShiftStageNext no_stage{};               // (one buffer: nothing for the loop to stage ahead)
                no_stage.meta = s.smeta;
                const char *run = s.stream + shift_run_record(b, wave, k, NW, s.nblk, rpw) * kShiftRecBlocks;
                const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)((k == 0 ? 1u : 0u) | (k == s.nblk - 1 ? 2u : 0u)));
                if constexpr (VOLUME)
                    shift_group_rows_volume(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                            (rows_of(k) + 1) / 2, lane_addr, no_stage, g.nz, g.ny * g.nz, a.z_scale, c,
                                            a.volume + t_first, (unsigned)(a.vol_stride * 8),
                                            (unsigned)lane * 32u, store_lanes);
                else if (s.lazy)
                    shift_group_rows_lazy(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                          (rows_of(k) + 1) / 2, lane_addr, no_stage, g.nz, g.ny * g.nz, a.z_scale, c);
                else
                    shift_group_rows(vmax, vsum, vidx, run, flags, next_run, (unsigned)lane * 64u, next_meta,
                                     (rows_of(k) + 1) / 2, lane_addr, no_stage, g.nz, g.ny * g.nz, a.z_scale, c);
            }
        }
        b = nb;
    }
    if (!a.want_scan) return;
    // cross-wave combine through LDS: thread k of the workgroup owns sample k of the tile
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    shift_publish<NW>(a, win, vmax, vsum, vidx, wave, lane, group, t_first);
}
#endif  // QM_SHIFT_TU

}  // namespace qm
