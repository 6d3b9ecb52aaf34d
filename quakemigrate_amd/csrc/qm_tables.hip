// qm_tables.hip -- everything the engine derives from ONE travel-time table (TableState,
// qm_engine.hpp): making it resident (qm_engine_load_lut: brick records, the layout search), the
// tables the stacking kernels build from it on first use (round-2 window offsets, the paired layout,
// the shift-reuse layout with its record stream, the screening sweep's offsets), tables parked under a
// key while another availability's table is worked on (qm_engine_table_select), and on-device serving
// of the int32 table from float64 travel-time grids (qm_engine_serve; lut.py:502-538, :102-140).
#define QM_TU_TABLES 1
#include "qm_engine.hpp"

int lds_cap_doubles(const qm_engine *e) { return e->cfg_lds_bytes / 8; }

// 16-bit window offsets of the round-2 stacking kernels (brick_rel_kernel), on first use per table
int ensure_rel(qm_engine *e) {
    if (e->rel_built) return 0;
    const qm::GridDesc &g = e->g;
    if (e->d_rel.ensure((size_t)g.nbricks * g.brick_nodes * g.row_pad)) return 1;
    hipLaunchKernelGGL(qm::brick_rel_kernel, dim3(g.nbricks), dim3(256), 0, e->stream, g,
                       e->d_lut.p, reinterpret_cast<const int4 *>(e->d_bmeta.p), e->d_btotal.p,
                       e->d_rel.p);
    QM_HIP(hipGetLastError());
    e->rel_built = true;
    return 0;
}

// Samples per lane: explicit, the table's layout search's choice (load_lut), or the largest J
// whose S row windows leave >= 20 % of the LDS budget for the delay spans (J = 4 up to 64 rows:
// beyond 40 its pipelined kernels spill a few offset chunks per node, and only the exact-row-count
// kernels are built for that).
int eff_j(const qm_engine *e) {
    const int S = e->n_rows_hint > 0 ? e->n_rows_hint : 1;
    if (e->cfg_j > 0) return (e->cfg_j == 4 && S > qm::kJ4MaxRows) ? 2 : e->cfg_j;   // see below
    if (e->auto_j > 0) return e->auto_j;
    for (int j : {4, 2, 1}) {
        if (j == 4 && S > qm::kJ4MaxRows) continue;
        if ((int64_t)S * qm::kWave * j * 8 * 5 <= (int64_t)e->cfg_lds_bytes * 4) return j;
    }
    return 1;
}

// Samples per lane for one launch over n_chunk samples: never more than eff_j (the brick shape
// was chosen for it), but fewer when the padding of the last time tile costs more than the
// smaller tile's overhead (measured on C3/C4: J = 2 is ~1.12x, J = 1 ~1.4x the work of J = 4 per
// sample) -- e.g. the Icequake example's 625-sample timestep runs 5 tiles of 128, not 3 of 256.
int run_j(const qm_engine *e, int n_chunk) {
    const int jmax = eff_j(e);
    if (e->cfg_j > 0) return jmax;
    int best = jmax;
    double best_cost = 1e300;
    for (int j : {4, 2, 1}) {
        if (j > jmax) continue;
        const int kt = qm::kWave * j;
        const double cost = (double)((n_chunk + kt - 1) / kt) * kt * (j == 4 ? 1.0 : j == 2 ? 1.12 : 1.4);
        if (cost < best_cost * 0.999) { best_cost = cost; best = j; }
    }
    return best;
}

// bricks whose windows do not fit the LDS budget for tile length 64*J
int plan_wide(qm_engine *e, int J) {
    const int KT = qm::kWave * J;
    const int cap = lds_cap_doubles(e);
    if (e->plan_j == J && e->plan_cap == cap) return 0;
    std::vector<int32_t> wide;
    for (int b = 0; b < e->g.nbricks; ++b) {
        if (!qm::brick_fits(e->h_btotal[b], e->g.n_rows, KT, cap)) wide.push_back(b);
    }
    e->n_wide = (int)wide.size();
    if (e->n_wide) {
        if (e->d_wide.ensure(wide.size())) return 1;
        QM_HIP(copy_in(e->d_wide.p, wide.data(), wide.size() * sizeof(int32_t), e->stream));
        QM_HIP(hipStreamSynchronize(e->stream));
    }
    e->plan_j = J;
    e->plan_cap = cap;
    return 0;
}

// ---- paired (16-byte operand) layout (qm_pair.hpp; row counts and constants: qm_launch.hpp) ----
using qm::kPairLdsBytes;
using qm::pair_jp_of;

// pairs per lane for a launch over n_chunk samples; 0 = the chunked / exact kernels run.
// pair = 1 (default): the volume-writing launches only -- there the paired layout pays (one
// 16-byte store per pair, the stores under the next node's LDS stream: C3 locate window 6.8 ->
// 6.4 ms); the fused detect gains nothing from it (the LDS array moves the same bytes and is
// ~80 % busy either way, profiles/r02_pmc_*) and keeps the two-workgroups-per-CU b64 kernel.
int pair_jp(const qm_engine *e, int n_chunk, bool volume) {
    if (!e->cfg_pair || e->cfg_generic || e->cfg_force_direct || e->user_waves || e->user_lds ||
        e->cfg_j > 0)
        return 0;
    const int jp = pair_jp_of(e->g.n_rows);
    if (e->cfg_pair == 2) return jp;                   // forced (tests): detect too, any scan length
    if (!volume) return 0;
    // short scans run on shorter tiles (run_j): leave those to the chunked kernels
    return (jp > 0 && run_j(e, n_chunk) == eff_j(e) && qm::kWave * eff_j(e) >= 128 * jp) ? jp : 0;
}

// Own brick grid (e->pg): the largest brick shape whose two staggered window copies fit 160 KB
// for (almost) every brick; per-brick (min, span2, prefix) records and the 16-bit offset table.
int ensure_pair_tables(qm_engine *e, int jp) {
    const int KT = 128 * jp;
    if (e->pair_kt == KT) return 0;
    static const int kShapes[][3] = {{8, 8, 8}, {4, 8, 8}, {4, 4, 8}, {4, 4, 4},
                                     {2, 4, 4}, {2, 2, 4}, {2, 2, 2}, {1, 1, 2}, {1, 1, 1}};
    const bool fixed = e->cfg_bx > 0;
    const int n_shapes = fixed ? 1 : (int)(sizeof(kShapes) / sizeof(kShapes[0]));
    qm::GridDesc g = e->g;
    std::vector<int32_t> total, wide;
    for (int s = 0; s < n_shapes; ++s) {
        g = e->g;
        if (!fixed) {
            g.bx = std::min(kShapes[s][0], g.nx);
            g.by = std::min(kShapes[s][1], g.ny);
            g.bz = std::min(kShapes[s][2], g.nz);
            g.nbx = (g.nx + g.bx - 1) / g.bx;
            g.nby = (g.ny + g.by - 1) / g.by;
            g.nbz = (g.nz + g.bz - 1) / g.bz;
            g.nbricks = g.nbx * g.nby * g.nbz;
            g.brick_nodes = g.bx * g.by * g.bz;
        }
        const size_t br = (size_t)g.nbricks * g.n_rows;
        if (e->d_pmeta_raw.ensure(4 * br) || e->d_pmeta.ensure(4 * br) ||
            e->d_ptotal.ensure(g.nbricks) || e->d_scalar.ensure(4))
            return 1;
        QM_HIP(hipMemsetAsync(e->d_scalar.p, 0, 4 * sizeof(int32_t), e->stream));
        hipLaunchKernelGGL(qm::brick_minmax_kernel, dim3(g.nbricks), dim3(64), 0, e->stream, g,
                           e->d_lut.p, reinterpret_cast<int4 *>(e->d_pmeta_raw.p), e->d_scalar.p);
        hipLaunchKernelGGL(qm::screen_prefix_kernel, dim3((g.nbricks + 255) / 256), dim3(256), 0,
                           e->stream, g, reinterpret_cast<const int4 *>(e->d_pmeta_raw.p),
                           reinterpret_cast<int4 *>(e->d_pmeta.p), e->d_ptotal.p);
        QM_HIP(hipGetLastError());
        total.resize(g.nbricks);
        QM_HIP(copy_back(total.data(), e->d_ptotal.p, (size_t)g.nbricks * sizeof(int32_t), e->stream));
        QM_HIP(hipStreamSynchronize(e->stream));
        wide.clear();
        for (int b = 0; b < g.nbricks; ++b)
            if (!qm::pair_fits(total[b], g.n_rows, KT, kPairLdsBytes)) wide.push_back(b);
        if ((int64_t)wide.size() * 200 <= g.nbricks) break;    // <= 0.5 % on the slow path
    }
    e->n_pwide = (int)wide.size();
    // an incoherent table (every shape leaves bricks that do not fit): the chunked kernels, whose
    // single-copy windows are half the size, take it
    // (with an explicit brick shape: whatever fits is paired, the rest goes to the direct kernel)
    e->pair_ok = fixed ? (int)wide.size() < g.nbricks : (int64_t)wide.size() * 200 <= g.nbricks;
    if (e->n_pwide) {
        if (e->d_pwide.ensure(wide.size())) return 1;
        QM_HIP(copy_in(e->d_pwide.p, wide.data(), wide.size() * sizeof(int32_t), e->stream));
    }
    if (e->pair_ok) {
        if (e->d_prel.ensure((size_t)g.nbricks * g.brick_nodes * g.row_pad)) return 1;
        hipLaunchKernelGGL(qm::pair_rel_kernel, dim3(g.nbricks), dim3(256), 0, e->stream, g,
                           e->d_lut.p, reinterpret_cast<const int4 *>(e->d_pmeta.p), e->d_ptotal.p,
                           KT, kPairLdsBytes, e->d_prel.p);
        QM_HIP(hipGetLastError());
    }
    QM_HIP(hipStreamSynchronize(e->stream));           // `wide` is a stack-lifetime buffer
    e->pg = g;
    e->pair_kt = KT;
    return 0;
}

// ---- shift-reuse layout (qm_shift.hpp) ----------------------------------------------------------
// Own brick grid (L.g, even brick dimensions: the kernel walks 2x2x2 node groups): the largest
// shape whose de-interleaved row windows fit 80 KB and whose groups' delay spread fits the
// register window for (almost) every brick; per-(brick, row) slot records and the record stream.
// e->shw (round 6): the same for the 8-wave shape with WIDE tiles in front -- both kinds of records and
// streams on one brick grid, a brick runs there if the windows of both tile kinds fit.
static int build_shift_tables(qm_engine *e, ShiftLayout &L, bool wide, bool wide_blocks);

// Outcome per resident table: the layout is built (L.ok), or the table does not qualify, or the
// tables could not be built -- most likely no memory for the record stream (4 S bytes per node, the
// table's size): that, too, is "does not qualify": the buffers are released, the error is dropped and
// the same step runs on the other kernels.
int ensure_shift_tables(qm_engine *e, ShiftLayout &L) {
    if (L.built) return 0;
    L.ok = false;
    const bool wide = &L == &e->shw;
    int rc = wide && e->cfg_shift_wide_rows == 2 ? 0 : build_shift_tables(e, L, wide, false);
    // (the wide layout: all rows of a brick in LDS where that fits -- up to ~36 rows of C3's geometry; beyond 64
    // rows, where the 256-sample tiles run on row blocks too, row blocks of <= 20 rows on 4x4x4 bricks: a C3 grid
    // x 128 rows x 1536 samples 44.0 ms against 48.6.  In between -- BASELINE configs[3]: 60 rows -- the row-block
    // form LOSES to the 8-wave kernel that holds all rows' 256-sample windows (a C4 slab 177.3 against 168.2 ms,
    // C3 x 60 rows 22.3 against 21.2, profiles/r06_ab_runs.txt): only on request there, shift_wide_rows = 2)
    if (wide && rc == 0 && !L.ok && e->cfg_shift_wide_rows != 0 &&
        (e->g.n_rows > qm::kShiftMaxRows || e->cfg_shift_wide_rows == 2))
        rc = build_shift_tables(e, L, true, true);
    if (rc != 0 || !L.ok) {
        L.ok = false;
        L.release();
        if (rc != 0) {
            (void)hipGetLastError();                    // (an allocation failure is not sticky)
            clear_error();
        }
    }
    L.built = true;
    return 0;
}

static int build_shift_tables(qm_engine *e, ShiftLayout &L, bool wide, bool wide_blocks) {
    const int S = e->g.n_rows;
    L.wide = wide;
    // More rows than a CU's LDS holds windows for: row blocks (stack_shift_rows_kernel) -- bricks of
    // 4x4x4 nodes = one 2x2x2 group per wavefront of the 8-wave workgroup, whose accumulators stay
    // in registers while the rows are staged in nblk blocks of sb <= 64 rows.
    const bool blocks = wide ? wide_blocks : S > qm::kShiftMaxRows;
    if (wide && !blocks && S > qm::kShiftMaxRows) return 0;        // (all rows of a brick in LDS at once)
    if (wide && blocks && (S < 2 || e->cfg_bx > 0)) return 0;
    // two forms (qm_shift.hpp): blocks of <= 34 rows staged by LDS-direct loads into the idle half of
    // a double-buffered LDS (default), or blocks of <= 64 staged through registers between two barriers
    // (that one only from 97 rows on: at 65-96 two blocks of <= 48 rows stage as often as they
    // compute and the chunked kernel with its 8x8x8 bricks is 4-5 % faster, profiles/r03_ab_runs.txt)
    // (round 4, form 2: the LDS-direct staging with TWO 4-wave workgroups per CU on bricks of 4x4x2
    // nodes, single-buffered -- stack_shift_rows4_kernel)
    // (wide tiles: the double-buffered 8-wave form only, blocks of <= 20 rows -- 20 x (384 + span) samples in
    // each 80 KB half)
    const int form = wide ? 1 : e->cfg_shift_rows_direct;   // 0 registers, 1 double-buffered 8 waves, 2 two x 4 waves
    const bool direct = form != 0;
    const bool quad = form == 2;
    if (blocks && !direct && S <= 96 && e->cfg_shift != 1) return 0;
    const int block_rows = wide ? 20 : direct ? 34 : qm::kShiftMaxRows;
    const int nblk = blocks ? (S + block_rows - 1) / block_rows : 1;
    const int sb = blocks ? ((S + nblk - 1) / nblk + 1) / 2 * 2 : S;
    if (S > 1024 || (blocks && !wide && e->cfg_shift_waves != 0 &&
                     e->cfg_shift_waves != (quad ? qm::kShiftWaves : qm::kShiftWaves8)))
        return 0;
    // a grid one node thick has half-empty 2x2x2 groups everywhere (e.g. the flat 1 x 1 x N view
    // of the reference-signature migrate): leave it to the other kernels unless asked explicitly
    if (e->cfg_shift < 0 && (e->g.nx < 2 || e->g.ny < 2 || e->g.nz < 2)) return 0;
    // Workgroup shape (qm_shift.hpp): two 4-wave workgroups per CU up to ~32 rows; beyond, ONE 8-wave
    // workgroup with all 160 KB (smaller bricks, 33-64 rows); 12 waves only on request.  Bricks are
    // shaped so that their 2x2x2 groups deal evenly over the wavefronts.
    static const int kShapes4[][3] = {{8, 8, 8}, {4, 8, 8}, {4, 4, 8}, {4, 4, 4}, {2, 4, 4}};
    static const int kShapes8[][3] = {{8, 8, 8}, {4, 8, 8}, {4, 4, 8}, {4, 4, 4}, {4, 4, 4}};
    static const int kShapes12[][3] = {{8, 8, 12}, {8, 8, 6}, {4, 8, 6}, {4, 4, 6}, {2, 4, 6}};
    // (wide tiles, 8 waves: 8x8x16 gives a wavefront sixteen groups per brick, as 8x8x8 does on four waves)
    static const int kShapesWide[][3] = {{8, 8, 16}, {8, 8, 8}, {4, 8, 8}, {4, 4, 8}, {4, 4, 4}};
    int candidates[2] = {qm::kShiftWaves, qm::kShiftWaves8};
    int n_candidates = 2;
    if (wide) {
        candidates[0] = qm::kShiftWaves8;
        n_candidates = 1;
    } else if (e->cfg_shift_waves != 0) {
        candidates[0] = e->cfg_shift_waves;
        n_candidates = 1;
    } else if (blocks && quad) {
        candidates[0] = qm::kShiftWaves;
        n_candidates = 1;
    } else if (S > 40) {                               // (80 KB cannot hold that many row windows)
        candidates[0] = qm::kShiftWaves8;
        n_candidates = 1;
    }
    const bool fixed = e->cfg_bx > 0 && !blocks;
    int n_shapes = fixed || blocks ? 1 : 5;
    // Round 6: a grid whose extent along one axis ends a node or two into a brick (the Icequake-sized C1: 57 nodes
    // in z = seven bricks of eight and one layer) pays a whole brick's staging and barriers for that layer.  Where a
    // FLATTER brick along one axis fills the grid's bounding bricks at least 4 % better than 8x8x8, it is tried
    // first (C1: 88 -> 94 %, a step 0.380 -> 0.363 ms at one timestep per launch, 0.352 -> 0.323 in the sweep;
    // C3 and its slabs: equal utilisation either way, 8x8x8 stays -- flatter bricks cost them 3 %,
    // profiles/r06b_ab_bricks.txt).  Four-wave shape only (tables of up to ~32 rows).
    int shapes4[6][3];
    for (int i = 0; i < 5; ++i)
        for (int k = 0; k < 3; ++k) shapes4[i][k] = kShapes4[i][k];
    if (!fixed && !blocks && !wide) {
        auto filled = [&](int bx, int by, int bz) {
            auto up = [](int n, int b) { return (int64_t)((n + b - 1) / b) * b; };
            return (double)e->n_nodes / (double)(up(e->g.nx, bx) * up(e->g.ny, by) * up(e->g.nz, bz));
        };
        static const int kFlat[][3] = {{8, 8, 4}, {8, 4, 8}, {4, 8, 8}};
        int best = -1;
        double most = 1.04 * filled(8, 8, 8);
        for (int i = 0; i < 3; ++i)
            if (filled(kFlat[i][0], kFlat[i][1], kFlat[i][2]) > most) {
                most = filled(kFlat[i][0], kFlat[i][1], kFlat[i][2]);
                best = i;
            }
        if (best >= 0) {
            for (int i = 5; i > 0; --i)
                for (int k = 0; k < 3; ++k) shapes4[i][k] = shapes4[i - 1][k];
            for (int k = 0; k < 3; ++k) shapes4[0][k] = kFlat[best][k];
            n_shapes = 6;
        }
    }
    static const int kShapesBlocks[][3] = {{4, 4, 4}};
    static const int kShapesBlocks4[][3] = {{4, 4, 2}};
    int nw = candidates[0];
    qm::GridDesc g = e->g;
    std::vector<int32_t> fit, fitw, list;
    bool ok = false;
    auto even_up = [](int v) { return v + (v & 1); };
    for (int cand = 0; cand < n_candidates && !ok; ++cand) {
    nw = candidates[cand];
    const int (*kShapes)[3] = blocks ? (quad ? kShapesBlocks4 : kShapesBlocks) : wide ? kShapesWide
                              : nw == qm::kShiftWaves3 ? kShapes12 : nw == qm::kShiftWaves8 ? kShapes8 : shapes4;
    const int n_try = kShapes == shapes4 ? n_shapes : std::min(n_shapes, 5);
    for (int s = 0; s < n_try; ++s) {
        g = e->g;
        g.bx = std::min(even_up(fixed ? e->cfg_bx : kShapes[s][0]), even_up(g.nx));
        g.by = std::min(even_up(fixed ? e->cfg_by : kShapes[s][1]), even_up(g.ny));
        g.bz = std::min(even_up(fixed ? e->cfg_bz : kShapes[s][2]), even_up(g.nz));
        g.nbx = (g.nx + g.bx - 1) / g.bx;
        g.nby = (g.ny + g.by - 1) / g.by;
        g.nbz = (g.nz + g.bz - 1) / g.bz;
        g.nbricks = g.nbx * g.nby * g.nbz;
        g.brick_nodes = g.bx * g.by * g.bz;
        const size_t br = (size_t)g.nbricks * S;
        const size_t nvb = (size_t)g.nbricks * nblk;               // (brick, row block) pairs
        // (meta: + 4 KB of slack -- the row-block loops prefetch that much metadata ahead)
        if (L.raw.ensure(4 * br) || L.meta.ensure(4 * nvb * sb + 1024) ||
            L.total.ensure(nvb) || L.fit.ensure(nvb) || e->d_scalar.ensure(12))
            return 1;
        if (wide && (L.wmeta.ensure(4 * nvb * sb + 1024) || L.wtotal.ensure(nvb) || e->d_work.ensure(nvb))) return 1;
        QM_HIP(hipMemsetAsync(e->d_scalar.p, 0, 12 * sizeof(int32_t), e->stream));
        hipLaunchKernelGGL(qm::brick_minmax_kernel, dim3(g.nbricks), dim3(64), 0, e->stream, g,
                           e->d_lut.p, reinterpret_cast<int4 *>(L.raw.p), e->d_scalar.p);
        unsigned long long *const tally_d = reinterpret_cast<unsigned long long *>(e->d_scalar.p + 4);
        unsigned long long tally[4] = {0, 0, 0, 0};
        fit.assign(nvb, 1);
        if (!(wide && blocks)) {                          // (wide row blocks: wide tiles only, no 256-sample ones)
            hipLaunchKernelGGL(qm::shift_need_kernel, dim3((unsigned)nvb), dim3(256), 0, e->stream, g,
                               e->d_lut.p, reinterpret_cast<const int4 *>(L.raw.p),
                               reinterpret_cast<int4 *>(L.meta.p), L.total.p, L.fit.p, tally_d,
                               blocks && direct ? qm::kShiftPlane : qm::shift_plane(nw), nblk, sb, 0);
            QM_HIP(hipGetLastError());
            QM_HIP(copy_back(fit.data(), L.fit.p, nvb * sizeof(int32_t), e->stream));
            QM_HIP(copy_back(tally, tally_d, sizeof(tally), e->stream));
            QM_HIP(hipStreamSynchronize(e->stream));
            L.quads = (int64_t)tally[0];
            L.group_rows = (int64_t)tally[1];
            L.stage_slots = (int)std::min<unsigned long long>(tally[2], 1u << 30);
            L.stage_reach = (int)std::min<unsigned long long>(tally[3], 1u << 30);
        }
        if (wide) {
            // the wide tiles' row windows: 384 + span samples in ONE contiguous region (slots of 32 bytes): all
            // 160 KB, or -- row blocks -- one of the two 80 KB halves
            QM_HIP(hipMemsetAsync(tally_d, 0, sizeof(tally), e->stream));
            hipLaunchKernelGGL(qm::shift_need_kernel, dim3((unsigned)nvb), dim3(256), 0, e->stream, g,
                               e->d_lut.p, reinterpret_cast<const int4 *>(L.raw.p),
                               reinterpret_cast<int4 *>(L.wmeta.p), L.wtotal.p, e->d_work.p, tally_d,
                               blocks ? qm::kShiftPlane : qm::kShiftLdsBytes8 / 2, nblk, sb, 1);   // (a half holds
                               // 2 x kShiftPlane bytes, as for the two-plane blocks: the launch's 163712 bytes)
            QM_HIP(hipGetLastError());
            fitw.resize(nvb);
            QM_HIP(copy_back(fitw.data(), e->d_work.p, nvb * sizeof(int32_t), e->stream));
            QM_HIP(copy_back(tally, tally_d, sizeof(tally), e->stream));
            QM_HIP(hipStreamSynchronize(e->stream));
            L.wquads = (int64_t)tally[0];
            if (blocks) {
                L.group_rows = (int64_t)tally[1];
                L.stage_slots = (int)std::min<unsigned long long>(tally[2], 1u << 30);
                L.stage_reach = (int)std::min<unsigned long long>(tally[3], 1u << 30);
            }
            for (size_t i = 0; i < nvb; ++i) fit[i] &= fitw[i];
        }
        list.clear();
        for (int b = 0; b < g.nbricks; ++b) {                      // a brick fits if all its blocks do
            int all = 1;
            for (int k = 0; k < nblk; ++k) all &= fit[(size_t)b * nblk + k];
            fit[b] = all;
            if (!all) list.push_back(b);
        }
        ok = fixed ? (int)list.size() < g.nbricks : (int64_t)list.size() * 200 <= g.nbricks;
        if (ok) break;
    }
    }
    if (!ok) return 0;                                   // an incoherent table: the other kernels
    const int rows2 = sb + (sb & 1);
    const int64_t words = (int64_t)g.nbricks * nw * nblk * qm::shift_recs_per_wave(g, rows2, nw) *
                          (qm::shift_rec_bytes(blocks) / 4);
    if (blocks || wide)                                  // per-brick verdicts for the kernels
        QM_HIP(copy_in(L.fit.p, fit.data(), (size_t)g.nbricks * sizeof(int32_t), e->stream));
    // (+ slack: the loop loads one record past a wavefront's run and touches the line 16 records
    // ahead with its L2 prefetch -- after the last run of the last brick that is past the stream)
    if (!(wide && blocks) && L.stream.ensure((size_t)words + 4096)) return 1;
    if (wide && L.wstream.ensure((size_t)words + 4096)) return 1;
    const size_t hdr_bytes = (size_t)qm::shift_groups_per_brick(g) * rows2 * sizeof(uint2);
    QM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(qm::shift_stream_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)hdr_bytes));
    if (!(wide && blocks)) {
        hipLaunchKernelGGL(qm::shift_stream_kernel, dim3((unsigned)((size_t)g.nbricks * nblk)), dim3(256),
                           hdr_bytes, e->stream, g, e->d_lut.p,
                           reinterpret_cast<const int4 *>(L.meta.p), L.total.p, L.fit.p,
                           rows2, nw, nblk, sb, qm::shift_packed(blocks) ? 1 : 0, 0, L.stream.p);
        QM_HIP(hipGetLastError());
    }
    if (wide) {
        hipLaunchKernelGGL(qm::shift_stream_kernel, dim3((unsigned)((size_t)g.nbricks * nblk)), dim3(256),
                           hdr_bytes, e->stream, g, e->d_lut.p,
                           reinterpret_cast<const int4 *>(L.wmeta.p), L.wtotal.p, L.fit.p,
                           rows2, nw, nblk, sb, qm::shift_packed(blocks) ? 1 : 0, 1, L.wstream.p);
        QM_HIP(hipGetLastError());
    }
    L.n_list = (int)list.size();
    if (L.n_list) {
        if (L.list.ensure(list.size())) return 1;
        QM_HIP(copy_in(L.list.p, list.data(), list.size() * sizeof(int32_t), e->stream));
    }
    QM_HIP(hipStreamSynchronize(e->stream));           // `list` is a stack-lifetime buffer
    L.raw.release();
    L.g = g;
    L.rows2 = rows2;
    L.nw = nw;
    L.nblk = nblk;
    L.sb = sb;
    L.direct = blocks && direct;
    L.quad = blocks && quad;
    L.ok = true;
    return 0;
}

// ---- float32 screening path (qm_screen.hpp): launch plan and the sweep's own offset tables ----
bool screen_plan_feasible(const qm_engine *e, int S, const ScreenPlan &p) {
    const int64_t rows_bytes = (int64_t)S * (8 * p.kt() - 8);
    return S <= 64 && (p.big || p.jp < 4) && rows_bytes * 5 <= (int64_t)p.lds_bytes(e) * 4;
}

// plan for a scan of n_samples (0 = unknown: the most LDS-hungry plan that could be chosen, for
// the brick-shape decision at load time)
ScreenPlan screen_plan(const qm_engine *e, int S, int n_samples) {
    ScreenPlan best;
    if (!e->cfg_screen || e->cfg_force_direct || (e->user_waves && e->cfg_waves != 8)) return best;
    double best_cost = 1e300;
    // relative cost per sample, measured on C3 / C4-sized tables (tools/ab.py): with the integer
    // sweep two pairs per lane in two 8-wave workgroups per CU run as fast as four pairs in one
    // 16-wave workgroup, need no scratch and pad short scans less
    const struct { int jp; bool big; double cost; } options[] = {
        {2, false, 1.00}, {4, true, 1.005}, {2, true, 1.05}, {1, false, 1.35}, {1, true, 1.35}};
    for (const auto &o : options) {
        ScreenPlan p;
        p.jp = o.jp;
        p.big = o.big;
        if (e->cfg_screen_pairs && o.jp != e->cfg_screen_pairs) continue;
        if (e->cfg_screen_big >= 0 && (int)o.big != e->cfg_screen_big) continue;
        if (!screen_plan_feasible(e, S, p)) continue;
        double cost = o.cost;
        if (n_samples > 0) cost *= (double)((n_samples + p.kt() - 1) / p.kt()) * p.kt();
        // unknown length: the plan that leaves the least LDS to the delay spans (the binding one)
        else cost = (double)p.window_bytes(e) - (double)S * (8 * p.kt() - 8);
        if (cost < best_cost) {
            best_cost = cost;
            best = p;
        }
    }
    return best;
}

// The sweep has its own brick grid (e->sg): its LDS budget and window layout differ from the
// float64 kernel's, so the largest brick shape whose windows fit is chosen for it separately.
int ensure_screen_tables(qm_engine *e, const ScreenPlan &plan) {
    const int KT = plan.kt();
    const int wb = plan.window_bytes(e);
    if (e->screen_kt == KT && e->screen_wb == wb) return 0;
    static const int kShapes[][3] = {{16, 8, 8}, {8, 8, 8}, {4, 8, 8}, {4, 4, 8}, {4, 4, 4},
                                     {2, 4, 4},  {2, 2, 4}, {2, 2, 2}, {1, 1, 2}, {1, 1, 1}};
    const bool fixed = e->cfg_bx > 0;
    const int n_shapes = fixed ? 1 : (int)(sizeof(kShapes) / sizeof(kShapes[0]));
    const int first = fixed ? 0 : (e->cfg_screen_brick16 ? 0 : 1);
    qm::GridDesc g = e->g;
    std::vector<int32_t> total, wide;
    for (int s = first; s < std::max(n_shapes, first + 1); ++s) {
        g = e->g;
        if (!fixed) {
            g.bx = std::min(kShapes[s][0], g.nx);
            g.by = std::min(kShapes[s][1], g.ny);
            g.bz = std::min(kShapes[s][2], g.nz);
            g.nbx = (g.nx + g.bx - 1) / g.bx;
            g.nby = (g.ny + g.by - 1) / g.by;
            g.nbz = (g.nz + g.bz - 1) / g.bz;
            g.nbricks = g.nbx * g.nby * g.nbz;
            g.brick_nodes = g.bx * g.by * g.bz;
        }
        const size_t br = (size_t)g.nbricks * g.n_rows;
        if (e->d_smeta_raw.ensure(4 * br) || e->d_smeta.ensure(4 * br) ||
            e->d_stotal.ensure(g.nbricks) || e->d_scalar.ensure(4))
            return 1;
        QM_HIP(hipMemsetAsync(e->d_scalar.p, 0, 4 * sizeof(int32_t), e->stream));
        hipLaunchKernelGGL(qm::brick_minmax_kernel, dim3(g.nbricks), dim3(64), 0, e->stream, g,
                           e->d_lut.p, reinterpret_cast<int4 *>(e->d_smeta_raw.p), e->d_scalar.p);
        hipLaunchKernelGGL(qm::screen_prefix_kernel, dim3((g.nbricks + 255) / 256), dim3(256), 0,
                           e->stream, g, reinterpret_cast<const int4 *>(e->d_smeta_raw.p),
                           reinterpret_cast<int4 *>(e->d_smeta.p), e->d_stotal.p);
        QM_HIP(hipGetLastError());
        total.resize(g.nbricks);
        QM_HIP(copy_back(total.data(), e->d_stotal.p, (size_t)g.nbricks * sizeof(int32_t), e->stream));
        QM_HIP(hipStreamSynchronize(e->stream));
        wide.clear();
        for (int b = 0; b < g.nbricks; ++b)
            if (!qm::screen_fits(total[b], g.n_rows, KT, wb)) wide.push_back(b);
        if ((int64_t)wide.size() * 200 <= g.nbricks) break;    // <= 0.5 % on the slow path
    }
    e->n_swide = (int)wide.size();
    if (e->n_swide) {
        if (e->d_swide.ensure(wide.size())) return 1;
        QM_HIP(copy_in(e->d_swide.p, wide.data(), wide.size() * sizeof(int32_t), e->stream));
    }
    if (e->d_srel.ensure((size_t)g.nbricks * g.brick_nodes * g.row_pad)) return 1;
    hipLaunchKernelGGL(qm::screen_rel_kernel, dim3(g.nbricks), dim3(256), 0, e->stream, g,
                       e->d_lut.p, reinterpret_cast<const int4 *>(e->d_smeta.p), e->d_stotal.p, KT,
                       wb, e->d_srel.p);
    QM_HIP(hipGetLastError());
    QM_HIP(hipStreamSynchronize(e->stream));           // `wide` is a stack-lifetime buffer
    e->sg = g;
    e->screen_kt = KT;
    e->screen_wb = wb;
    return 0;
}

extern "C" {

int qm_engine_load_lut(qm_engine *e, const int32_t *lut, int lut_on_device, int32_t nx,
                       int32_t ny, int32_t nz, int32_t n_rows, int64_t node_offset) {
    if (!e || !lut) return fail("qm_engine_load_lut: NULL argument");
    if (nx < 1 || ny < 1 || nz < 1 || n_rows < 1) return fail("bad table shape");
    const int64_t n_nodes = (int64_t)nx * ny * nz;
    if (n_nodes >= INT32_MAX) return fail("more than 2^31-1 nodes on one GPU is not supported");
    DeviceGuard guard(e->device);
    // A key names the table that is loaded after qm_engine_table_select reported a miss (nothing is
    // resident then).  A load on top of a resident table replaces it WITHOUT a select -- e.g. the
    // reference-signature functions on the shared default engine between two steps of a MigrationScan
    // (core/lib.py: _resident) -- and must not inherit that table's key: the key's owner would be told
    // "resident" and stack through a foreign table.
    if (e->have_lut) e->cur_keyed = false;
    e->have_lut = false;
    const size_t lut_elems = (size_t)n_nodes * n_rows;
    if (e->d_lut.ensure(lut_elems) || e->d_scalar.ensure(4)) return 1;
    if (lut_on_device)
        QM_HIP(hipMemcpyAsync(e->d_lut.p, lut, lut_elems * sizeof(int32_t), hipMemcpyDeviceToDevice,
                              e->stream));
    else
        QM_HIP(copy_in(e->d_lut.p, lut, lut_elems * sizeof(int32_t), e->stream));

    // Brick shape: the configured one, or (brick_x == 0) the largest candidate whose windows fit
    // the LDS budget for (almost) every brick -- larger bricks amortise window staging, smaller
    // ones have smaller delay spans.  Bricks that still do not fit go to the direct kernel.
    static const int kShapes[][3] = {{8, 8, 8}, {4, 8, 8}, {4, 4, 8}, {4, 4, 4},
                                     {2, 4, 4}, {2, 2, 4}, {2, 2, 2}, {1, 1, 2}, {1, 1, 1}};
    const int n_shapes = e->cfg_bx > 0 ? 1 : (int)(sizeof(kShapes) / sizeof(kShapes[0]));
    e->n_rows_hint = n_rows;
    e->auto_j = 0;
    qm::GridDesc g{};
    // Per candidate shape ONE pass over the table (min / span per (brick, row), the bricks' totals,
    // the table's largest delay); whether a shape's windows fit depends on the tile length and the
    // LDS budget and is decided on the host from the cached totals -- the layout search below asks
    // for up to seven (tile length, budget) pairs, which used to cost as many passes and host
    // round trips per load.  `on_device`: the shape whose records d_bmeta / d_btotal hold.
    std::vector<std::vector<int32_t>> totals(n_shapes);
    std::vector<qm::GridDesc> shapes(n_shapes);
    int on_device = -1;
    auto measure = [&](int s) -> int {
        qm::GridDesc &gs = shapes[s];
        gs = qm::GridDesc{};
        gs.nx = nx; gs.ny = ny; gs.nz = nz;
        gs.bx = std::min(e->cfg_bx > 0 ? e->cfg_bx : kShapes[s][0], (int)nx);
        gs.by = std::min(e->cfg_bx > 0 ? e->cfg_by : kShapes[s][1], (int)ny);
        gs.bz = std::min(e->cfg_bx > 0 ? e->cfg_bz : kShapes[s][2], (int)nz);
        gs.nbx = (nx + gs.bx - 1) / gs.bx;
        gs.nby = (ny + gs.by - 1) / gs.by;
        gs.nbz = (nz + gs.bz - 1) / gs.bz;
        const int64_t nbricks = (int64_t)gs.nbx * gs.nby * gs.nbz;
        if (nbricks >= INT32_MAX) return fail("too many bricks");
        gs.nbricks = (int)nbricks;
        gs.brick_nodes = gs.bx * gs.by * gs.bz;
        gs.n_rows = n_rows;
        gs.row_pad = (n_rows + 7) / 8 * 8;
        const size_t br = (size_t)nbricks * n_rows;
        if (e->d_bmeta.ensure(4 * br) || e->d_btotal.ensure(nbricks)) return 1;
        QM_HIP(hipMemsetAsync(e->d_scalar.p, 0, 4 * sizeof(int32_t), e->stream));
        hipLaunchKernelGGL(qm::brick_minmax_kernel, dim3(gs.nbricks), dim3(64), 0, e->stream, gs,
                           e->d_lut.p, reinterpret_cast<int4 *>(e->d_bmeta.p), e->d_scalar.p);
        QM_HIP(hipGetLastError());
        hipLaunchKernelGGL(qm::brick_prefix_kernel, dim3((gs.nbricks + 255) / 256), dim3(256),
                           0, e->stream, gs, reinterpret_cast<int4 *>(e->d_bmeta.p),
                           e->d_btotal.p);
        QM_HIP(hipGetLastError());
        totals[s].resize(nbricks);
        QM_HIP(copy_back(totals[s].data(), e->d_btotal.p, nbricks * sizeof(int32_t), e->stream));
        QM_HIP(copy_back(&e->lut_max, e->d_scalar.p, sizeof(int32_t), e->stream));
        QM_HIP(hipStreamSynchronize(e->stream));
        on_device = s;
        return 0;
    };
    // largest candidate shape whose windows fit for tile length 64 * J under the current budget
    // (result: g and e->h_btotal; `chosen` = its index)
    int chosen = 0;
    auto search = [&](int J) -> int {
        const int KT = qm::kWave * J;
        for (int s = 0; s < n_shapes; ++s) {
            if (totals[s].empty() && measure(s)) return 1;
            chosen = s;
            int64_t wide = 0;
            for (int32_t t : totals[s])
                if (!qm::brick_fits(t, n_rows, KT, lds_cap_doubles(e))) ++wide;
            if (wide * 200 <= (int64_t)totals[s].size()) break;   // <= 0.5 % of the bricks on the slow path
        }
        g = shapes[chosen];
        return 0;
    };
    if (e->cfg_j == 0 && e->cfg_bx == 0 && !e->user_waves && !e->user_lds) {
        // Automatic layout.  All S windows of a brick sit in LDS together, so workgroup shape
        // (two 8-wave workgroups with 80 KB each, or one 16-wave workgroup with all 160 KB --
        // measured 4 % slower at equal bricks: barriers), samples per lane and brick size trade
        // against each other: more samples per lane cost less per sample (measured 1.0 / 1.12 /
        // 1.4 for 4 / 2 / 1) but leave less room for the delay spans, and smaller bricks amortise
        // their staging over fewer nodes (measured on 20-200 rows: time ~ 1 + 30 / nodes per
        // brick).  Up to 64 rows the samples per lane follow from the budget (eff_j) and an
        // exact-row-count kernel exists for one of them (3 % faster); beyond, both 2 and 1 are
        // tried.  C3 (30 rows) keeps 2 x 80 KB; 33-64 rows and coarse grids get 160 KB.
        double best_cost = 1e300;
        int best_j = 0, best_waves = 8, best_lds = 80 * 1024;
        for (int single = 0; single < 2; ++single) {
            e->cfg_waves = single ? 16 : 8;
            e->cfg_lds_bytes = single ? 160 * 1024 : 80 * 1024;
            const int j_budget = eff_j(e);
            for (int j : {4, 2, 1}) {
                if (j > j_budget || (j == 1 && j_budget > 1 && n_rows <= 64)) continue;
                if (j == 4 && n_rows > 40 && !e->cfg_exact) continue;   // exact kernels only
                if (search(j)) return 1;
                // (four samples per lane beyond 40 rows keep a ring of four offset chunks in
                // registers instead of the whole node's: 7 % ahead of two samples per lane at
                // equal bricks on the C3 grid x 60 rows)
                double cost = (j == 4 ? (n_rows > 40 ? 1.04 : 1.0) : j == 2 ? 1.12 : 1.4) *
                              (1.0 + 30.0 / g.brick_nodes) * (single ? 1.04 : 1.0);
                if (e->cfg_exact && qm::exact_built(n_rows, j)) cost *= 0.97;
                if (cost < best_cost) {
                    best_cost = cost;
                    best_j = j;
                    best_waves = e->cfg_waves;
                    best_lds = e->cfg_lds_bytes;
                }
            }
        }
        e->cfg_waves = best_waves;
        e->cfg_lds_bytes = best_lds;
        e->auto_j = best_j;
    } else if (!e->user_waves && !e->user_lds) {
        // brick shape or samples per lane given: the workgroup shape by the row count alone
        const bool big = n_rows > 40;
        e->cfg_waves = big ? 16 : 8;
        e->cfg_lds_bytes = big ? 160 * 1024 : 80 * 1024;
    }
    if (search(eff_j(e))) return 1;
    if (on_device != chosen && measure(chosen)) return 1;   // the chosen shape's records on the device
    g = shapes[chosen];
    e->h_btotal = totals[chosen];
    // (the window-offset table of the round-2 kernels -- 2 bytes per table entry padded to 8 rows --
    // is built when one of them first runs: ensure_rel; tables the shift-reuse kernel takes never
    // need it)
    e->rel_built = false;
    QM_HIP(hipStreamSynchronize(e->stream));
    e->g = g;
    e->n_nodes = n_nodes;
    e->node_offset = node_offset;
    e->tab_waves = e->cfg_waves;                        // (what the layout search left in the tunables)
    e->tab_lds_bytes = e->cfg_lds_bytes;
    e->plan_j = -1;
    e->screen_kt = 0;
    e->pair_kt = 0;
    e->sh.built = e->sh.ok = false;
    e->shw.built = e->shw.ok = false;
    e->have_lut = true;
    static std::atomic<uint64_t> next_serial{1};
    e->serial = next_serial.fetch_add(1);
    return plan_wide(e, eff_j(e));
}

int qm_engine_table_select(qm_engine *e, uint64_t key, int32_t capacity, int32_t *resident) {
    if (!e || !resident) return fail("qm_engine_table_select: NULL argument");
    if (capacity < 0 || capacity > 64) return fail("qm_engine_table_select: capacity must be in 0..64");
    *resident = 0;
    if (e->cur_keyed && e->cur_key == key && e->have_lut) {
        *resident = 1;
        ++e->table_hits;
        return 0;
    }
    DeviceGuard guard(e->device);
    TableState &cur = *e;
    const bool park = e->have_lut && e->cur_keyed && capacity > 0;
    // Is the requested table parked?  Then it trades places with the current one in ITS slot -- looked
    // up before anything is evicted: with the slots full, the least recently used one may be the very
    // table that is asked for (cache of 1, tables A and B alternating: no rebuild).
    for (TableSlot &sl : e->slots) {
        if (!sl.used || sl.key != key) continue;
        std::swap(cur, sl.state);                       // the slot now holds what was being worked on
        if (park) {
            sl.key = e->cur_key;
            sl.stamp = ++e->table_clock;
        } else {                                        // (an un-keyed table, or nothing may be parked)
            sl.state.release_all();
            sl.state = TableState{};
            sl.used = false;
        }
        e->cur_key = key;
        e->cur_keyed = true;
        if (!e->user_waves && e->tab_waves) e->cfg_waves = e->tab_waves;
        if (!e->user_lds && e->tab_lds_bytes) e->cfg_lds_bytes = e->tab_lds_bytes;
        *resident = 1;
        ++e->table_hits;
        return 0;
    }
    // Not parked: the caller will load it.  Park the table being worked on (if it has a key: one
    // loaded without a key is simply replaced)
    if (park) {
        TableSlot *slot = nullptr;
        for (TableSlot &sl : e->slots)
            if (!sl.used) { slot = &sl; break; }
        if (!slot && (int)e->slots.size() < capacity) {
            e->slots.emplace_back();
            slot = &e->slots.back();
        }
        if (!slot) {                                    // evict the least recently used
            slot = &e->slots[0];
            for (TableSlot &sl : e->slots)
                if (sl.stamp < slot->stamp) slot = &sl;
            // (its device memory goes back to the pool: one wait for work that may still read it)
            slot->state.release_all();
            slot->state = TableState{};
            ++e->table_evictions;
        }
        std::swap(cur, slot->state);                    // the engine now holds the slot's empty state
        slot->key = e->cur_key;
        slot->stamp = ++e->table_clock;
        slot->used = true;
    } else if (e->have_lut) {
        // nothing may be parked: keep the buffers for the next table (load_lut reuses allocations)
        e->have_lut = false;
        e->sh.built = e->sh.ok = false;
        e->shw.built = e->shw.ok = false;
        e->pair_kt = 0;
        e->screen_kt = 0;
        e->plan_j = -1;
    }
    e->cur_key = key;
    e->cur_keyed = true;
    ++e->table_misses;
    return 0;
}

int qm_engine_grids_begin(qm_engine *e, int32_t nx, int32_t ny, int32_t nz, int32_t n_grids) {
    if (!e) return fail("engine is NULL");
    if (nx < 1 || ny < 1 || nz < 1 || n_grids < 1) return fail("bad grid shape");
    DeviceGuard guard(e->device);
    if (e->d_grids.ensure((size_t)n_grids * nx * ny * nz)) return 1;
    e->gx = nx; e->gy = ny; e->gz = nz; e->g_rows = n_grids;
    return 0;
}

int qm_engine_grids_set(qm_engine *e, int32_t index, const double *grid, int on_device) {
    if (!e || !grid) return fail("NULL argument");
    if (index < 0 || index >= e->g_rows) return fail("grid index %d out of range", index);
    DeviceGuard guard(e->device);
    const size_t n = (size_t)e->gx * e->gy * e->gz;
    if (on_device)
        QM_HIP(hipMemcpyAsync(e->d_grids.p + (size_t)index * n, grid, n * sizeof(double),
                              hipMemcpyDeviceToDevice, e->stream));
    else
        QM_HIP(copy_in(e->d_grids.p + (size_t)index * n, grid, n * sizeof(double), e->stream));
    return 0;
}

int qm_engine_serve(qm_engine *e, double sampling_rate, const int32_t *rows, int32_t n_rows,
                    int32_t dfx, int32_t dfy, int32_t dfz, int64_t node_offset) {
    if (!e || !rows) return fail("NULL argument");
    if (e->g_rows < 1) return fail("no travel-time grids resident: call qm_engine_grids_begin/set");
    if (n_rows < 1) return fail("no rows selected");
    if (dfx < 1 || dfy < 1 || dfz < 1) return fail("decimation factors must be >= 1");
    for (int i = 0; i < n_rows; ++i)
        if (rows[i] < 0 || rows[i] >= e->g_rows) return fail("row %d selects grid %d of %d", i, rows[i], e->g_rows);
    DeviceGuard guard(e->device);
    qm::ServeArgs a{};
    a.nxf = e->gx; a.nyf = e->gy; a.nzf = e->gz;
    a.dfx = dfx; a.dfy = dfy; a.dfz = dfz;
    // Grid3D.decimate (lut.py:121-122): new = 1 + (n - 1) // df ; c1 = (n - df*(new-1) - 1) // 2
    a.nx = 1 + (e->gx - 1) / dfx; a.ny = 1 + (e->gy - 1) / dfy; a.nz = 1 + (e->gz - 1) / dfz;
    a.c1x = (e->gx - dfx * (a.nx - 1) - 1) / 2;
    a.c1y = (e->gy - dfy * (a.ny - 1) - 1) / 2;
    a.c1z = (e->gz - dfz * (a.nz - 1) - 1) / 2;
    a.S = n_rows;
    a.rate = sampling_rate;
    const int64_t n_out = (int64_t)a.nx * a.ny * a.nz;
    if (e->d_rows.ensure(n_rows) || e->d_served.ensure((size_t)n_out * n_rows)) return 1;
    QM_HIP(copy_in(e->d_rows.p, rows, n_rows * sizeof(int32_t), e->stream));
    a.grids = e->d_grids.p;
    a.rows = e->d_rows.p;
    a.out = e->d_served.p;
    // 256 nodes per workgroup while their rows fit 64 KB of LDS (up to 63 rows), else 64
    const int pitch = (n_rows + 1) | 1;
    const bool wide = (size_t)256 * pitch * sizeof(int32_t) <= 64 * 1024;
    const int npb = wide ? 256 : 64;
    const size_t lds = (size_t)npb * pitch * sizeof(int32_t);
    if (lds > 64 * 1024) return fail("too many rows (%d) for the serving kernel", n_rows);
    if (wide)
        hipLaunchKernelGGL(qm::serve_table_kernel<256>, dim3((unsigned)((n_out + 255) / 256)), dim3(256),
                           lds, e->stream, a);
    else
        hipLaunchKernelGGL(qm::serve_table_kernel<64>, dim3((unsigned)((n_out + 63) / 64)), dim3(256), lds,
                           e->stream, a);
    QM_HIP(hipGetLastError());
    return qm_engine_load_lut(e, e->d_served.p, 1, a.nx, a.ny, a.nz, n_rows, node_offset);
}

int qm_engine_lut_download(qm_engine *e, int32_t *out) {
    if (!e || !out) return fail("NULL argument");
    if (!e->have_lut) return fail("no travel-time table resident");
    DeviceGuard guard(e->device);
    QM_HIP(copy_back(out, e->d_lut.p, (size_t)e->n_nodes * e->g.n_rows * sizeof(int32_t), e->stream));
    QM_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int qm_engine_lut_max(qm_engine *e, int32_t *max_delay) {
    if (!e || !max_delay) return fail("NULL argument");
    if (!e->have_lut) return fail("no travel-time table resident");
    *max_delay = e->lut_max;
    return 0;
}

}  // extern "C"
