// qm_engine.hip -- the engine handle and the steps (C ABI part 2, include/qmhip.h).
//
// One engine = one GPU.  It keeps the travel-time table resident together with the tables derived
// from it (qm_tables.hip), owns the scratch for partial reductions, and launches the stacking kernels
// (qm_launch_*.hip) on a caller-supplied (e.g. torch) or private HIP stream.  Nothing here falls back
// to the CPU: if HIP fails, the call fails.
#define QM_TU_STEPS 1
#include "qm_engine.hpp"

namespace {

qm::LaunchShape stack_shape(const qm_engine *e, const qm::StackArgs &a, int groups, int threads,
                            size_t lds) {
    // the grid is padded to a multiple of 8 groups (XCD-aware workgroup -> (tile, group) map)
    // (several timesteps per launch: the tile axis runs over (step, tile))
    const int steps = a.n_steps > 1 ? a.n_steps : 1;
    return {(unsigned)(steps * a.ntiles * ((groups + 7) / 8 * 8)), threads, lds, e->stream};
}

int launch_direct(qm_engine *e, const qm::StackArgs &a, int J, bool volume, int groups,
                  int threads, size_t publish_bytes) {
    const qm::LaunchShape shape = stack_shape(e, a, groups, threads, publish_bytes);
    bool built = false;
    if (volume) QM_TABLE(qm::launch_direct_volume(J, a, shape, &built));
    else QM_TABLE(qm::launch_direct_detect(J, a, shape, &built));
    if (!built) return fail("no direct stacking kernel for %d samples per lane", J);
    return 0;
}

template <int J, bool VOLUME>
int launch_stack_j(qm_engine *e, qm::StackArgs &a, int groups_lds, int groups_direct,
                   bool use_lds, bool use_direct) {
    const int KT = qm::kWave * J;
    const int threads = e->cfg_waves * qm::kWave;
    const size_t publish_bytes = (size_t)3 * e->cfg_waves * KT * sizeof(double);
    if (use_lds) {
        const size_t lds = std::max((size_t)e->cfg_lds_bytes, publish_bytes);
        a.ngroups = groups_lds;
        a.brick_list = nullptr;
        a.n_list = 0;
        const qm::LaunchShape shape = stack_shape(e, a, groups_lds, threads, lds);
        const int S = e->g.n_rows;
        bool exact = false;
        // the exact-row-count kernels: fused detect and the marginalised map for up to 64 rows,
        // volume-writing for 33-64 rows (up to 32 the paired kernel writes volumes), when the
        // launch uses the table width's own samples per lane
        if (e->cfg_exact && !e->cfg_generic && !a.accumulate && qm::exact_built(S, J)) {
            const bool j4_wide = J == 4 && S > 40;     // the second variant of 41-64 rows
            if (a.marginal != nullptr) {
                if (S <= 32) QM_TABLE(qm::launch_exact_marginal_1_32(S, a, shape, &exact));
                else if (!j4_wide) QM_TABLE(qm::launch_exact_marginal_33_64(S, a, shape, &exact));
                else QM_TABLE(qm::launch_exact_marginal_j4_41_64(S, a, shape, &exact));
            } else if (!VOLUME) {
                if (S <= 32) QM_TABLE(qm::launch_exact_detect_1_32(S, a, shape, &exact));
                else if (!j4_wide) QM_TABLE(qm::launch_exact_detect_33_64(S, a, shape, &exact));
                else QM_TABLE(qm::launch_exact_detect_j4_41_64(S, a, shape, &exact));
            } else if (S > qm::kPairMaxRows) {
                if (!j4_wide) QM_TABLE(qm::launch_exact_volume_33_64(S, a, shape, &exact));
                else QM_TABLE(qm::launch_exact_volume_j4_41_64(S, a, shape, &exact));
            }
        }
        e->last_kernel = exact ? 1 : 0;
        e->last_j = J;
        // Variants specialised on the number of 8-row offset chunks (whole-node offset prefetch;
        // detect: software-pipelined node loop) for up to 64 table rows; otherwise, and for the
        // reference's accumulate-into-volume semantics, the generic kernel.
        if (!exact) {
            int nch = (e->cfg_generic || a.accumulate) ? 0 : e->g.row_pad / 8;
            if (nch > 8) nch = 0;
            bool built = false;
            if (VOLUME) QM_TABLE(qm::launch_chunked_volume(J, nch, a, shape, &built));
            else QM_TABLE(qm::launch_chunked_detect(J, nch, a, shape, &built));
            if (!built) return fail("no chunked stacking kernel for %d samples per lane", J);
        }
        a.set0 += groups_lds;
    }
    if (use_direct) {
        a.ngroups = groups_direct;
        if (e->cfg_force_direct) {
            a.brick_list = nullptr;
            a.n_list = e->g.nbricks;
        } else {
            a.brick_list = e->d_wide.p;
            a.n_list = e->n_wide;
        }
        if (launch_direct(e, a, J, VOLUME, groups_direct, threads, publish_bytes)) return 1;
        a.set0 += groups_direct;
    }
    return 0;
}

using qm::kPairLdsBytes;
using qm::pair_jp_of;

// LDS launch over the bricks that fit the paired layout + direct launch over those that do not
template <int JP, bool VOLUME>
int launch_pair_path(qm_engine *e, qm::StackArgs &a, int groups_lds, int groups_direct,
                     bool use_lds, bool use_direct) {
    if (use_lds) {
        a.ngroups = groups_lds;
        a.brick_list = nullptr;
        a.n_list = 0;
        bool done = false;
        const qm::LaunchShape shape = stack_shape(e, a, a.ngroups, 1024, kPairLdsBytes);
        if (pair_jp_of(e->g.n_rows) == JP) {
            if (VOLUME) QM_TABLE(qm::launch_pair_volume(e->g.n_rows, a, shape, &done));
            else QM_TABLE(qm::launch_pair_detect(e->g.n_rows, a, shape, &done));
        }
        if (!done) return fail("no paired kernel built for %d rows", e->g.n_rows);
        e->last_kernel = 2;
        e->last_j = 2 * JP;
        a.set0 += groups_lds;
    }
    if (use_direct) {
        constexpr int J = 2 * JP;                      // same tile length: 64 * J = 128 * JP
        const int threads = 1024;
        const size_t publish_bytes = (size_t)3 * (threads / qm::kWave) * qm::kWave * J * sizeof(double);
        a.ngroups = groups_direct;
        a.brick_list = e->d_pwide.p;
        a.n_list = e->n_pwide;
        if (launch_direct(e, a, J, VOLUME, groups_direct, threads, publish_bytes)) return 1;
        a.set0 += groups_direct;
    }
    return 0;
}

bool pair_built(int S) { return S >= 1 && S <= qm::kPairMaxRows; }

// ---- shift-reuse layout (qm_shift.hpp) ----------------------------------------------------------
// (the layouts themselves: ShiftLayout, qm_engine.hpp; built by ensure_shift_tables, qm_tables.hip)

// does this launch take the shift-reuse kernel?  Whole 256-sample tiles plus, for what a scan leaves
// beyond them, one tail tile of 64 / 128 / 192 samples (qm_shift.hpp: shift_work); the kernels
// without tail flavours (row blocks, the 12-wave shape) and remainders of more than 192 samples pull a
// last whole tile back over its predecessor, which needs a scan of at least one tile.
bool shift_wanted(const qm_engine *e, int n_chunk, bool plain, bool volume, int64_t vol_stride) {
    if (!plain || e->cfg_shift == 0 || e->cfg_generic || e->cfg_force_direct ||
        e->user_waves || e->user_lds || e->cfg_j > 0 || e->cfg_pair == 2)
        return false;
    // volume-writing launches: the row stride goes into a 32-bit byte count (as do, for the marginal
    // map, the offsets of a group's nodes: an x-plane of fewer than 2^29 nodes)
    if (volume && vol_stride * 8 >= ((int64_t)1 << 32)) return false;
    if ((int64_t)e->g.ny * e->g.nz * 8 >= ((int64_t)1 << 32)) return false;
    return n_chunk >= 1;
}

// samples per lane of the scan's tail tile (0: none -- whole tiles only, the last one pulled back), for the
// `rest` samples a scan leaves beyond the tiles in front
int shift_tail_spl(const qm_engine *e, int rest) {
    const int rem = rest % qm::kShiftKT;
    if (rem == 0 || rem > 192 || !e->cfg_shift_tail) return 0;
    return (rem + qm::kWave - 1) / qm::kWave;
}

// Tiles of a fused detect on the wide layout: 384-sample tiles in front; what the scan leaves beyond them runs
// as ONE tail tile (<= 192 samples), ONE 256-sample tile pulled back over its predecessor (<= 256), or one more
// wide tile pulled back (qm_shift.hpp: shift_work).
void shift_wide_tiles(const qm_engine *e, int n_chunk, bool only_wide, qm::StackArgs &a) {
    int wide = n_chunk / qm::kShiftWideKT;
    const int rest = n_chunk - wide * qm::kShiftWideKT;
    a.tail_spl = 0;
    int behind = 0;
    if (rest > qm::kShiftKT || (only_wide && rest > 0)) ++wide;      // (row blocks: no other tile kinds)
    else if (rest > 0) {
        a.tail_spl = shift_tail_spl(e, rest);
        behind = 1;
    }
    a.wide_tiles = wide;
    a.ntiles = wide + behind;
}

int launch_shift_path(qm_engine *e, const ShiftLayout &L, qm::StackArgs &a, int groups_lds, int groups_direct,
                      bool use_lds, bool use_direct, int mode) {
    const bool volume = mode != qm::kShiftDetect;        // (the direct kernel's VOLUME covers the map too)
    if (use_lds) {
        a.ngroups = groups_lds;
        a.brick_list = nullptr;
        a.n_list = 0;
        qm::ShiftArgs s{};
        s.a = a;
        s.smeta = reinterpret_cast<const int4 *>(L.meta.p);
        s.stotal = L.total.p;
        s.sfit = L.fit.p;
        s.stream = reinterpret_cast<const char *>(L.stream.p);
        s.wmeta = reinterpret_cast<const int4 *>(L.wmeta.p);
        s.wtotal = L.wtotal.p;
        s.wstream = reinterpret_cast<const char *>(L.wstream.p);
        s.rows2 = L.rows2;
        s.nw = L.nw;
        s.nblk = L.nblk;
        s.sb = L.sb;
        s.stage_slots = L.stage_slots;
        s.stage_reach = L.stage_reach;
        // groups a wavefront sees before its running maximum is reset: bricks per workgroup x
        // groups per (brick, wavefront)
        // (brick maxima, tie_rule = 1: the wide tiles' loop raises them itself -- the running maximum lives as long
        // as ever --, the other tiles fold and reset it after every brick)
        const bool per_brick = a.brick_max && a.wide_tiles == 0;
        const int64_t life = (per_brick ? 1 : (int64_t)L.g.nbricks / std::max(1, a.ngroups)) *
                             std::max(1, L.g.brick_nodes / 8 / L.nw);
        s.lazy = e->cfg_shift_lazy >= 0 ? e->cfg_shift_lazy : (life >= qm::kShiftLazyGroups ? 1 : 0);
        if (L.nblk > 1 && !L.direct) s.lazy = 0;    // (the register-staged form: eager only)
        e->shift_lazy_last = s.lazy;
        e->shift_tail_last = a.tail_spl;
        e->shift_wide_last = a.wide_tiles;
        const qm::LaunchShape shape = stack_shape(e, a, a.ngroups, L.nw * qm::kWave,
                                                  qm::shift_lds_bytes(L.nw));
        const bool rows = L.nblk > 1, big = L.nw == qm::kShiftWaves8;
        if (L.wide && L.direct) QM_TABLE(qm::launch_shift_wide_rows(s, shape));   // (also a single block)
        else if (rows && L.quad && mode == qm::kShiftVolume) QM_TABLE(qm::launch_shift_rows4_volume(s, shape));
        else if (rows && L.quad) QM_TABLE(qm::launch_shift_rows4(s, shape));
        else if (rows && L.direct && mode == qm::kShiftVolume) QM_TABLE(qm::launch_shift_rows2_volume(s, shape));
        else if (rows && L.direct) QM_TABLE(qm::launch_shift_rows2(s, shape));
        else if (rows) QM_TABLE(qm::launch_shift_rows8(s, shape));
        else if (mode == qm::kShiftMarginal && big) QM_TABLE(qm::launch_shift_marginal8(s, shape));
        else if (mode == qm::kShiftMarginal) QM_TABLE(qm::launch_shift_marginal(s, shape));
        else if (mode == qm::kShiftVolume && big) QM_TABLE(qm::launch_shift_volume8(s, shape));
        else if (mode == qm::kShiftVolume) QM_TABLE(qm::launch_shift_volume(s, shape));
        else if (L.nw == qm::kShiftWaves3) QM_TABLE(qm::launch_shift_detect3(s, shape));
        else if (a.brick_max && big) QM_TABLE(qm::launch_shift_detect8_sets(s, shape));
        else if (a.brick_max) QM_TABLE(qm::launch_shift_detect_sets(s, shape));
        else if (big) QM_TABLE(qm::launch_shift_detect8(s, shape));
        else QM_TABLE(qm::launch_shift_detect(s, shape));
        e->last_kernel = 3;
        e->last_j = a.wide_tiles > 0 ? qm::kShiftWideSpl : 4;
        a.set0 += groups_lds;
    }
    if (use_direct) {
        const int threads = 512;
        const size_t publish_bytes = (size_t)3 * (threads / qm::kWave) * qm::kShiftKT * sizeof(double);
        a.ngroups = groups_direct;
        a.brick_list = L.list.p;
        a.n_list = L.n_list;
        a.tail_spl = 0;                                // (its own whole tiles of 256 samples, clamped)
        a.brick_max = nullptr;
        a.wide_tiles = 0;
        a.ntiles = (a.n_chunk + qm::kShiftKT - 1) / qm::kShiftKT;
        if (launch_direct(e, a, 4, volume, groups_direct, threads, publish_bytes)) return 1;
        a.set0 += groups_direct;
    }
    return 0;
}

}  // namespace

int auto_groups(const qm_engine *e, int ntiles, int units, int blocks_per_cu, int rounds) {
    // Workgroups all do the same amount of work, so the grid should be a whole number of
    // "rounds" over the resident slots (n_cu * blocks_per_cu): ntiles * groups <= rounds * slots,
    // as close from below as possible.  12 rounds measured best on C2/C3 (finer load balance
    // than 4; flat beyond); the partial sets stay a few tens of MB.  Never more groups than
    // there are bricks.
    const int64_t slots = (int64_t)e->n_cu * blocks_per_cu;
    int64_t want = ((int64_t)(rounds > 0 ? rounds : e->cfg_rounds) * slots) / ntiles;
    if (want < 1) want = std::max<int64_t>(1, slots / ntiles);
    want = std::max<int64_t>(1, std::min<int64_t>(want, units));
    // Group g runs on XCD g % 8 (the XCD-aware workgroup map of the stacking kernels), so a
    // group count that is not a multiple of 8 leaves some XCDs one group short: 65 groups = 9 on
    // one XCD, 8 on the others = 11 % of the step spent waiting for one XCD (C3 x 60 rows at
    // 12000 samples: 255 -> see profiles/r02_ab_runs.txt).  Round down to a multiple of 8.
    if (want > 8) want -= want % 8;
    return (int)want;
}

// Stack samples [sample0, sample0+n_chunk) of the scan.  Partials (if want_scan) land in
// e->d_pmax/d_pidx/d_psum as [*n_sets][n_chunk].
int run_stack(qm_engine *e, const double *d_onsets, int T, int fsmp, int n_samples,
              int available, int sample0, int n_chunk, double *volume, int64_t vol_stride,
              int accumulate, bool want_scan, int *n_sets, bool marginal, int m0, int m1,
              const int32_t *run_if, int n_steps, int64_t step_stride, bool *batched) {
    // n_steps > 1: that many timesteps in ONE launch (fused detect only) -- onset arrays step_stride
    // doubles apart, partial sets [*n_sets][n_steps * n_chunk].  Not every kernel can (row blocks, the
    // 12-wave shape): *batched = false then, nothing is launched and the caller goes step by step.
    // marginal: per-tile sums over the samples [m0, m1) land in e->d_marg as [*n_tiles][n_nodes]
    // (e->marg_tiles: the kernels differ in their tile length)
    const int J = run_j(e, n_chunk);
    if (plan_wide(e, J)) return 1;
    const int KT = qm::kWave * J;
    qm::StackArgs a{};
    a.g = e->g;
    a.onsets = d_onsets;
    a.lut = e->d_lut.p;
    a.rel = e->d_rel.p;
    a.brick_meta = e->d_bmeta.p;
    a.brick_total = e->d_btotal.p;
    a.T = T;
    a.fsmp = fsmp;
    a.n_samples = n_samples;
    a.sample0 = sample0;
    a.n_chunk = n_chunk;
    a.ntiles = (n_chunk + KT - 1) / KT;
    a.cap_doubles = lds_cap_doubles(e);
    // coa = exp(stack/available) = 2^(stack * log2(e)/available)
    a.z_scale = 1.4426950408889634074 / (double)available;
    a.volume = volume;
    a.vol_stride = vol_stride;
    a.accumulate = accumulate;
    a.want_scan = want_scan ? 1 : 0;
    a.set0 = 0;
    a.marginal = nullptr;
    a.m0 = m0;
    a.m1 = m1;
    a.n_nodes = e->n_nodes;
    a.run_if = run_if;

    // ---- the shift-reuse kernel (qm_shift.hpp): the fused detect's default where the table fits
    bool shift = shift_wanted(e, n_chunk, !accumulate && (volume || marginal || want_scan),
                              volume != nullptr, vol_stride);
    const int shift_mode = marginal ? qm::kShiftMarginal : volume ? qm::kShiftVolume : qm::kShiftDetect;
    // Round 6: the fused detect of a scan that holds at least one 384-sample tile runs on the WIDE layout where
    // the table has one (ShiftLayout, qm_engine.hpp): its own brick grid, the 8-wave shape
    // (automatic: only where one timestep is at least eight rounds of one-brick workgroups over the CUs and holds
    // four wide tiles -- the Icequake-sized C1, 288 bricks x 2 tiles, runs 0.49 ms on the wide layout and 0.42 on
    // the other, C3's 401-sample locate window 5.4 against 4.0: one wide tile and a short tail tile side by side
    // leave the CUs with the short one idle; the count is the single step's, so that a step's bits do not depend
    // on how many share its launch)
    ShiftLayout *L = &e->sh;
    const int64_t wide_work = (int64_t)((e->g.nx + 7) / 8) * ((e->g.ny + 7) / 8) * ((e->g.nz + 15) / 16) *
                              ((n_chunk + qm::kShiftWideKT - 1) / qm::kShiftWideKT);
    if (shift && shift_mode == qm::kShiftDetect && e->cfg_shift_wide != 0 && n_chunk >= qm::kShiftWideKT &&
        // (tie_rule = 1 WITHOUT a row of maxima per brick -- tie_sets = 0, round 5's form -- re-stacks one set of bricks per
        // sample: the wide layout's few long workgroups would publish sets of tens of thousands of nodes)
        (e->cfg_shift_wide == 1 || (wide_work >= 8 * (int64_t)e->n_cu && n_chunk >= 4 * qm::kShiftWideKT &&
                                    (!e->cfg_tie_rule || e->cfg_tie_sets) &&
                                    std::min(e->g.nx, std::min(e->g.ny, e->g.nz)) >= 8)) &&   // (thin boxes of a
                                    // rank's column partition: their 8 x 8 x 16 bricks would be mostly empty)
        e->cfg_shift_waves == 0) {
        if (ensure_shift_tables(e, e->shw)) return 1;
        if (e->shw.ok) L = &e->shw;
    }
    const bool wide = L == &e->shw;
    if (shift && !wide) {
        if (ensure_shift_tables(e, e->sh)) return 1;
        // the 12-wave shape and the register-staged row blocks are built for the fused detect only,
        // row blocks have no marginal-map flavour; none of the three has tail tiles
        const bool plain = e->sh.nblk == 1 && e->sh.nw != qm::kShiftWaves3;
        shift = e->sh.ok && (plain || (shift_mode == qm::kShiftDetect) ||
                                (shift_mode == qm::kShiftVolume && e->sh.nblk > 1 && e->sh.direct));
        a.tail_spl = (shift && plain) ? shift_tail_spl(e, n_chunk) : 0;
        // a last tile that is pulled back needs a whole tile of scan (and the detect flavours of the
        // kernels without tail tiles keep their former lower bound)
        if (shift && a.tail_spl == 0 && n_chunk % qm::kShiftKT != 0 &&
            (shift_mode == qm::kShiftDetect ? n_chunk < 192 : n_chunk < qm::kShiftKT))
            shift = false;
    }
    if (shift) {
        a.g = L->g;
        a.rel = nullptr;
        a.brick_meta = nullptr;
        a.brick_total = nullptr;
        a.ntiles = (n_chunk + qm::kShiftKT - 1) / qm::kShiftKT;
        a.cap_doubles = qm::kShiftLdsBytes / 8;
        if (wide) shift_wide_tiles(e, n_chunk, L->direct, a);        // (direct: the wide layout's row-block form)
    } else {
        a.tail_spl = 0;
    }
    // ---- the paired (16-byte operand) kernel where it applies and the shift-reuse kernel does not
    // take the launch: own brick grid and tables (built only then)
    int jp = (shift || accumulate || marginal) ? 0 : pair_jp(e, n_chunk, volume != nullptr);
    if (jp > 0) {
        if (!pair_built(e->g.n_rows)) jp = 0;
        else if (ensure_pair_tables(e, jp) != 0) return 1;   // a HIP failure while building the tables
        else if (!e->pair_ok) jp = 0;                         // the layout does not fit this table
    }
    if (jp > 0) {
        const int PKT = 128 * jp;
        a.g = e->pg;
        a.rel = e->d_prel.p;
        a.brick_meta = e->d_pmeta.p;
        a.brick_total = e->d_ptotal.p;
        a.ntiles = (n_chunk + PKT - 1) / PKT;
        a.cap_doubles = kPairLdsBytes / 8;
    }
    if (n_steps > 1) {
        const bool ok = !volume && !marginal && !accumulate && run_if == nullptr && want_scan &&
                        (!shift || (L->nblk == 1 && L->nw != qm::kShiftWaves3 && !(L->wide && L->direct)));
        if (batched) *batched = ok;
        if (!ok) return batched ? 0 : fail("run_stack: this launch cannot hold several timesteps");
        a.n_steps = n_steps;
        a.step_stride = step_stride;
    }
    const int steps = a.n_steps > 1 ? a.n_steps : 1;
    a.part_stride = (int64_t)steps * n_chunk;
    if (!shift && jp == 0) {                            // the round-2 kernels' own offsets
        if (ensure_rel(e)) return 1;
        a.rel = e->d_rel.p;
    }
    if (marginal) {
        if (e->d_marg.ensure((size_t)a.ntiles * e->n_nodes)) return 1;
        a.marginal = e->d_marg.p;
        e->marg_tiles = a.ntiles;
    }
    const int n_wide_now = shift ? L->n_list : jp > 0 ? e->n_pwide : e->n_wide;
    const int nbricks_now = shift ? L->g.nbricks : jp > 0 ? e->pg.nbricks : e->g.nbricks;
    const bool use_direct = e->cfg_force_direct || n_wide_now > 0;
    const bool use_lds = !e->cfg_force_direct && n_wide_now < nbricks_now;
    const int threads = shift ? 512 : jp > 0 ? 1024 : e->cfg_waves * qm::kWave;   // (direct launch)
    const int lds_blocks_per_cu =
        shift ? (L->nw == qm::kShiftWaves ? 2 : 1) : jp > 0 ? 1
               : std::max(1, std::min(160 * 1024 / std::max(1, e->cfg_lds_bytes), 2048 / threads));
    int groups_lds = 0, groups_direct = 0, brick_rows = 0;
    if (use_lds) {
        // (several timesteps per launch: the group count is the single step's -- the groups are the
        // order in which a sample's coalescence is summed over the nodes, and a step's result must
        // not depend on how many steps share its launch; the extra steps only make the grid longer)
        // (bricks of <= 64 nodes -- coarse grids with wide tables, 4 nodes per wavefront and brick: a
        // workgroup's fixed costs weigh more than the grid's tail, three rounds instead of twelve:
        // E2 0.418 -> 0.394 ms, profiles/r04_rounds_sweep.txt)
        int rounds = (!shift && jp == 0 && !e->user_rounds && e->g.brick_nodes <= 64) ? 3 : 0;
        // (the wide layout's 8-wave workgroups, one per CU: FEW, long workgroups -- C3 41.7 / 42.1 / 42.3 / 42.6 ms
        // at 1 / 4 / 8 / 12 rounds, where the 4-wave shape runs 50.1 / 46.5 / 45.7 at 4 / 8 / 12,
        // profiles/r06_ab_runs.txt: the fewest rounds that leave no slot idle, 0.2 % charged per round)
        // (the 8-wave shape of 33-64 rows likewise: a C4 slab, 47 tiles, 170.2 ms at 12 rounds = 64 groups, 168.6 at
        // 16 groups = 2.94 rounds, 181.5 at 40 groups = 7.3 rounds: what the count must avoid is a last round that is
        // mostly empty)
        if (shift && (wide || (L->nw == qm::kShiftWaves8 && L->nblk == 1)) && !e->user_rounds) {
            double best = 1e300;
            for (int r = 1; r <= e->cfg_rounds; ++r) {
                const int g = auto_groups(e, a.ntiles, nbricks_now, lds_blocks_per_cu, r);
                const double busy = (double)a.ntiles * g / ((double)r * e->n_cu * lds_blocks_per_cu);
                const double cost = (1.0 - std::min(1.0, busy)) + 0.002 * r;
                if (cost < best - 1e-9) { best = cost; rounds = r; }
            }
        }
        // (tie_rule = 1 stacks one SET of bricks again per sample, qm_ties.hpp: eight times as many,
        // smaller sets -- the refinement's cost falls with the set size, the stacking launch loses ~1 %)
        groups_lds = e->cfg_groups > 0 ? std::min(e->cfg_groups, nbricks_now)
                                       : auto_groups(e, a.ntiles, nbricks_now, lds_blocks_per_cu, rounds);
        // Round 6: the shift-reuse fused detect leaves the largest z PER BRICK and sample beside its workgroups'
        // partial sets (StackArgs::brick_max): the refinement stacks one brick per sample, the launch keeps its
        // group count -- and its bits
        brick_rows = (e->cfg_tie_rule && want_scan && e->cfg_tie_sets && shift && shift_mode == qm::kShiftDetect &&
                      L->nblk == 1 && L->nw != qm::kShiftWaves3 && !(L->wide && L->direct)) ? nbricks_now : 0;
        if (e->cfg_tie_rule && want_scan && !brick_rows && !e->user_rounds && e->cfg_groups == 0) {
            // ... the other kernels: eight times as many, smaller sets, but never sets of fewer than four bricks:
            // the workgroup's fixed costs (C2: +8 % on the stacking launch at one brick per set)
            const int fine = auto_groups(e, a.ntiles, std::max(1, nbricks_now / 4), lds_blocks_per_cu,
                                         8 * (rounds > 0 ? rounds : e->cfg_rounds));
            groups_lds = std::max(groups_lds, fine);
        }
    }
    if (use_direct) {
        const int units = e->cfg_force_direct ? nbricks_now : n_wide_now;
        groups_direct = e->cfg_groups > 0 ? std::min(e->cfg_groups, units)
                                          : auto_groups(e, a.ntiles, units, 2048 / threads);
    }
    const int sets = groups_lds + groups_direct;
    if (want_scan) {
        const size_t need = (size_t)sets * steps * n_chunk;
        if (e->d_pmax.ensure(need) || e->d_psum.ensure(need) || e->d_pidx.ensure(need)) return 1;
        if (brick_rows && e->d_bmax.ensure((size_t)brick_rows * steps * n_chunk)) return 1;
    }
    a.brick_max = brick_rows ? e->d_bmax.p : nullptr;
    if (brick_rows && a.wide_tiles > 0) {
        // (the wide tiles' loop raises its rows by atomic maxima, and only where a group comes near the running
        // maximum: they start at -inf; the other tiles write theirs whole)
        const size_t n = (size_t)brick_rows * steps * n_chunk;
        hipLaunchKernelGGL(qm::fill_kernel, dim3((unsigned)std::min<size_t>((n + 1023) / 1024, 65535)), dim3(256), 0,
                           e->stream, e->d_bmax.p, n, -std::numeric_limits<double>::infinity());
        QM_HIP(hipGetLastError());
    }
    a.part_max = e->d_pmax.p;
    a.part_idx = e->d_pidx.p;
    a.part_sum = e->d_psum.p;
    *n_sets = sets;
    e->last_sets = sets;
    e->last_scan_n = steps * n_chunk;
    e->last_g = a.g;
    e->last_groups_lds = groups_lds;
    e->last_brick_rows = brick_rows;
    e->last_groups_direct = groups_direct;
    e->last_list = !use_direct || e->cfg_force_direct ? nullptr
                   : shift ? L->list.p : jp > 0 ? e->d_pwide.p : e->d_wide.p;
    e->last_n_list = !use_direct ? 0 : e->cfg_force_direct ? nbricks_now : n_wide_now;

    // a conditional launch (the fallback of a screened step) is not part of the timing log: it
    // returns at once unless the step has to be redone
    const bool logged = run_if == nullptr;
    hipEvent_t ev_begin = e->ev0, ev_end = e->ev1;
    if (e->log_timing && logged) {
        if (e->ev_used + 2 > e->ev_log.size()) {
            for (int i = 0; i < 2; ++i) {
                hipEvent_t ev;
                QM_HIP(hipEventCreate(&ev));
                e->ev_log.push_back(ev);
            }
        }
        ev_begin = e->ev_log[e->ev_used];
        ev_end = e->ev_log[e->ev_used + 1];
        e->ev_used += 2;
    }
    if (logged) QM_HIP(hipEventRecord(ev_begin, e->stream));
    int rc = 0;
#define QM_LAUNCH(JJ)                                                                        \
    rc = (volume || marginal)                                                                 \
             ? launch_stack_j<JJ, true>(e, a, groups_lds, groups_direct, use_lds, use_direct)  \
                : launch_stack_j<JJ, false>(e, a, groups_lds, groups_direct, use_lds, use_direct)
    if (shift)
        rc = launch_shift_path(e, *L, a, groups_lds, groups_direct, use_lds, use_direct, shift_mode);
    else if (jp == 2)
        rc = volume ? launch_pair_path<2, true>(e, a, groups_lds, groups_direct, use_lds, use_direct)
                    : launch_pair_path<2, false>(e, a, groups_lds, groups_direct, use_lds, use_direct);
    else if (jp == 1)
        rc = volume ? launch_pair_path<1, true>(e, a, groups_lds, groups_direct, use_lds, use_direct)
                    : launch_pair_path<1, false>(e, a, groups_lds, groups_direct, use_lds, use_direct);
    else switch (J) {
        case 1: QM_LAUNCH(1); break;
        case 2: QM_LAUNCH(2); break;
        case 4: QM_LAUNCH(4); break;
        default: return fail("samples_per_lane must be 1, 2 or 4 (got %d)", J);
    }
#undef QM_LAUNCH
    if (rc) return rc;
    if (logged) {
        QM_HIP(hipEventRecord(ev_end, e->stream));
        e->timed = !e->log_timing;
    }
    return 0;
}

int combine(qm_engine *e, const double *pmax, const int64_t *pidx, const double *psum, int sets,
            int n, int mode, int64_t node_offset, int64_t n_nodes_total, double *o_max,
            double *o_second, int64_t *o_idx, const int32_t *run_if, int64_t set_stride) {
    // (tie_rule = 1 with a row of maxima per brick: the fold of the workgroups' own sets also leaves the largest z)
    double *o_z = nullptr;
    if (e->cfg_tie_rule && e->last_brick_rows > 0 && pmax == e->d_pmax.p && run_if == nullptr) {
        if (e->d_tie_zext.ensure(n)) return 1;
        o_z = e->d_tie_zext.p;
    }
    hipLaunchKernelGGL(qm::combine_kernel, dim3((n + qm::kWave - 1) / qm::kWave),
                       dim3(qm::kCombineWaves * qm::kWave), 0,
                       e->stream, pmax, pidx, psum, sets, n, set_stride > 0 ? set_stride : (int64_t)n,
                       mode, node_offset, (double)n_nodes_total, o_max, o_second, o_idx, run_if, o_z);
    QM_HIP(hipGetLastError());
    return 0;
}

// Detect-type stacking of the whole scan plus the combine of its partial sets into the three
// output series (mode as combine_kernel).  With screening: the screened result is combined first;
// then the float64 kernel and its combine are enqueued conditionally on the step's flag word, so a
// step that could not be screened (too many candidate cells, non-finite onsets) is redone on the
// device without the host ever waiting -- those launches return at once otherwise.
int detect_core(qm_engine *e, const double *d_on, int T, int fsmp, int ns, int available, int mode,
                int64_t n_nodes_total, double *o_max, double *o_second, int64_t *o_idx) {
    int sets = 0;
    bool screened = false;
    if (run_screen(e, d_on, T, fsmp, ns, available, &sets, &screened)) return 1;
    const int32_t *run_if = nullptr;
    if (screened) {
        if (combine(e, e->d_pmax.p, e->d_pidx.p, e->d_psum.p, sets, ns, mode, e->node_offset,
                    n_nodes_total, o_max, o_second, o_idx))
            return 1;
        run_if = e->d_flags.p;
    }
    if (run_stack(e, d_on, T, fsmp, ns, available, 0, ns, nullptr, 0, 0, true, &sets, false, 0, 0,
                  run_if))
        return 1;
    e->last_sets_own = !screened;                       // (d_pmax holds the float64 launch's sets for certain)
    if (combine(e, e->d_pmax.p, e->d_pidx.p, e->d_psum.p, sets, ns, mode, e->node_offset,
                n_nodes_total, o_max, o_second, o_idx, run_if))
        return 1;
    // (the reference's rule on near-ties: final series of this engine's own nodes only -- the partial
    // sets of a sharded detect leave the engine before anyone knows the global maximum)
    if (e->cfg_tie_rule && mode == 1 && !screened)
        return refine_ties(e, d_on, T, fsmp, available, 0, ns, sets, o_idx);
    return 0;
}

// tie_rule = 1 (qm_ties.hpp): the index series of the launch whose partial sets lie in d_pmax, refined.
// n_steps > 1: the launch held several timesteps (sets [sets][n_steps * n_chunk], step k's onsets step_stride
// doubles behind step 0's).  zext / o_key: a sharded detect's engine -- its sets against the grid's maxima, the
// outcome exported (tie_export_kernel) instead of applied.
int refine_ties(qm_engine *e, const double *d_on, int T, int fsmp, int available, int sample0,
                int n_chunk, int sets, int64_t *o_idx, int n_steps, int64_t step_stride,
                const double *zext, unsigned long long *o_key) {
    const int n = n_chunk * std::max(1, n_steps);
    if (n > INT32_MAX / (2 * qm::kTieMaxSets)) return fail("tie_rule: %d samples in one launch are too many", n);
    const int max_pairs = qm::kTieMaxSets * n;
    const int max_cands = 4 * n + 65536;
    if (e->d_tie_z.ensure(n) || e->d_tie_pairs.ensure(2 * (size_t)max_pairs) || e->d_tie_imin.ensure(n) ||
        e->d_tie_count.ensure(4) || e->d_tie_emax.ensure(n) || e->d_tie_cands.ensure(2 * (size_t)max_cands) ||
        e->d_tie_keys.ensure(max_cands))
        return 1;
    hipStream_t s = e->stream;
    QM_HIP(hipMemsetAsync(e->d_tie_count.p, 0, 4 * sizeof(int32_t), s));
    int2 *pairs = reinterpret_cast<int2 *>(e->d_tie_pairs.p);
    // (brick maxima: the sample's largest z is already in the workgroups' few partial sets; the per-brick rows are
    // read once, for the candidates)
    // (... and the sample's largest z came out of the combine that preceded this call)
    const int rows = e->last_brick_rows;
    if (rows > 0 && !zext) zext = e->d_tie_zext.p;
    hipLaunchKernelGGL(qm::tie_pairs_kernel, dim3((n + 63) / 64), dim3(64, qm::kTieSetLanes), 0, s,
                       rows > 0 ? (const double *)e->d_bmax.p : (const double *)e->d_pmax.p, rows > 0 ? rows : sets,
                       (const double *)e->d_pmax.p + (int64_t)e->last_groups_lds * n,
                       rows > 0 ? e->last_groups_direct : 0, n, (int64_t)n, e->d_tie_z.p, pairs,
                       e->d_tie_count.p, max_pairs, e->d_tie_emax.p, e->d_tie_imin.p, e->d_tie_count.p + 1, zext);
    QM_HIP(hipGetLastError());
    // How many pairs there are is known on the device only, and nobody waits for it: the evaluation is
    // launched for what a generic step has (one pair per sample) with room to spare; workgroups beyond the
    // list return at once, a longer list is covered by the grid's stride.  The
    // counters travel to the host when somebody asks (qm_engine_get "tie_pairs", "tie_overflow_samples").
    ++e->tie_refined_steps;
    e->tie_counts_pending = true;
    qm::TieArgs a{};
    a.g = e->last_g;
    a.onsets = d_on;
    a.lut = e->d_lut.p;
    a.T = T; a.fsmp = fsmp; a.sample0 = sample0; a.n_chunk = n;
    a.ns_step = n_steps > 1 ? n_chunk : 0;
    a.step_stride = step_stride;
    a.z_scale = 1.4426950408889634074 / (double)available;
    a.recip = 1.0 / (double)available;
    a.groups_lds = e->last_groups_lds;
    a.groups_direct = e->last_groups_direct;
    a.brick_rows = rows;
    a.brick_list = e->last_list;
    a.n_list = e->last_n_list;
    // workgroups per pair: ~2048 nodes each
    const int64_t per_set = (int64_t)e->n_nodes / std::max(1, e->last_groups_lds + e->last_groups_direct);
    a.chunks = rows > 0 ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(64, per_set / 2048));   // (rows: a brick)
    a.zbest = e->d_tie_z.p;
    a.pairs = pairs;
    a.n_pairs = e->d_tie_count.p;
    a.emax = e->d_tie_emax.p;
    a.imin = e->d_tie_imin.p;
    a.cands = reinterpret_cast<int2 *>(e->d_tie_cands.p);
    a.cand_keys = e->d_tie_keys.p;
    a.n_cands = e->d_tie_count.p + 2;
    a.max_cands = max_cands;
    const unsigned grid = (unsigned)((int64_t)(n + 1024) * a.chunks);
    hipLaunchKernelGGL(qm::tie_eval_kernel<0>, dim3(grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(qm::tie_pick_kernel, dim3(256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(qm::tie_eval_kernel<1>, dim3(grid), dim3(256), 0, s, a);   // (returns at once unless the list overflowed)
    if (o_key)
        hipLaunchKernelGGL(qm::tie_export_kernel, dim3((n + 255) / 256), dim3(256), 0, s,
                           (const unsigned long long *)e->d_tie_emax.p, (const int32_t *)e->d_tie_imin.p, n,
                           e->node_offset, o_key, o_idx);
    else
        hipLaunchKernelGGL(qm::tie_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, s,
                           (const int32_t *)e->d_tie_imin.p, n, e->node_offset, o_idx + sample0);
    QM_HIP(hipGetLastError());
    return 0;
}

int check_step(qm_engine *e, int T, int fsmp, int lsmp, int available, int *n_samples) {
    if (!e->have_lut) return fail("no travel-time table resident: call qm_engine_load_lut first");
    if (fsmp < 0 || lsmp < 0) return fail("negative pad (fsmp=%d, lsmp=%d)", fsmp, lsmp);
    const int ns = T - fsmp - lsmp;
    if (ns <= 0) return fail("no samples to scan: T=%d fsmp=%d lsmp=%d", T, fsmp, lsmp);
    if (available <= 0) return fail("available must be positive (got %d)", available);
    if (e->lut_max > lsmp)
        return fail("largest travel time (%d samples) exceeds the post-pad lsmp=%d: the scan "
                    "would read past the onset rows (undefined behaviour in the reference)",
                    e->lut_max, lsmp);
    *n_samples = ns;
    return 0;
}

// device-resident copy of the onsets (or the caller's device pointer)
int stage_onsets(qm_engine *e, const double *onsets, int on_device, int T, const double **out) {
    if (on_device) {
        *out = onsets;
        return 0;
    }
    const size_t n = (size_t)e->g.n_rows * T;
    if (e->d_onsets.ensure(n)) return 1;
    QM_HIP(copy_in(e->d_onsets.p, onsets, n * sizeof(double), e->stream));
    *out = e->d_onsets.p;
    return 0;
}

int stage_out(qm_engine *e, int n, int out_on_device, double *max_coa, double *max_norm,
              int64_t *idx, OutStage *st) {
    if (out_on_device) {
        *st = OutStage{max_coa, max_norm, idx};
        return 0;
    }
    // (the three series back to back in ONE buffer: they travel to the host as one copy, fetch_out)
    if (e->d_out_a.ensure(3 * (size_t)n)) return 1;
    *st = OutStage{e->d_out_a.p, e->d_out_a.p + n, reinterpret_cast<int64_t *>(e->d_out_a.p + 2 * (size_t)n)};
    return 0;
}

int fetch_out(qm_engine *e, int n, int out_on_device, const OutStage &st, double *max_coa,
              double *max_norm, int64_t *idx) {
    if (out_on_device) return 0;
    // one DMA of the packed [3][n] buffer, three CPU copies out of the pinned half (round 4: three
    // DMAs with a wait each -- two round trips more per host-array call)
    void *dst[3] = {max_coa, max_norm, idx};
    QM_HIP(copy_back_pieces(dst, st.a, 3, (size_t)n * sizeof(double), e->stream));
    return 0;
}

// ------------------------------------------------------------------------------- C ABI
extern "C" {

int qm_engine_create(int device_id, qm_engine **out) {
    if (!out) return fail("qm_engine_create: out is NULL");
    int n = 0;
    QM_HIP(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n)
        return fail("device %d not available (%d HIP devices visible)", device_id, n);
    DeviceGuard guard(device_id);
    qm_engine *e = new qm_engine();
    e->device = device_id;
    hipDeviceProp_t prop;
    QM_HIP(hipGetDeviceProperties(&prop, device_id));
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    QM_HIP(acquire_stream(device_id, &e->own_stream));
    e->stream = e->own_stream;
    QM_HIP(hipEventCreate(&e->ev0));
    QM_HIP(hipEventCreate(&e->ev1));
    *out = e;
    return 0;
}

void qm_engine_destroy(qm_engine *e) {
    if (!e) return;
    DeviceGuard guard(e->device);
    (void)hipStreamSynchronize(e->stream);
    streams_orphan(e);
    PoolReleaseScope one_wait;                          // every buffer of the engine behind ONE device-wide wait
    e->release_all();
    for (TableSlot &slot : e->slots) slot.state.release_all();
    e->d_grids.release(); e->d_rows.release(); e->d_served.release();
    e->d_sig.release(); e->d_sta.release(); e->d_lta.release(); e->d_raw.release();
    e->d_onset_meta.release(); e->d_scalar.release();
    e->d_onsets.release(); e->d_pmax.release(); e->d_psum.release(); e->d_out_a.release();
    e->d_chunk.release(); e->d_marg.release(); e->d_marg_out.release(); e->d_pidx.release();
    e->d_fit_a.release(); e->d_fit_b.release(); e->d_fit_c.release(); e->d_fit_part.release();
    e->d_fit_val.release(); e->d_fit_win.release(); e->d_fit_pidx.release();
    e->d_counts.release(); e->d_cells.release(); e->d_work.release(); e->d_flags.release();
    e->d_onq.release(); e->d_sparams.release();
    e->d_cell.release(); e->d_gmax.release(); e->d_pm.release(); e->d_rowmax.release(); e->d_ssum.release();
    e->d_cand_z.release(); e->d_cand_idx.release();
    e->d_bmax.release(); e->d_tie_zext.release(); e->d_tie_zgrid.release(); e->d_tie_z.release(); e->d_tie_pairs.release(); e->d_tie_imin.release(); e->d_tie_count.release();
    e->d_tie_emax.release(); e->d_tie_cands.release(); e->d_tie_keys.release();
    if (e->h_flags) (void)hipHostFree(e->h_flags);
    for (hipEvent_t ev : e->ev_log) (void)hipEventDestroy(ev);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->own_stream) {
        (void)hipStreamSynchronize(e->own_stream);
        park_stream(e->device, e->own_stream);
    }
    delete e;
}

int qm_engine_set_stream(qm_engine *e, void *hip_stream, int use_own) {
    if (!e) return fail("engine is NULL");
    // NULL with use_own == 0 is the device's default (null) stream -- what
    // torch.cuda.current_stream().cuda_stream is unless a stream context is active
    {
        DeviceGuard guard(e->device);
        if (drain_flags(e)) return 1;       // per-step outcomes still travelling on the old stream
    }
    e->stream = use_own ? e->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
    return 0;
}

int qm_engine_synchronize(qm_engine *e) {
    if (!e) return fail("engine is NULL");
    DeviceGuard guard(e->device);
    QM_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int qm_engine_config(qm_engine *e, const char *key, int64_t v) {
    if (!e || !key) return fail("qm_engine_config: NULL argument");
    const std::string k(key);
    if (k == "brick_x" || k == "brick_y" || k == "brick_z") {
        if (v < 0 || v > 64) return fail("%s must be in 0..64 (0 = automatic)", key);
        (k == "brick_x" ? e->cfg_bx : k == "brick_y" ? e->cfg_by : e->cfg_bz) = (int)v;
        if (e->cfg_bx > 0) {                           // an explicit shape needs all three
            if (e->cfg_by < 1) e->cfg_by = 1;
            if (e->cfg_bz < 1) e->cfg_bz = 1;
        }
    } else if (k == "samples_per_lane") {
        if (v != 0 && v != 1 && v != 2 && v != 4)
            return fail("samples_per_lane must be 0 (automatic), 1, 2 or 4");
        e->cfg_j = (int)v;
    } else if (k == "waves") {
        if (v < 1 || v > 16) return fail("waves must be 1..16");
        e->cfg_waves = (int)v;
        e->user_waves = true;
    } else if (k == "groups") {
        if (v < 0) return fail("groups must be >= 0");
        e->cfg_groups = (int)v;
    } else if (k == "rounds") {
        if (v < 1 || v > 1024) return fail("rounds must be in 1..1024");
        e->cfg_rounds = (int)v;
        e->user_rounds = true;
    } else if (k == "lds_bytes") {
        if (v < 1024 || v > 160 * 1024) return fail("lds_bytes must be in 1 KiB..160 KiB");
        e->cfg_lds_bytes = (int)(v / 16 * 16);
        e->user_lds = true;
    } else if (k == "force_direct") {
        e->cfg_force_direct = v ? 1 : 0;
    } else if (k == "generic") {
        e->cfg_generic = v ? 1 : 0;
    } else if (k == "exact") {
        e->cfg_exact = v ? 1 : 0;
    } else if (k == "scan_waves") {
        if (v < 1 || v > 4096) return fail("scan_waves must be in 1..4096");
        e->cfg_scan_waves = (int)v;

    } else if (k == "pair") {
        if (v < 0 || v > 2) return fail("pair must be 0 (off), 1 (automatic) or 2 (any scan length)");
        e->cfg_pair = (int)v;
    } else if (k == "shift") {
        if (v < -1 || v > 1) return fail("shift must be -1 (automatic), 0 (off) or 1");
        e->cfg_shift = (int)v;
    } else if (k == "shift_waves") {
        if (v != 0 && v != qm::kShiftWaves && v != qm::kShiftWaves8 && v != qm::kShiftWaves3)
            return fail("shift_waves must be 0 (automatic), 4, 8 or 12");
        e->cfg_shift_waves = (int)v;
        e->sh.built = false;
    } else if (k == "shift_rows_direct") {
        if (v < 0 || v > 2) return fail("shift_rows_direct must be 0, 1 or 2");
        e->cfg_shift_rows_direct = (int)v;
        e->sh.built = false;
    } else if (k == "shift_tail") {
        e->cfg_shift_tail = v ? 1 : 0;
    } else if (k == "shift_wide") {
        if (v < -1 || v > 1) return fail("shift_wide must be -1 (automatic), 0 (off) or 1");
        e->cfg_shift_wide = (int)v;
    } else if (k == "shift_wide_rows") {
        if (v < 0 || v > 2) return fail("shift_wide_rows must be 0 (never), 1 (where all rows do not fit) or 2 (always)");
        e->cfg_shift_wide_rows = (int)v;
        e->shw.built = false;
    } else if (k == "tie_rule") {
        if (v != 0 && v != 1) return fail("tie_rule must be 0 (largest sum) or 1 (the reference's exp rule)");
        e->cfg_tie_rule = (int)v;
    } else if (k == "tie_sets") {
        e->cfg_tie_sets = v ? 1 : 0;
    } else if (k == "shift_lazy") {
        if (v < -1 || v > 1) return fail("shift_lazy must be -1 (automatic), 0 or 1");
        e->cfg_shift_lazy = (int)v;
    } else if (k == "screen") {
        e->cfg_screen = v ? 1 : 0;
    } else if (k == "screen_pairs") {
        if (v != 0 && v != 1 && v != 2 && v != 4) return fail("screen_pairs must be 0, 1, 2 or 4");
        e->cfg_screen_pairs = (int)v;
        e->screen_kt = 0;
    } else if (k == "screen_brick16") {
        e->cfg_screen_brick16 = v ? 1 : 0;
        e->screen_kt = 0;
    } else if (k == "screen_big") {
        if (v < -1 || v > 1) return fail("screen_big must be -1 (automatic), 0 or 1");
        e->cfg_screen_big = (int)v;
        e->screen_kt = 0;
    } else if (k == "stream_pull") {
        if (v < -1 || v > 1) return fail("stream_pull must be -1 (slots of <= 1 MB), 0 (never) or 1 (always)");
        e->cfg_stream_pull = (int)v;
    } else if (k == "stream_stamps") {
        e->cfg_stream_stamps = v ? 1 : 0;
    } else if (k == "log_timing") {
        e->log_timing = v != 0;
        e->ev_used = 0;
    } else if (k == "chunk_bytes") {
        if (v < (1 << 20)) return fail("chunk_bytes must be >= 1 MiB");
        e->cfg_chunk_bytes = v;
    } else {
        return fail("unknown config key '%s'", key);
    }
    return 0;
}

int qm_engine_get(qm_engine *e, const char *key, int64_t *v) {
    if (!e || !key || !v) return fail("qm_engine_get: NULL argument");
    const std::string k(key);
    if (k == "brick_x") *v = e->have_lut ? e->g.bx : e->cfg_bx;
    else if (k == "brick_y") *v = e->have_lut ? e->g.by : e->cfg_by;
    else if (k == "brick_z") *v = e->have_lut ? e->g.bz : e->cfg_bz;
    else if (k == "samples_per_lane") *v = e->have_lut ? eff_j(e) : e->cfg_j;
    else if (k == "waves") *v = e->cfg_waves;
    else if (k == "groups") *v = e->cfg_groups;
    else if (k == "lds_bytes") *v = e->cfg_lds_bytes;
    else if (k == "force_direct") *v = e->cfg_force_direct;
    else if (k == "chunk_bytes") *v = e->cfg_chunk_bytes;
    else if (k == "screen") *v = e->cfg_screen;
    else if (k == "screened_steps" || k == "fallback_steps" || k == "last_candidates") {
        DeviceGuard guard(e->device);
        if (drain_flags(e)) return 1;
        *v = k == "screened_steps" ? e->screened_steps
             : k == "fallback_steps" ? e->fallback_steps : e->last_candidates;
    }
    else if (k == "screen_pairs") *v = e->last_plan_jp;
    else if (k == "screen_big") *v = e->last_plan_big;
    else if (k == "screen_brick_nodes") *v = e->sg.brick_nodes;
    else if (k == "last_kernel") *v = e->last_kernel;
    else if (k == "last_kernel_j") *v = e->last_j;
    else if (k == "shift") *v = e->cfg_shift;
    else if (k == "shift_ok") *v = e->sh.built && e->sh.ok ? 1 : 0;
    else if (k == "shift_waves")                        // (of the layout the last shift-reuse launch ran on)
        *v = e->shift_wide_last > 0 && e->shw.ok ? e->shw.nw : e->sh.ok ? e->sh.nw : e->cfg_shift_waves;
    else if (k == "shift_lazy") *v = e->shift_lazy_last;
    else if (k == "shift_tail") *v = e->cfg_shift_tail;
    else if (k == "shift_tail_spl") *v = e->shift_tail_last;
    else if (k == "shift_wide") *v = e->cfg_shift_wide;
    else if (k == "shift_wide_ok") *v = e->shw.built && e->shw.ok ? 1 : 0;
    else if (k == "shift_wide_rows") *v = e->cfg_shift_wide_rows;
    else if (k == "shift_wide_row_blocks") *v = e->shw.ok ? e->shw.nblk : 0;
    else if (k == "shift_wide_tiles") *v = e->shift_wide_last;
    else if (k == "shift_wide_brick_nodes") *v = e->shw.ok ? e->shw.g.brick_nodes : 0;
    else if (k == "shift_wide_direct_bricks") *v = e->shw.ok ? e->shw.n_list : 0;
    else if (k == "shift_wide_operands_per_add_x1000")   // 8-byte LDS operands fetched per add of a wide tile (x 1000)
        *v = e->shw.ok && e->shw.group_rows > 0 ? (e->shw.wquads * 4 * 1000) / (e->shw.group_rows * 48) : 0;
    else if (k == "steps_per_launch") *v = e->last_batched;
    else if (k == "tie_rule") *v = e->cfg_tie_rule;
    else if (k == "tie_sets") *v = e->cfg_tie_sets;
    else if (k == "stream_pull") *v = e->cfg_stream_pull;
    else if (k == "tie_brick_rows") *v = e->last_brick_rows;

    else if (k == "tie_refined_steps") *v = e->tie_refined_steps;
    else if (k == "tie_overflow_samples" || k == "tie_pairs") {
        if (e->tie_counts_pending) {                       // (the last refinement's counters: now)
            DeviceGuard guard(e->device);
            int32_t h[2] = {0, 0};
            QM_HIP(copy_back(h, e->d_tie_count.p, sizeof(h), e->stream));
            e->tie_pairs_last = h[0];
            e->tie_overflow_last = h[1];
            e->tie_counts_pending = false;
        }
        *v = k == "tie_pairs" ? e->tie_pairs_last : e->tie_overflow_last;
    }
    else if (k == "table_hits") *v = e->table_hits;
    else if (k == "table_misses") *v = e->table_misses;
    else if (k == "table_evictions") *v = e->table_evictions;
    else if (k == "tables_parked") {
        *v = 0;
        for (const TableSlot &sl : e->slots) *v += sl.used ? 1 : 0;
    } else if (k == "table_bytes") {                     // device bytes of the resident table's state
        *v = (int64_t)e->device_bytes();
    } else if (k == "tables_parked_bytes") {
        *v = 0;
        for (const TableSlot &sl : e->slots) *v += sl.used ? (int64_t)sl.state.device_bytes() : 0;
    }
    else if (k == "shift_row_blocks") *v = e->sh.ok ? e->sh.nblk : 0;
    else if (k == "shift_brick_nodes") *v = e->sh.ok ? e->sh.g.brick_nodes : 0;
    else if (k == "shift_wide_bricks") *v = e->sh.ok ? e->sh.n_list : 0;
    else if (k == "shift_operands_per_add_x1000")   // 8-byte LDS operands fetched per add (x 1000)
        *v = e->sh.ok && e->sh.group_rows > 0
                 ? (e->sh.quads * 4 * 1000) / (e->sh.group_rows * 32) : 0;
    else if (k == "pair_brick_nodes") *v = e->pair_kt ? e->pg.brick_nodes : 0;
    else if (k == "pair_wide_bricks") *v = e->pair_kt ? e->n_pwide : 0;
    else if (k == "pair_tile") *v = e->pair_ok ? e->pair_kt : 0;
    else if (k == "n_bricks") *v = e->g.nbricks;
    else if (k == "n_wide_bricks") {
        if (e->have_lut) {
            DeviceGuard guard(e->device);
            if (plan_wide(e, eff_j(e))) return 1;
        }
        *v = e->n_wide;
    } else if (k == "mean_span") {                     // mean delay span per (brick, row), samples
        int64_t sum = 0;
        for (int32_t t : e->h_btotal) sum += t;
        *v = e->h_btotal.empty() ? 0 : sum / ((int64_t)e->h_btotal.size() * std::max(1, e->g.n_rows));
    } else if (k == "n_cu") *v = e->n_cu;
    else if (k == "n_nodes") *v = e->n_nodes;
    else if (k == "n_rows") *v = e->g.n_rows;
    else if (k == "nx") *v = e->g.nx;
    else if (k == "ny") *v = e->g.ny;
    else if (k == "nz") *v = e->g.nz;
    else return fail("unknown key '%s'", key);
    return 0;
}

int qm_engine_detect_partial(qm_engine *e, const double *log_onsets, int onsets_on_device,
                             int32_t T, int32_t fsmp, int32_t lsmp, int32_t available,
                             double *d_part_max, int64_t *d_part_idx, double *d_part_sum) {
    if (!e || !log_onsets || !d_part_max || !d_part_idx || !d_part_sum)
        return fail("qm_engine_detect_partial: NULL argument");
    DeviceGuard guard(e->device);
    int ns = 0;
    if (check_step(e, T, fsmp, lsmp, available, &ns)) return 1;
    const double *d_on = nullptr;
    if (stage_onsets(e, log_onsets, onsets_on_device, T, &d_on)) return 1;
    return detect_core(e, d_on, T, fsmp, ns, available, 0, 0, d_part_max, d_part_sum, d_part_idx);
}

int qm_engine_finalize(qm_engine *e, const double *d_part_max, const int64_t *d_part_idx,
                       const double *d_part_sum, int32_t n_sets, int32_t n_samples,
                       int64_t n_nodes_total, double *max_coa, double *max_norm_coa,
                       int64_t *max_coa_idx, int out_on_device) {
    if (!e || !d_part_max || !d_part_idx || !d_part_sum || !max_coa || !max_norm_coa ||
        !max_coa_idx)
        return fail("qm_engine_finalize: NULL argument");
    if (n_sets < 1 || n_samples < 1) return fail("qm_engine_finalize: empty input");
    DeviceGuard guard(e->device);
    OutStage st;
    if (stage_out(e, n_samples, out_on_device, max_coa, max_norm_coa, max_coa_idx, &st)) return 1;
    if (combine(e, d_part_max, d_part_idx, d_part_sum, n_sets, n_samples, 1, 0, n_nodes_total,
                st.a, st.b, st.i))
        return 1;
    return fetch_out(e, n_samples, out_on_device, st, max_coa, max_norm_coa, max_coa_idx);
}

int qm_engine_finalize_packed(qm_engine *e, const double *d_packed, int32_t n_sets,
                              int32_t n_samples, int64_t n_nodes_total, double *max_coa,
                              double *max_norm_coa, int64_t *max_coa_idx, int out_on_device) {
    if (!e || !d_packed || !max_coa || !max_norm_coa || !max_coa_idx)
        return fail("qm_engine_finalize_packed: NULL argument");
    if (n_sets < 1 || n_samples < 1) return fail("qm_engine_finalize_packed: empty input");
    DeviceGuard guard(e->device);
    OutStage st;
    if (stage_out(e, n_samples, out_on_device, max_coa, max_norm_coa, max_coa_idx, &st)) return 1;
    // set s = rows [s][0] (maxima), [s][1] (indices, int64 bits), [s][2] (sums) of [n_sets][3][n]
    if (combine(e, d_packed, reinterpret_cast<const int64_t *>(d_packed + n_samples),
                d_packed + 2 * (int64_t)n_samples, n_sets, n_samples, 1, 0, n_nodes_total, st.a,
                st.b, st.i, nullptr, 3 * (int64_t)n_samples))
        return 1;
    return fetch_out(e, n_samples, out_on_device, st, max_coa, max_norm_coa, max_coa_idx);
}

int qm_engine_detect(qm_engine *e, const double *log_onsets, int onsets_on_device, int32_t T,
                     int32_t fsmp, int32_t lsmp, int32_t available, int64_t n_nodes_total,
                     double *max_coa, double *max_norm_coa, int64_t *max_coa_idx,
                     int out_on_device) {
    if (!e || !log_onsets || !max_coa || !max_norm_coa || !max_coa_idx)
        return fail("qm_engine_detect: NULL argument");
    DeviceGuard guard(e->device);
    int ns = 0;
    if (check_step(e, T, fsmp, lsmp, available, &ns)) return 1;
    const double *d_on = nullptr;
    if (stage_onsets(e, log_onsets, onsets_on_device, T, &d_on)) return 1;
    OutStage st;
    if (stage_out(e, ns, out_on_device, max_coa, max_norm_coa, max_coa_idx, &st)) return 1;
    if (detect_core(e, d_on, T, fsmp, ns, available, 1, n_nodes_total, st.a, st.b, st.i)) return 1;
    return fetch_out(e, ns, out_on_device, st, max_coa, max_norm_coa, max_coa_idx);
}

int qm_engine_detect_batch(qm_engine *e, const double *log_onsets, int onsets_on_device,
                           int32_t n_steps, int32_t T, int32_t fsmp, int32_t lsmp, int32_t available,
                           int64_t n_nodes_total, double *max_coa, double *max_norm_coa,
                           int64_t *max_coa_idx, int out_on_device) {
    if (!e || !log_onsets || !max_coa || !max_norm_coa || !max_coa_idx)
        return fail("qm_engine_detect_batch: NULL argument");
    if (n_steps < 1) return fail("qm_engine_detect_batch: n_steps must be >= 1 (got %d)", n_steps);
    DeviceGuard guard(e->device);
    int ns = 0;
    if (check_step(e, T, fsmp, lsmp, available, &ns)) return 1;
    if ((int64_t)n_steps * ns >= INT32_MAX) return fail("qm_engine_detect_batch: too many samples");
    const size_t per_step = (size_t)e->g.n_rows * T;
    const double *d_on = log_onsets;
    if (!onsets_on_device) {
        if (e->d_onsets.ensure(per_step * n_steps)) return 1;
        QM_HIP(copy_in(e->d_onsets.p, log_onsets, per_step * n_steps * sizeof(double), e->stream));
        d_on = e->d_onsets.p;
    }
    const int n_all = n_steps * ns;
    OutStage st;
    if (stage_out(e, n_all, out_on_device, max_coa, max_norm_coa, max_coa_idx, &st)) return 1;
    bool batched = false;
    int sets = 0;
    if (n_steps > 1 && !e->cfg_screen) {
        if (run_stack(e, d_on, T, fsmp, ns, available, 0, ns, nullptr, 0, 0, true, &sets, false, 0, 0,
                      nullptr, n_steps, (int64_t)per_step, &batched))
            return 1;
        // partial sets [sets][n_steps * ns] -> the steps' series back to back
        if (batched && combine(e, e->d_pmax.p, e->d_pidx.p, e->d_psum.p, sets, n_all, 1, e->node_offset,
                               n_nodes_total, st.a, st.b, st.i))
            return 1;
        // (tie_rule = 1 keeps the step axis: the refinement reads the launch's sets with sample t = step * ns + ...)
        if (batched && e->cfg_tie_rule &&
            refine_ties(e, d_on, T, fsmp, available, 0, ns, sets, st.i, n_steps, (int64_t)per_step))
            return 1;
    }
    if (!batched)                                       // step by step (a single step, the screened
        for (int k = 0; k < n_steps; ++k)               // detect, kernels without the step axis)
            if (detect_core(e, d_on + (size_t)k * per_step, T, fsmp, ns, available, 1, n_nodes_total,
                            st.a + (size_t)k * ns, st.b + (size_t)k * ns, st.i + (size_t)k * ns))
                return 1;
    e->last_batched = batched ? n_steps : 1;
    return fetch_out(e, n_all, out_on_device, st, max_coa, max_norm_coa, max_coa_idx);
}

// tie_rule = 1 on a sharded detect (qm_ties.hpp, tie_export_kernel / tie_fold_kernel): after the exchange of
// the ranks' partials, every engine examines the sets its last qm_engine_detect_partial left behind against the
// GRID's maxima ...
int qm_engine_tie_partial(qm_engine *e, const double *log_onsets, int onsets_on_device, int32_t T,
                          int32_t fsmp, int32_t lsmp, int32_t available, const double *d_packed,
                          int32_t n_sets, double *d_tie_packed) {
    if (!e || !log_onsets || !d_packed || !d_tie_packed) return fail("qm_engine_tie_partial: NULL argument");
    if (n_sets < 1) return fail("qm_engine_tie_partial: empty input");
    DeviceGuard guard(e->device);
    int ns = 0;
    if (check_step(e, T, fsmp, lsmp, available, &ns)) return 1;
    if (!e->cfg_tie_rule) return fail("qm_engine_tie_partial: the engine was not configured with tie_rule = 1");
    if (e->last_scan_n != ns || e->last_sets < 1 || !e->last_sets_own)
        return fail("qm_engine_tie_partial: no partial sets of a float64 detect of %d samples on this engine "
                    "(call qm_engine_detect_partial for the step first; the screened sweep leaves none)", ns);
    const double *d_on = nullptr;
    if (stage_onsets(e, log_onsets, onsets_on_device, T, &d_on)) return 1;
    // the grid's largest z per sample: the fold of the gathered maxima (rows [s][0] of [n_sets][3][ns])
    if (e->d_tie_zgrid.ensure(ns)) return 1;
    hipLaunchKernelGGL(qm::tie_zmax_kernel, dim3((ns + 255) / 256), dim3(256), 0, e->stream, d_packed, (int)n_sets,
                       ns, 3 * (int64_t)ns, e->d_tie_zgrid.p);
    QM_HIP(hipGetLastError());
    return refine_ties(e, d_on, T, fsmp, available, 0, ns, e->last_sets,
                       reinterpret_cast<int64_t *>(d_tie_packed + ns), 1, 0, e->d_tie_zgrid.p,
                       reinterpret_cast<unsigned long long *>(d_tie_packed));
}

// ... and folds the gathered outcomes [n_sets][2][n_samples] into the index series (device, in place).
int qm_engine_tie_fold(qm_engine *e, const double *d_tie_gathered, int32_t n_sets, int32_t n_samples,
                       int64_t *d_max_coa_idx) {
    if (!e || !d_tie_gathered || !d_max_coa_idx) return fail("qm_engine_tie_fold: NULL argument");
    if (n_sets < 1 || n_samples < 1) return fail("qm_engine_tie_fold: empty input");
    DeviceGuard guard(e->device);
    hipLaunchKernelGGL(qm::tie_fold_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, e->stream,
                       reinterpret_cast<const unsigned long long *>(d_tie_gathered), (int)n_sets, (int)n_samples,
                       d_max_coa_idx);
    QM_HIP(hipGetLastError());
    return 0;
}

int qm_engine_migrate(qm_engine *e, const double *log_onsets, int onsets_on_device, int32_t T,
                      int32_t fsmp, int32_t lsmp, int32_t available, int64_t n_nodes_total,
                      double *map4d, int map_on_device, int accumulate, double *max_coa,
                      double *max_norm_coa, int64_t *max_coa_idx, int out_on_device) {
    if (!e || !log_onsets || !map4d) return fail("qm_engine_migrate: NULL argument");
    const bool want_scan = max_coa != nullptr;
    if (want_scan && (!max_norm_coa || !max_coa_idx))
        return fail("qm_engine_migrate: all three scan outputs or none");
    DeviceGuard guard(e->device);
    int ns = 0, sets = 0;
    if (check_step(e, T, fsmp, lsmp, available, &ns)) return 1;
    const double *d_on = nullptr;
    if (stage_onsets(e, log_onsets, onsets_on_device, T, &d_on)) return 1;
    OutStage st{nullptr, nullptr, nullptr};
    if (want_scan &&
        stage_out(e, ns, out_on_device, max_coa, max_norm_coa, max_coa_idx, &st))
        return 1;

    if (map_on_device) {
        if (run_stack(e, d_on, T, fsmp, ns, available, 0, ns, map4d, ns, accumulate, want_scan,
                      &sets))
            return 1;
        if (want_scan && combine(e, e->d_pmax.p, e->d_pidx.p, e->d_psum.p, sets, ns, 1,
                                 e->node_offset, n_nodes_total, st.a, st.b, st.i))
            return 1;
        if (want_scan && e->cfg_tie_rule && refine_ties(e, d_on, T, fsmp, available, 0, ns, sets, st.i))
            return 1;
    } else {
        // host volume: stream it through a device chunk buffer, time-chunk by time-chunk
        const int KT = qm::kWave * eff_j(e);
        int64_t chunk = e->cfg_chunk_bytes / (8 * e->n_nodes);
        chunk = std::max<int64_t>(KT, chunk / KT * KT);
        chunk = std::min<int64_t>(chunk, ns);
        if (e->d_chunk.ensure((size_t)e->n_nodes * chunk)) return 1;
        for (int k0 = 0; k0 < ns; k0 += (int)chunk) {
            const int nk = (int)std::min<int64_t>(chunk, ns - k0);
            if (accumulate)
                QM_HIP(copy_in_2d(e->d_chunk.p, nk * sizeof(double), map4d + k0,
                                        (size_t)ns * sizeof(double), nk * sizeof(double),
                                        e->n_nodes, e->stream));
            if (run_stack(e, d_on, T, fsmp, ns, available, k0, nk, e->d_chunk.p, nk, accumulate,
                          want_scan, &sets))
                return 1;
            if (want_scan && combine(e, e->d_pmax.p, e->d_pidx.p, e->d_psum.p, sets, nk, 1,
                                     e->node_offset, n_nodes_total, st.a + k0, st.b + k0,
                                     st.i + k0))
                return 1;
            if (want_scan && e->cfg_tie_rule && refine_ties(e, d_on, T, fsmp, available, k0, nk, sets, st.i))
                return 1;
            QM_HIP(copy_back_2d(map4d + k0, (size_t)ns * sizeof(double), e->d_chunk.p,
                                    nk * sizeof(double), nk * sizeof(double), e->n_nodes, e->stream));
            QM_HIP(hipStreamSynchronize(e->stream));
        }
    }
    if (want_scan) return fetch_out(e, ns, out_on_device, st, max_coa, max_norm_coa, max_coa_idx);
    if (!map_on_device) QM_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int qm_engine_marginal(qm_engine *e, const double *log_onsets, int onsets_on_device, int32_t T,
                       int32_t fsmp, int32_t lsmp, int32_t available, int64_t n_nodes_total,
                       int32_t first_sample, int32_t end_sample, double *coa_map,
                       int map_on_device, double *max_coa, double *max_norm_coa,
                       int64_t *max_coa_idx, int out_on_device) {
    if (!e || !log_onsets || !coa_map) return fail("qm_engine_marginal: NULL argument");
    const bool want_scan = max_coa != nullptr;
    if (want_scan && (!max_norm_coa || !max_coa_idx))
        return fail("qm_engine_marginal: all three scan outputs or none");
    DeviceGuard guard(e->device);
    int ns = 0, sets = 0;
    if (check_step(e, T, fsmp, lsmp, available, &ns)) return 1;
    if (first_sample < 0 || end_sample > ns || first_sample >= end_sample)
        return fail("marginal window [%d, %d) outside the %d scanned samples", first_sample,
                    end_sample, ns);
    const double *d_on = nullptr;
    if (stage_onsets(e, log_onsets, onsets_on_device, T, &d_on)) return 1;
    OutStage st{nullptr, nullptr, nullptr};
    if (want_scan && stage_out(e, ns, out_on_device, max_coa, max_norm_coa, max_coa_idx, &st))
        return 1;
    double *d_map = coa_map;
    if (!map_on_device) {
        if (e->d_marg_out.ensure((size_t)e->n_nodes)) return 1;
        d_map = e->d_marg_out.p;
    }
    if (run_stack(e, d_on, T, fsmp, ns, available, 0, ns, nullptr, 0, 0, want_scan, &sets, true,
                  first_sample, end_sample))
        return 1;
    hipLaunchKernelGGL(qm::marginal_reduce_kernel, dim3((unsigned)((e->n_nodes + 255) / 256)),
                       dim3(256), 0, e->stream, e->d_marg.p, e->marg_tiles, e->n_nodes, d_map);
    QM_HIP(hipGetLastError());
    if (want_scan && combine(e, e->d_pmax.p, e->d_pidx.p, e->d_psum.p, sets, ns, 1,
                             e->node_offset, n_nodes_total, st.a, st.b, st.i))
        return 1;
    if (want_scan && e->cfg_tie_rule && refine_ties(e, d_on, T, fsmp, available, 0, ns, sets, st.i))
        return 1;
    if (!map_on_device) {
        QM_HIP(copy_back(coa_map, d_map, (size_t)e->n_nodes * sizeof(double), e->stream));
        QM_HIP(hipStreamSynchronize(e->stream));
    }
    if (want_scan) return fetch_out(e, ns, out_on_device, st, max_coa, max_norm_coa, max_coa_idx);
    return 0;
}

int qm_engine_find_max_coa(qm_engine *e, const double *map4d, int map_on_device,
                           int32_t n_samples, int64_t n_nodes, double *max_coa,
                           double *max_norm_coa, int64_t *max_coa_idx, int out_on_device) {
    if (!e || !map4d || !max_coa || !max_norm_coa || !max_coa_idx)
        return fail("qm_engine_find_max_coa: NULL argument");
    if (n_samples < 1 || n_nodes < 1) return fail("qm_engine_find_max_coa: empty volume");
    DeviceGuard guard(e->device);
    OutStage st;
    if (stage_out(e, n_samples, out_on_device, max_coa, max_norm_coa, max_coa_idx, &st)) return 1;
    auto scan = [&](const double *vol, int64_t stride, int nk, int k0) -> int {
        // a workgroup = up to 16 adjacent tiles (one wavefront each); ~cfg_scan_waves wavefronts
        // per CU in total; at least 256 nodes per chunk
        const int tiles = (nk + qm::kWave - 1) / qm::kWave;
        const int groups = (tiles + qm::kScanWaves - 1) / qm::kScanWaves;
        const int waves = (tiles + groups - 1) / groups;          // per workgroup, balanced
        const int xgroups = (tiles + waves - 1) / waves;
        int64_t sets = std::max<int64_t>(1, ((int64_t)e->cfg_scan_waves * e->n_cu + tiles - 1) / tiles);
        sets = std::min<int64_t>(sets, std::max<int64_t>(1, n_nodes / 256));
        sets = std::min<int64_t>(sets, 65535);
        const int64_t per = (n_nodes + sets - 1) / sets;
        sets = (n_nodes + per - 1) / per;
        const size_t need = (size_t)sets * nk;
        if (e->d_pmax.ensure(need) || e->d_psum.ensure(need) || e->d_pidx.ensure(need)) return 1;
        hipLaunchKernelGGL(qm::scan_volume_kernel, dim3(xgroups, (unsigned)sets),
                           dim3(waves * qm::kWave), 0, e->stream, vol, stride, nk, n_nodes, per,
                           e->d_pmax.p, e->d_pidx.p, e->d_psum.p);
        QM_HIP(hipGetLastError());
        return combine(e, e->d_pmax.p, e->d_pidx.p, e->d_psum.p, (int)sets, nk, 2, 0, n_nodes,
                       st.a + k0, st.b + k0, st.i + k0);
    };
    if (map_on_device) {
        if (scan(map4d, n_samples, n_samples, 0)) return 1;
    } else {
        int64_t chunk = std::max<int64_t>(1, e->cfg_chunk_bytes / (8 * n_nodes));
        chunk = std::min<int64_t>(chunk, n_samples);
        if (e->d_chunk.ensure((size_t)n_nodes * chunk)) return 1;
        for (int k0 = 0; k0 < n_samples; k0 += (int)chunk) {
            const int nk = (int)std::min<int64_t>(chunk, n_samples - k0);
            QM_HIP(copy_in_2d(e->d_chunk.p, nk * sizeof(double), map4d + k0,
                                    (size_t)n_samples * sizeof(double), nk * sizeof(double),
                                    n_nodes, e->stream));
            if (scan(e->d_chunk.p, nk, nk, k0)) return 1;
            QM_HIP(hipStreamSynchronize(e->stream));
        }
    }
    return fetch_out(e, n_samples, out_on_device, st, max_coa, max_norm_coa, max_coa_idx);
}

double qm_exp_correctly_rounded(double x) { return qm::exp_correctly_rounded(x); }

int qm_engine_exp_correctly_rounded(qm_engine *e, const double *x, int64_t n, double *out) {
    if (!e || !x || !out) return fail("qm_engine_exp_correctly_rounded: NULL argument");
    if (n < 1) return 0;
    DeviceGuard guard(e->device);
    if (e->d_fit_a.ensure((size_t)n) || e->d_fit_b.ensure((size_t)n)) return 1;
    QM_HIP(copy_in(e->d_fit_a.p, x, (size_t)n * sizeof(double), e->stream));
    hipLaunchKernelGGL(qm::exp_cr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream,
                       (const double *)e->d_fit_a.p, n, e->d_fit_b.p);
    QM_HIP(hipGetLastError());
    QM_HIP(copy_back(out, e->d_fit_b.p, (size_t)n * sizeof(double), e->stream));
    return 0;
}

int qm_engine_kernel_log(qm_engine *e, double *total_ms, int32_t *n_calls) {
    if (!e || !total_ms || !n_calls) return fail("NULL argument");
    DeviceGuard guard(e->device);
    double sum = 0.0;
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        QM_HIP(hipEventSynchronize(e->ev_log[i + 1]));
        float f = 0.f;
        QM_HIP(hipEventElapsedTime(&f, e->ev_log[i], e->ev_log[i + 1]));
        sum += f;
    }
    *total_ms = sum;
    *n_calls = (int32_t)(e->ev_used / 2);
    e->ev_used = 0;
    return 0;
}

int qm_engine_last_kernel_ms(qm_engine *e, double *ms) {
    if (!e || !ms) return fail("NULL argument");
    *ms = -1.0;
    if (!e->timed) return 0;
    DeviceGuard guard(e->device);
    QM_HIP(hipEventSynchronize(e->ev1));
    float f = 0.f;
    QM_HIP(hipEventElapsedTime(&f, e->ev0, e->ev1));
    *ms = f;
    return 0;
}

}  // extern "C"
