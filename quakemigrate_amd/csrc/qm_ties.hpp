// qm_ties.hpp -- the reference's arg-max rule on near-ties, opt-in ("tie_rule" = 1).
//
// The reference exponentiates every stack and THEN looks for the maximum with a strict '>'
// (migratelib.c:60-62 exp(stack * (1 / available)), :98-105 the scan): two nodes whose float64 stacks
// differ by an ulp or two can round to the same exp() value -- the lower flat index then wins -- so
// on a near-tie the index it returns is a property of exp()'s rounding.  The engine's default keeps
// the largest float64 z = stack * log2(e) / available, lowest index among equal ones: identical on
// equal sums and wherever the maximum stands alone, different on 11 % of the samples of the
// adversarial twin fixtures (profiles/r04_near_tie_study.txt).
//
// tie_rule = 1 adds a refinement AFTER the stacking launch and its combine, from what they leave
// behind (the partial sets: per (set of bricks, sample) the largest z):
//   1. tie_pairs_kernel: per sample the largest z over the sets, and the sets whose maximum lies
//      within `slack` of it -- only they can hold a node whose exp() ties with the maximum's.  Generic
//      data: one (set, sample) pair per sample.  What a pair costs is its set's nodes: round 6, the
//      shift-reuse fused detect leaves a ROW OF MAXIMA PER BRICK beside its partial sets
//      (StackArgs::brick_max, stack_shift_bricks_kernel) and a "set" is one brick; the other kernels
//      publish eight times as many, smaller sets (four bricks) with the rule on -- run_stack.
//   2. tie_eval_kernel: every node of such a set is stacked again for that ONE sample, rows in
//      ascending order (the reference's sum, bit for bit); nodes within the slack form x = stack *
//      (1 / available) as the reference does and a CORRECTLY ROUNDED exp(x) (double-double
//      arithmetic below: what glibc's scalar exp returns in all but ~1e-3 of its arguments), take
//      part in the sample's largest exp (atomic max on the bit pattern: positive doubles order as
//      integers) and go on a candidate list; tie_pick_kernel then takes, per sample, the lowest flat
//      index among the listed nodes that reach the largest exp.  (A list that overflows -- several
//      whole sets of equal nodes -- is replaced by a second stacking pass, tie_eval_kernel<1>.)
//   3. tie_apply_kernel: the index series takes the refined index.  Values are left as they are
//      (2^z of the largest z: the same maximum within 1e-15).
// A sample with more than kTieMaxSets candidate sets (flat data: everything ties) keeps the default
// rule's index, which for EQUAL sums is the reference's; the count is reported
// (qm_engine_get "tie_overflow_samples").
// Sharded detects (round 6): tie_zmax_kernel takes the GRID's largest z per sample from the gathered partials,
// steps 1-2 run against it on every rank's own sets, tie_export_kernel leaves (largest exp, lowest global
// index reaching it) per sample for one more all-gather, tie_fold_kernel folds the ranks' pairs.
// slack: exp(x1) and exp(x2) can round to one double only if |x1 - x2| <= 2^-52 (one ulp of the result,
// relative), i.e. |z1 - z2| <= 1.4427 * 2^-52; z and x are both products of the same stack with a
// rounded constant (half an ulp each, relative).  slack = 4e-16 + 2^-50 |z| covers both with room; a
// wider net only costs evaluations, never correctness.
#pragma once

#include "qm_kernels.hpp"

namespace qm {

// ---- correctly rounded exp in double-double arithmetic ------------------------------------------
// (every function below switches floating-point contraction OFF: device code is compiled with
// -ffp-contract=fast, which fuses a product into an addition ACROSS statements -- s = p.h + p.l with
// p.h = a * b becomes fma(a, b, p.l) -- and silently breaks the error-free transformations: the first
// device build was off by 3 ulps on 97 % of its arguments while the host build of the same source
// was exact)
struct DD {
    double h, l;
};
__host__ __device__ __forceinline__ DD dd_two_sum(double a, double b) {
#pragma clang fp contract(off)
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__host__ __device__ __forceinline__ DD dd_quick(double a, double b) {     // |a| >= |b|
#pragma clang fp contract(off)
    const double s = a + b;
    return {s, b - (s - a)};
}
__host__ __device__ __forceinline__ DD dd_two_prod(double a, double b) {
#pragma clang fp contract(off)
    const double p = a * b;
    return {p, __builtin_fma(a, b, -p)};
}
__host__ __device__ __forceinline__ DD dd_add(DD a, DD b) {
#pragma clang fp contract(off)
    DD s = dd_two_sum(a.h, b.h);
    const DD t = dd_two_sum(a.l, b.l);
    s.l += t.h;
    s = dd_quick(s.h, s.l);
    s.l += t.l;
    return dd_quick(s.h, s.l);
}
__host__ __device__ __forceinline__ DD dd_mul(DD a, DD b) {
#pragma clang fp contract(off)
    DD p = dd_two_prod(a.h, b.h);
    p.l += a.h * b.l + a.l * b.h;
    return dd_quick(p.h, p.l);
}

// exp(x) rounded to nearest from a double-double evaluation (relative error ~2^-95 before the final
// rounding: the result is the correctly rounded one unless exp(x) lies within ~2^-95 of the midpoint
// of two doubles).  x = k ln2 + r, |r| <= ln2 / 2 (ln2 as a triple double, k ln2_hi exact);
// exp(r) = exp(r / 64)^64 with a degree-12 Taylor polynomial of r / 64 (|r / 64| < 0.0055:
// truncation 2^-119) and six squarings.
__host__ __device__ inline double exp_correctly_rounded(double x) {
#pragma clang fp contract(off)
    if (!(x == x)) return x;                                        // NaN
    if (x > 709.782712893384) return __builtin_inf();
    if (x < -745.2) return 0.0;
    constexpr double kInvLn2 = 0x1.71547652b82fep+0;
    constexpr double kLn2Hi = 0x1.62e42fee00000p-1;                  // 32 significant bits: k * hi is exact
    constexpr double kLn2Lo = 0x1.a39ef35793c76p-33, kLn2LoLo = 0x1.cc01f97b57a08p-87;
    const double k = __builtin_rint(x * kInvLn2);
    DD r = dd_two_sum(x, -k * kLn2Hi);
    DD t = dd_two_prod(k, kLn2Lo);
    r = dd_add(r, DD{-t.h, -t.l});
    r = dd_add(r, DD{-k * kLn2LoLo, 0.0});
    r.h *= 0.015625;                                                // / 64: exact
    r.l *= 0.015625;
    // 1 / n!, n = 12 .. 2, as double-doubles
    constexpr double ch[11] = {0x1.1eed8eff8d898p-29, 0x1.ae64567f544e4p-26, 0x1.27e4fb7789f5cp-22,
                               0x1.71de3a556c734p-19, 0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-13,
                               0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7,  0x1.5555555555555p-5,
                               0x1.5555555555555p-3,  0x1.0000000000000p-1};
    constexpr double cl[11] = {-0x1.2aec959e14c06p-83, -0x1.c062e06d1f209p-80, 0x1.cbbc05b4fa99ap-76,
                               -0x1.c154f8ddc6c00p-73, 0x1.a01a01a01a01ap-76,  0x1.a01a01a01a01ap-73,
                               -0x1.f49f49f49f49fp-65, 0x1.1111111111111p-63,  0x1.5555555555555p-59,
                               0x1.5555555555555p-57,  0.0};
    DD p{ch[0], cl[0]};
    for (int i = 1; i < 11; ++i) p = dd_add(dd_mul(p, r), DD{ch[i], cl[i]});
    p = dd_add(dd_mul(p, r), DD{1.0, 0.0});                         // ... + r
    p = dd_add(dd_mul(p, r), DD{1.0, 0.0});                         // ... + 1
    for (int i = 0; i < 6; ++i) p = dd_mul(p, p);
    // (h + l rounded to nearest is h: the pair is normalised; 2^k scales exactly above the subnormals)
    return __builtin_ldexp(p.h, (int)k);
}

// ---- the refinement ---------------------------------------------------------------------------
constexpr int kTieMaxSets = 8;          // candidate sets per sample beyond which the default index stays
constexpr unsigned long long kTieOverflowKey = ~0ull;   // ... marked so in the sample's exp slot (no double's bits
                                                        // that a finite stack produces)

__host__ __device__ __forceinline__ double tie_slack(double zb) {
    return 4.0e-16 + 0x1p-50 * (zb < 0 ? -zb : zb);
}

struct TieArgs {
    GridDesc g;                    // the brick grid of the launch whose sets are refined
    const double *onsets;          // [S][T] log-onsets of this step
    const int32_t *lut;            // [N][S]
    int T, fsmp, sample0, n_chunk;
    double z_scale, recip;         // log2(e) / available; 1 / available (the reference's factor)
    int groups_lds, groups_direct; // set s < groups_lds: bricks s, s + groups_lds, ...; else the direct launch's
    const int32_t *brick_list;     // ... bricks list[i], i = s - groups_lds, + groups_direct, ... (nullptr: all)
    int n_list;
    int brick_rows;                // > 0 (StackArgs::brick_max, round 6): the LDS launch left a row of maxima per BRICK
                                   // -- set s < brick_rows is brick s alone --, the direct launch's sets follow
    int ns_step;                   // several timesteps in one launch (> 0): sample t of the series belongs to step
    int64_t step_stride;           // t / ns_step, whose onsets start step_stride doubles further on
    int chunks;                    // workgroups per (set, sample) pair
    const double *zbest;           // [n_chunk] largest z over the sets
    const int2 *pairs;             // work list: (set, sample)
    const int32_t *n_pairs;        // its length (device)
    unsigned long long *emax;      // [n_chunk] bit pattern of the largest correctly rounded exp
    int32_t *imin;                 // [n_chunk] lowest local node index reaching it
    int2 *cands;                   // candidate list: (local node, sample) ...
    unsigned long long *cand_keys; // ... and the bit pattern of its exp
    int32_t *n_cands;              // candidates seen (may exceed max_cands: the list then does not count)
    int max_cands;
};

#ifdef QM_TU_STEPS
// Workgroup = 64 samples x kTieSetLanes set-lanes (thread (x, y) scans sets y, y + kTieSetLanes, ...: a
// single thread per sample would walk thousands of sets one load at a time): the largest z over the
// sets, the sets within the slack -> work list.
constexpr int kTieSetLanes = 16;
// (two sources of maxima: sets [0, sets) of pmax, then sets_b more of pmax_b -- the per-brick rows of the
// shift-reuse detect and the direct launch's partial sets behind them)
__global__ __launch_bounds__(64 * kTieSetLanes) void tie_pairs_kernel(
    const double *__restrict__ pmax, int sets, const double *__restrict__ pmax_b, int sets_b, int n,
    int64_t set_stride, double *__restrict__ zbest,
    int2 *__restrict__ pairs, int32_t *__restrict__ n_pairs, int max_pairs, unsigned long long *__restrict__ emax,
    int32_t *__restrict__ imin, int32_t *__restrict__ overflow, const double *__restrict__ zext) {
    __shared__ double smax[kTieSetLanes][64];
    __shared__ int scount[kTieSetLanes][64];
    __shared__ int sbase[64];
    const int x = threadIdx.x, y = threadIdx.y;
    const int t = blockIdx.x * 64 + x;
    const bool live = t < n;
    // (zext: the largest z over EVERY rank's sets of a sharded detect -- this engine's sets are examined against
    // the grid's maximum, not their own)
    auto at = [&](int s) { return s < sets ? pmax[(int64_t)s * set_stride + t]
                                           : pmax_b[(int64_t)(s - sets) * set_stride + t]; };
    const int all = sets + sets_b;
    double zb = -__builtin_inf();
    if (live && zext) zb = zext[t];
    if (live && !zext)
        for (int s = y; s < all; s += kTieSetLanes) {
            const double v = at(s);
            zb = v > zb ? v : zb;                               // (a NaN never wins)
        }
    smax[y][x] = zb;
    __syncthreads();
    for (int k = 0; k < kTieSetLanes; ++k) zb = smax[k][x] > zb ? smax[k][x] : zb;
    const bool finite = zb > -__builtin_inf();                  // nothing finite: the default's (0, index 0) stays
    const double lo = zb - tie_slack(zb);
    int count = 0;
    if (live && finite)
        for (int s = y; s < all; s += kTieSetLanes) count += at(s) >= lo ? 1 : 0;
    scount[y][x] = count;
    __syncthreads();
    if (y == 0 && live) {
        int total = 0;
        for (int k = 0; k < kTieSetLanes; ++k) total += scount[k][x];
        zbest[t] = zb;
        emax[t] = 0ull;
        imin[t] = INT32_MAX;
        int base = -1;
        if (total > kTieMaxSets) {
            emax[t] = kTieOverflowKey;                          // (no pair of this sample: nobody touches it again)
            atomicAdd(overflow, 1);
        } else if (total > 0) {
            base = atomicAdd(n_pairs, total);
            if (base + total > max_pairs) {                     // (cannot happen: max_pairs = kTieMaxSets * n)
                atomicAdd(overflow, 1);
                base = -1;
            }
        }
        sbase[x] = base;
    }
    __syncthreads();
    if (!live || !finite || sbase[x] < 0) return;
    int slot = sbase[x];
    for (int k = 0; k < y; ++k) slot += scount[k][x];
    for (int s = y; s < all; s += kTieSetLanes)
        if (at(s) >= lo) pairs[slot++] = make_int2(s, t);
}

// Workgroup = (pair, chunk): the nodes of the pair's set, for the pair's one sample.
template <int PASS>
__global__ __launch_bounds__(256) void tie_eval_kernel(TieArgs a) {
    // (the grid is sized for a generic step's pairs; a step with more is covered by the stride)
    const int n_pairs = *a.n_pairs, chunk = blockIdx.x % a.chunks, stride = gridDim.x / a.chunks;
    if (PASS == 1 && *a.n_cands <= a.max_cands) return;     // the candidate list held them all
    const GridDesc &g = a.g;
    const int S = g.n_rows;
    for (int pair = blockIdx.x / a.chunks; pair < n_pairs; pair += stride) {
        const int2 w = a.pairs[pair];
        const int set = w.x, t = w.y;
        const double zb = a.zbest[t], lo = zb - tie_slack(zb);
        const int kstep = a.ns_step > 0 ? t / a.ns_step : 0;
        const double *onsets = a.onsets + (int64_t)kstep * a.step_stride;
        const int64_t col = (int64_t)(t - kstep * a.ns_step) + a.sample0 + a.fsmp;
        const int sets_lds = a.brick_rows > 0 ? a.brick_rows : a.groups_lds;
        const bool direct = set >= sets_lds;
        const int step = direct ? a.groups_direct : a.groups_lds;
        const int first = direct ? set - sets_lds : set;
        int n_list = direct ? (a.brick_list ? a.n_list : g.nbricks) : g.nbricks;
        if (!direct && a.brick_rows > 0) n_list = first + 1;        // (the one brick)
        unsigned long long best = 0ull;
        int best_node = INT32_MAX;
        for (int i = first + chunk * step; i < n_list; i += a.chunks * step) {
            const int b = (direct && a.brick_list) ? a.brick_list[i] : i;
            int x0, y0, z0, vx, vy, vz;
            brick_extents(g, b, x0, y0, z0, vx, vy, vz);
            const int nvalid = vx * vy * vz;
            for (int m = threadIdx.x; m < nvalid; m += blockDim.x) {
                const int node = brick_walk_node(g, x0, y0, z0, vy, vz, m);
                const int32_t *row = a.lut + (int64_t)node * S;
                double stack = 0.0;
                // ascending rows, one add each (migratelib.c:54-59); six rows' loads in flight at a time -- the
                // two dependent loads per row one after the other made this kernel 1.35 ms of a C3 step
                constexpr int U = 6;
                int r = 0;
                for (; r + U <= S; r += U) {
                    int d[U];
                    double v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) d[u] = row[r + u];
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] = onsets[(int64_t)(r + u) * a.T + (d[u] < 0 ? 0 : d[u]) + col];
#pragma unroll
                    for (int u = 0; u < U; ++u) stack += v[u];
                }
                for (; r < S; ++r) {
                    int d = row[r];
                    d = d < 0 ? 0 : d;
                    stack += onsets[(int64_t)r * a.T + d + col];
                }
                if (!(stack * a.z_scale >= lo)) continue;
                const double e = exp_correctly_rounded(stack * a.recip);
                const unsigned long long key = (unsigned long long)__double_as_longlong(e);
                if (PASS == 0) {
                    best = key > best ? key : best;
                    const int at = atomicAdd(a.n_cands, 1);
                    if (at < a.max_cands) {
                        a.cands[at] = make_int2(node, t);
                        a.cand_keys[at] = key;
                    }
                } else if (key == a.emax[t]) {
                    best_node = node < best_node ? node : best_node;
                }
            }
        }
        if (PASS == 0) {
            if (best) atomicMax(&a.emax[t], best);
        } else if (best_node != INT32_MAX) {
            atomicMin(&a.imin[t], best_node);
        }
    }
}

// the lowest node among the listed candidates that reach their sample's largest exp
__global__ __launch_bounds__(256) void tie_pick_kernel(TieArgs a) {
    const int n = *a.n_cands;
    if (n > a.max_cands) return;                            // (tie_eval_kernel<1> does it)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 c = a.cands[i];
        if (a.cand_keys[i] == a.emax[c.y]) atomicMin(&a.imin[c.y], c.x);
    }
}

// the device's evaluation of exp_correctly_rounded (the tests pin it to the host's, bit for bit)
__global__ __launch_bounds__(256) void exp_cr_kernel(const double *__restrict__ x, int64_t n,
                                                     double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = exp_correctly_rounded(x[i]);
}

__global__ __launch_bounds__(256) void tie_apply_kernel(const int32_t *__restrict__ imin, int n,
                                                        int64_t node_offset, int64_t *__restrict__ o_idx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && imin[t] != INT32_MAX) o_idx[t] = node_offset + imin[t];
}

// Sharded detects: an engine's refinement against the grid's maxima leaves, per sample, the largest correctly
// rounded exp among ITS nodes within the slack (0: none; kTieOverflowKey: more candidate sets than it follows) and
// the lowest GLOBAL index reaching it -- packed [2][n] for one all-gather --, and every rank folds the ranks'
// pairs: the largest exp, the lowest index among the ranks that reach it.  A sample that overflowed anywhere
// keeps the default rule's index on every rank.
__global__ __launch_bounds__(256) void fill_kernel(double *__restrict__ p, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void tie_zmax_kernel(const double *__restrict__ pmax, int sets, int n,
                                                       int64_t set_stride, double *__restrict__ zmax) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double zb = -__builtin_inf();
    for (int s = 0; s < sets; ++s) {
        const double v = pmax[(int64_t)s * set_stride + t];
        zb = v > zb ? v : zb;
    }
    zmax[t] = zb;
}
__global__ __launch_bounds__(256) void tie_export_kernel(const unsigned long long *__restrict__ emax,
                                                         const int32_t *__restrict__ imin, int n,
                                                         int64_t node_offset, unsigned long long *__restrict__ o_key,
                                                         int64_t *__restrict__ o_idx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const bool none = imin[t] == INT32_MAX;
    o_key[t] = (none && emax[t] != kTieOverflowKey) ? 0ull : emax[t];
    o_idx[t] = none ? INT64_MAX : node_offset + imin[t];
}
__global__ __launch_bounds__(256) void tie_fold_kernel(const unsigned long long *__restrict__ packed, int sets, int n,
                                                       int64_t *__restrict__ o_idx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    unsigned long long best = 0ull;
    int64_t at = INT64_MAX;
    for (int s = 0; s < sets; ++s) {
        const unsigned long long key = packed[(int64_t)s * 2 * n + t];
        const int64_t i = (int64_t)packed[(int64_t)s * 2 * n + n + t];
        if (key > best) {
            best = key;
            at = i;
        } else if (key == best && i < at) {
            at = i;
        }
    }
    if (best != 0ull && best != kTieOverflowKey && at != INT64_MAX) o_idx[t] = at;
}
#endif  // QM_TU_STEPS

}  // namespace qm
