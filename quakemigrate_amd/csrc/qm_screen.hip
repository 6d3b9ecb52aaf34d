// qm_screen.hip -- launch sequence of the opt-in screened detect (qm_screen.hpp; "screen" = 1): the
// exact-integer sweep over every node-sample, the candidate cells, their float64 refinement, and the
// per-step outcome flags that travel to the host asynchronously.  Never the default.
#define QM_TU_SCREEN 1
#include "qm_engine.hpp"

// fold the per-step outcomes that have reached the host into the counters (synchronises)
int drain_flags(qm_engine *e) {
    if (e->flags_pending == 0) return 0;
    QM_HIP(hipStreamSynchronize(e->stream));
    for (; e->flags_pending > 0; --e->flags_pending) {
        const int32_t *f = e->h_flags + 2 * e->flags_head;
        if (f[0] != 0) ++e->fallback_steps;
        else ++e->screened_steps;
        e->last_candidates = f[1];
        e->flags_head = (e->flags_head + 1) % kFlagRing;
    }
    return 0;
}

namespace {
template <int JP, int NCH>
int launch_screen(qm_engine *e, qm::ScreenArgs &a, size_t lds, int threads) {
    QM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&qm::screen_lds_kernel<JP, NCH>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((qm::screen_lds_kernel<JP, NCH>),
                       dim3((unsigned)(a.ntiles * ((a.ngroups + 7) / 8 * 8))), dim3(threads), lds,
                       e->stream, a);
    QM_HIP(hipGetLastError());
    return 0;
}

template <int JP>
int launch_screen_jp(qm_engine *e, qm::ScreenArgs &a, size_t lds, int threads) {
    switch (e->g.row_pad / 8) {
        case 1: return launch_screen<JP, 1>(e, a, lds, threads);
        case 2: return launch_screen<JP, 2>(e, a, lds, threads);
        case 3: return launch_screen<JP, 3>(e, a, lds, threads);
        case 4: return launch_screen<JP, 4>(e, a, lds, threads);
        case 5: return launch_screen<JP, 5>(e, a, lds, threads);
        case 6: return launch_screen<JP, 6>(e, a, lds, threads);
        case 7: return launch_screen<JP, 7>(e, a, lds, threads);
        case 8: return launch_screen<JP, 8>(e, a, lds, threads);
        default: return fail("screening supports at most 64 table rows");
    }
}
}  // namespace

// Whole-scan detect through the screening path.  On success with *screened = true the partial
// sets [*n_sets][ns] are in e->d_pmax/d_pidx/d_psum exactly as run_stack leaves them.  *screened =
// false (nothing usable was produced) if some sample had more candidate cells than slots or the
// onsets hold a non-finite value: the caller then runs the float64 kernel.
int run_screen(qm_engine *e, const double *d_onsets, int T, int fsmp, int ns, int available,
               int *n_sets, bool *screened) {
    *screened = false;
    const ScreenPlan plan = screen_plan(e, e->g.n_rows, ns);
    const int JP = plan.jp;
    if (JP == 0) return 0;
    if (ensure_screen_tables(e, plan)) return 1;
    e->last_plan_jp = plan.jp;
    e->last_plan_big = plan.big ? 1 : 0;
    const qm::GridDesc &g = e->sg;
    if (!e->h_flags)
        QM_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_flags),
                             2 * kFlagRing * sizeof(int32_t), hipHostMallocDefault));
    const int KT = 128 * JP;
    const int ntiles = (ns + KT - 1) / KT;
    const int64_t ns_pad = (int64_t)ntiles * KT;
    const int S = g.n_rows;
    const int n_fit = g.nbricks - e->n_swide;
    if (n_fit < 2) return 0;                            // a single cell: nothing to screen
    const int groups = n_fit > 0 ? (e->cfg_groups > 0 ? std::min(e->cfg_groups, g.nbricks)
                                                       : auto_groups(e, ntiles, g.nbricks, plan.big ? 1 : 2))
                                 : 0;
    const int groups_direct =
        e->n_swide > 0 ? (e->cfg_groups > 0 ? std::min(e->cfg_groups, e->n_swide)
                                            : auto_groups(e, (ns + 63) / 64, e->n_swide, 4))
                       : 0;
    const int sets = groups_direct + 1;
    constexpr int kGroupsPerBlock = 32;
    if (e->d_onq.ensure((size_t)S * T) || e->d_rowmax.ensure(S) || e->d_sparams.ensure(4) ||
        e->d_cell.ensure((size_t)g.nbricks * ns_pad) ||
        e->d_gmax.ensure((size_t)std::max(1, groups) * ns_pad) || e->d_pm.ensure(ns) || e->d_ssum.ensure((size_t)std::max(1, groups) * ns) ||
        e->d_counts.ensure(ns) || e->d_cells.ensure((size_t)ns * qm::kScreenSlots) ||
        e->d_work.ensure((size_t)ns * qm::kScreenSlots) ||
        e->d_flags.ensure(4) || e->d_cand_z.ensure((size_t)ns * qm::kScreenSlots) ||
        e->d_cand_idx.ensure((size_t)ns * qm::kScreenSlots))
        return 1;
    const size_t need = (size_t)sets * ns;
    if (e->d_pmax.ensure(need) || e->d_psum.ensure(need) || e->d_pidx.ensure(need)) return 1;

    hipEvent_t ev_begin = e->ev0, ev_end = e->ev1;
    if (e->log_timing) {
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
    hipStream_t s = e->stream;
    QM_HIP(hipMemsetAsync(e->d_counts.p, 0, (size_t)ns * sizeof(int32_t), s));
    QM_HIP(hipMemsetAsync(e->d_flags.p, 0, 4 * sizeof(int32_t), s));
    // this step's fixed-point scale (device-side: max |L| -> k) and the quantised log-onsets
    hipLaunchKernelGGL(qm::screen_rowmax_kernel, dim3(S), dim3(256), 0, s, d_onsets, T,
                       e->d_rowmax.p);
    hipLaunchKernelGGL(qm::screen_quantise_kernel, dim3(S), dim3(256), 0, s, d_onsets, T, S,
                       available, (const double *)e->d_rowmax.p, e->d_onq.p,
                       reinterpret_cast<qm::ScreenParams *>(e->d_sparams.p), e->d_flags.p);
    QM_HIP(hipGetLastError());

    qm::ScreenArgs a{};
    a.g = g;
    a.onsets_q = e->d_onq.p;
    a.rel = e->d_srel.p;
    a.brick_meta = e->d_smeta.p;
    a.brick_total = e->d_stotal.p;
    a.T = T;
    a.fsmp = fsmp;
    a.n_samples = ns;
    a.ntiles = ntiles;
    a.ngroups = groups;
    a.window_bytes = plan.window_bytes(e);
    a.params = reinterpret_cast<const qm::ScreenParams *>(e->d_sparams.p);
    a.cell_max = e->d_cell.p;
    a.group_max = e->d_gmax.p;
    a.ns_pad = ns_pad;
    a.part_sum = e->d_ssum.p;
    QM_HIP(hipEventRecord(ev_begin, s));               // the timing log brackets the sweep kernel
    if (groups > 0) {
        const size_t lds = (size_t)plan.lds_bytes(e);
        const int threads = plan.threads();
        if (JP == 4 ? launch_screen_jp<4>(e, a, lds, threads)
                    : JP == 2 ? launch_screen_jp<2>(e, a, lds, threads)
                              : launch_screen_jp<1>(e, a, lds, threads))
            return 1;
    }
    QM_HIP(hipEventRecord(ev_end, s));
    if (groups_direct > 0) {
        // bricks whose windows do not fit: exact float64 partial sets from the direct kernel
        qm::StackArgs d{};
        d.g = g;
        d.onsets = d_onsets;
        d.lut = e->d_lut.p;
        d.T = T;
        d.fsmp = fsmp;
        d.n_samples = ns;
        d.sample0 = 0;
        d.n_chunk = ns;
        d.ntiles = (ns + 63) / 64;
        d.ngroups = groups_direct;
        d.z_scale = 1.4426950408889634074 / (double)available;
        d.want_scan = 1;
        d.set0 = 0;
        d.part_max = e->d_pmax.p;
        d.part_idx = e->d_pidx.p;
        d.part_sum = e->d_psum.p;
        d.brick_list = e->d_swide.p;
        d.n_list = e->n_swide;
        d.n_nodes = e->n_nodes;
        const size_t publish_bytes = (size_t)3 * 8 * 64 * sizeof(double);
        bool built = false;
        QM_TABLE(qm::launch_direct_detect(
            1, d, {(unsigned)(d.ntiles * ((groups_direct + 7) / 8 * 8)), 512, publish_bytes, s},
            &built));
        if (!built) return fail("no direct stacking kernel built");
    }
    const unsigned tcols = (unsigned)((ns + 63) / 64);
    hipLaunchKernelGGL(qm::screen_peak_kernel, dim3(tcols), dim3(256), 0, s,
                       (const int32_t *)e->d_gmax.p, ns_pad, ns, groups, e->d_pm.p);
    hipLaunchKernelGGL(qm::screen_candidates_kernel,
                       dim3(tcols, (unsigned)std::max(1, (groups + kGroupsPerBlock - 1) / kGroupsPerBlock)),
                       dim3(256), 0, s, (const int32_t *)e->d_cell.p, (const int32_t *)e->d_gmax.p,
                       ns_pad, ns, g.nbricks, groups, kGroupsPerBlock, (const int32_t *)e->d_pm.p,
                       reinterpret_cast<const qm::ScreenParams *>(e->d_sparams.p), e->d_counts.p,
                       e->d_cells.p, e->d_work.p, e->d_flags.p);
    QM_HIP(hipGetLastError());
    qm::RefineArgs r{};
    r.g = g;
    r.onsets = d_onsets;
    r.lut = e->d_lut.p;
    r.T = T;
    r.fsmp = fsmp;
    r.n_samples = ns;
    r.z_scale = 1.4426950408889634074 / (double)available;
    r.cells = e->d_cells.p;
    r.work = e->d_work.p;
    r.flags = e->d_flags.p;
    r.cand_z = e->d_cand_z.p;
    r.cand_idx = e->d_cand_idx.p;
    hipLaunchKernelGGL(qm::screen_refine_kernel, dim3((unsigned)(8 * e->n_cu)), dim3(256), 0, s, r);
    hipLaunchKernelGGL(qm::screen_collect_kernel, dim3((ns + 63) / 64), dim3(256), 0, s,
                       (const int32_t *)e->d_counts.p, (const double *)e->d_cand_z.p,
                       (const int64_t *)e->d_cand_idx.p, (const double *)e->d_ssum.p, groups, ns,
                       e->d_pmax.p + (size_t)groups_direct * ns,
                       e->d_pidx.p + (size_t)groups_direct * ns,
                       e->d_psum.p + (size_t)groups_direct * ns);
    QM_HIP(hipGetLastError());
    e->timed = !e->log_timing;
    // the outcome travels to the host asynchronously (statistics only: the decision to redo the
    // step in float64 is taken on the device, see detect_core)
    if (e->flags_pending == kFlagRing && drain_flags(e)) return 1;
    QM_HIP(hipMemcpyAsync(e->h_flags + 2 * ((e->flags_head + e->flags_pending) % kFlagRing),
                          e->d_flags.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ++e->flags_pending;
    *n_sets = sets;
    *screened = true;
    return 0;
}

extern "C" {

int qm_exp2f_max_error(qm_engine *e, float lo, float hi, double *max_rel_error) {
    if (!e || !max_rel_error) return fail("qm_exp2f_max_error: NULL argument");
    if (!(lo <= hi) || (lo < 0.f) != (hi < 0.f))
        return fail("qm_exp2f_max_error: need lo <= hi of one sign");
    DeviceGuard guard(e->device);
    constexpr int kBlocks = 4096;
    if (e->d_fit_part.ensure(kBlocks)) return 1;
    hipLaunchKernelGGL(qm::exp2f_error_kernel, dim3(kBlocks), dim3(256), 0, e->stream, lo, hi,
                       e->d_fit_part.p);
    QM_HIP(hipGetLastError());
    std::vector<double> h(kBlocks);
    QM_HIP(copy_back(h.data(), e->d_fit_part.p, kBlocks * sizeof(double), e->stream));
    QM_HIP(hipStreamSynchronize(e->stream));
    *max_rel_error = *std::max_element(h.begin(), h.end());
    return 0;
}

}  // extern "C"
