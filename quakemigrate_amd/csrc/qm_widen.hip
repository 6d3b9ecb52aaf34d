// qm_widen.hip -- the rows SURVEY.md section 8(f) marks "next", each side of the path: the onset
// stage on the device (STALTAOnset._onset + lib.migrate's clip/log; stalta.py:491-583, onsetlib.c:35-148)
// and locate's post-reductions of the marginalised map (QuakeScan._calculate_location's array work,
// scan.py:696-1077; the cubic RBF's fine-grid maximum, scan.py:777-812).
#define QM_TU_WIDEN 1
#include "qm_engine.hpp"
#include "qm_locate.hpp"

namespace {

// Weights of one axis of the reference's filter as a function of d = i - j.
// util.gaussian_3d (util.py:76-116) samples exp(-x^2 / (2 sgm^2)) at x = k - (n-1)/2,
// k = 0..n-1; fftconvolve(..., mode="same") centres the full convolution at (n-1)//2, so
// out[i] = sum_j in[j] * flt[i - j + (n-1)//2]: symmetric for odd n, shifted by half a node
// for even n (which is why the reference filters twice, mirrored).  `mirror` gives the second
// pass, w(d) -> w(-d).  Weights below 1e-20 of the peak are dropped: the reference's own FFT
// round-off is four orders of magnitude above that.
int axis_taps(int n, double sgm, bool mirror, qm::Taps *t) {
    const int c = (n - 1) / 2;
    const double half = 0.5 * (n - 1);
    int R = (int)std::ceil(sgm * 9.6) + 1;              // exp(-(9.6)^2 / 2) = 1e-20
    int lo = 0, hi = -1;
    bool any = false;
    for (int d = -R; d <= R; ++d) {
        const int k = (mirror ? -d : d) + c;
        if (k < 0 || k > n - 1) continue;
        if (!any) lo = d;
        hi = d;
        any = true;
    }
    if (!any) return fail("gaussian filter: empty support");
    if (hi - lo + 1 > qm::kMaxTaps)
        return fail("gaussian filter: sgm %.3g needs %d taps, more than %d", sgm, hi - lo + 1,
                    qm::kMaxTaps);
    t->lo = lo;
    t->n = hi - lo + 1;
    for (int d = lo; d <= hi; ++d) {
        const double x = (double)((mirror ? -d : d) + c) - half;
        t->w[d - lo] = std::exp(-(x * x) / (2.0 * sgm * sgm));
    }
    return 0;
}

}  // namespace

extern "C" {

int qm_engine_onsets(qm_engine *e, const double *signals, int signals_on_device,
                     int32_t n_traces, int32_t t_samples, const int32_t *trace_row, int32_t n_rows,
                     const int32_t *nsta, const int32_t *nlta, int transform, int position,
                     int32_t taper_pad, double min_onset_value, double *raw_onsets,
                     double *log_onsets, int out_on_device) {
    if (!e || !signals || !trace_row || !nsta || !nlta || !log_onsets)
        return fail("qm_engine_onsets: NULL argument");
    if (n_traces < 1 || n_rows < 1 || t_samples < 1) return fail("qm_engine_onsets: empty input");
    if (transform != 0 && transform != 1) return fail("transform must be 0 (energy) or 1 (abs)");
    if (position < 0 || position > 2)
        return fail("position must be 0 (classic), 1 (centred) or 2 (recursive)");
    std::vector<int> per_row(n_rows, 0);
    for (int i = 0; i < n_traces; ++i) {
        if (trace_row[i] < 0 || trace_row[i] >= n_rows) return fail("trace %d: row out of range", i);
        ++per_row[trace_row[i]];
    }
    for (int r = 0; r < n_rows; ++r)
        if (per_row[r] == 0) return fail("onset row %d has no trace", r);
    DeviceGuard guard(e->device);
    const size_t sig = (size_t)n_traces * t_samples, out = (size_t)n_rows * t_samples;
    const double *d_sig = signals;
    if (!signals_on_device) {
        if (e->d_sig.ensure(sig)) return 1;
        QM_HIP(copy_in(e->d_sig.p, signals, sig * sizeof(double), e->stream));
        d_sig = e->d_sig.p;
    }
    if (e->d_sta.ensure(sig) || e->d_lta.ensure(sig) ||
        e->d_onset_meta.ensure((size_t)n_traces + 2 * n_rows))
        return 1;
    std::vector<int32_t> meta(trace_row, trace_row + n_traces);
    meta.insert(meta.end(), nsta, nsta + n_rows);
    meta.insert(meta.end(), nlta, nlta + n_rows);
    QM_HIP(copy_in(e->d_onset_meta.p, meta.data(), meta.size() * sizeof(int32_t), e->stream));
    QM_HIP(hipStreamSynchronize(e->stream));            // `meta` is a stack-lifetime buffer
    qm::OnsetArgs a{};
    a.signals = d_sig;
    a.trace_row = e->d_onset_meta.p;
    a.nsta = e->d_onset_meta.p + n_traces;
    a.nlta = e->d_onset_meta.p + n_traces + n_rows;
    a.sta = e->d_sta.p;
    a.lta = e->d_lta.p;
    a.n_traces = n_traces; a.n_rows = n_rows; a.T = t_samples;
    a.transform = transform; a.position = position; a.taper_pad = taper_pad;
    a.min_onset_value = min_onset_value;
    double *d_log = log_onsets, *d_raw = raw_onsets;
    if (!out_on_device) {
        if (e->d_onsets.ensure(out)) return 1;
        d_log = e->d_onsets.p;
        if (raw_onsets) {
            if (e->d_raw.ensure(out)) return 1;
            d_raw = e->d_raw.p;
        }
    }
    a.raw = d_raw;
    a.logged = d_log;
    {
        // one workgroup per trace; the transformed trace lives in LDS if it fits (20 480 samples)
        const size_t lds = (size_t)t_samples * sizeof(double);
        const int in_lds = lds <= 160 * 1024 ? 1 : 0;
        if (in_lds)
            QM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&qm::stalta_sums_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(qm::stalta_sums_kernel, dim3(n_traces), dim3(256), in_lds ? lds : 0,
                           e->stream, a, in_lds);
        QM_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(qm::onset_rows_kernel, dim3((unsigned)((out + 255) / 256)), dim3(256), 0,
                       e->stream, a);
    QM_HIP(hipGetLastError());
    if (!out_on_device) {
        QM_HIP(copy_back(log_onsets, d_log, out * sizeof(double), e->stream));
        if (raw_onsets)
            QM_HIP(copy_back(raw_onsets, d_raw, out * sizeof(double), e->stream));
        QM_HIP(hipStreamSynchronize(e->stream));
    }
    return 0;
}

int qm_engine_locate_fits(qm_engine *e, const double *coa_map, int map_on_device, int32_t nx,
                          int32_t ny, int32_t nz, double sgm, double cov_thresh,
                          const double *node_spacing, double *norm_map, double *smoothed_map,
                          int out_on_device, double *summary, double *gau_window,
                          double *spline_window) {
    if (!e || !coa_map || !node_spacing || !summary || !gau_window || !spline_window)
        return fail("qm_engine_locate_fits: NULL argument");
    if (nx < 1 || ny < 1 || nz < 1) return fail("qm_engine_locate_fits: empty grid");
    if (!(sgm > 0.0)) return fail("qm_engine_locate_fits: sgm must be positive");
    DeviceGuard guard(e->device);
    const int64_t n = (int64_t)nx * ny * nz;
    constexpr int NB = qm::kFitBlocks, BS = qm::kFitBlock;
    if (e->d_fit_a.ensure((size_t)n) || e->d_fit_b.ensure((size_t)n) ||
        e->d_fit_c.ensure((size_t)n) || e->d_fit_part.ensure((size_t)NB * 6) ||
        e->d_fit_pidx.ensure(NB) || e->d_fit_val.ensure(32) || e->d_fit_win.ensure(343 + 125))
        return 1;
    hipStream_t s = e->stream;
    const double *d_in = coa_map;
    if (!map_on_device) {
        QM_HIP(copy_in(e->d_fit_c.p, coa_map, (size_t)n * sizeof(double), s));
        d_in = e->d_fit_c.p;
    }
    double *val = e->d_fit_val.p;
    // device scalars: 0 map max, 1 map argmax, 2 pass-1 max, 3 -, 4 pass-2 max, 5 -,
    // 6 smoothed mean, 7 smoothed argmax, 8..11 first moments, 12..17 second moments, 18 -
    auto argmax = [&](const double *m, double *out_v, double *out_i) -> int {
        hipLaunchKernelGGL(qm::argmax_partial_kernel, dim3(NB), dim3(BS), 0, s, m, n,
                           e->d_fit_part.p, e->d_fit_pidx.p);
        hipLaunchKernelGGL(qm::argmax_final_kernel, dim3(1), dim3(BS), 0, s, e->d_fit_part.p,
                           e->d_fit_pidx.p, NB, out_v, out_i);
        QM_HIP(hipGetLastError());
        return 0;
    };
    const unsigned node_blocks = (unsigned)((n + BS - 1) / BS);
    auto smooth = [&](const double *in, double *tmp, double *out, bool mirror,
                      const double *div) -> int {
        qm::Taps tx, ty, tz;
        if (axis_taps(nx, sgm, mirror, &tx) || axis_taps(ny, sgm, mirror, &ty) ||
            axis_taps(nz, sgm, mirror, &tz))
            return 1;
        hipLaunchKernelGGL(qm::smooth_axis_kernel, dim3(node_blocks), dim3(BS), 0, s, in, out,
                           nx, ny, nz, 0, tx, div);
        hipLaunchKernelGGL(qm::smooth_axis_kernel, dim3(node_blocks), dim3(BS), 0, s,
                           (const double *)out, tmp, nx, ny, nz, 1, ty, (const double *)nullptr);
        hipLaunchKernelGGL(qm::smooth_axis_kernel, dim3(node_blocks), dim3(BS), 0, s,
                           (const double *)tmp, out, nx, ny, nz, 2, tz, (const double *)nullptr);
        QM_HIP(hipGetLastError());
        return 0;
    };

    // (1) coa_map / nanmax(coa_map)                                       scan.py:721
    double *d_norm = (norm_map && out_on_device) ? norm_map : e->d_fit_a.p;
    if (argmax(d_in, val + 0, val + 1)) return 1;
    hipLaunchKernelGGL(qm::divide_kernel, dim3(NB), dim3(BS), 0, s, d_in, (const double *)val, n,
                       d_norm);
    if (argmax(d_norm, val + 18, val + 1)) return 1;

    // (2) _gaufilt3d: filter, normalise, filter mirrored, normalise        scan.py:1033-1041
    double *d_smooth = (smoothed_map && out_on_device) ? smoothed_map : e->d_fit_b.p;
    double *d_tmp = e->d_fit_c.p;           // the staged input is dead once d_norm exists
    if (smooth(d_norm, d_tmp, d_smooth, false, nullptr)) return 1;
    if (argmax(d_smooth, val + 2, val + 3)) return 1;
    // second pass: its first axis divides by the pass-1 maximum (the filter is linear)
    {
        // no axis may filter in place: x -> d_tmp, y -> d_smooth, z -> d_tmp
        qm::Taps tx, ty, tz;
        if (axis_taps(nx, sgm, true, &tx) || axis_taps(ny, sgm, true, &ty) ||
            axis_taps(nz, sgm, true, &tz))
            return 1;
        hipLaunchKernelGGL(qm::smooth_axis_kernel, dim3(node_blocks), dim3(BS), 0, s,
                           (const double *)d_smooth, d_tmp, nx, ny, nz, 0, tx,
                           (const double *)(val + 2));
        hipLaunchKernelGGL(qm::smooth_axis_kernel, dim3(node_blocks), dim3(BS), 0, s,
                           (const double *)d_tmp, d_smooth, nx, ny, nz, 1, ty,
                           (const double *)nullptr);
        hipLaunchKernelGGL(qm::smooth_axis_kernel, dim3(node_blocks), dim3(BS), 0, s,
                           (const double *)d_smooth, d_tmp, nx, ny, nz, 2, tz,
                           (const double *)nullptr);
        QM_HIP(hipGetLastError());
    }
    if (argmax(d_tmp, val + 4, val + 5)) return 1;
    hipLaunchKernelGGL(qm::divide_kernel, dim3(NB), dim3(BS), 0, s, (const double *)d_tmp,
                       (const double *)(val + 4), n, d_smooth);
    if (argmax(d_smooth, val + 18, val + 7)) return 1;
    hipLaunchKernelGGL(qm::sum_partial_kernel, dim3(NB), dim3(BS), 0, s,
                       (const double *)d_smooth, n, e->d_fit_part.p);
    hipLaunchKernelGGL(qm::sums_final_kernel, dim3(1), dim3(BS), 0, s,
                       (const double *)e->d_fit_part.p, NB, 1, 1.0 / (double)n, val + 6);

    // (3) _covfit3d on the normalised (unsmoothed) map                     scan.py:973-999
    hipLaunchKernelGGL(qm::moments_partial_kernel<0>, dim3(NB), dim3(BS), 0, s,
                       (const double *)d_norm, nx, ny, nz, cov_thresh, node_spacing[0],
                       node_spacing[1], node_spacing[2], (const double *)nullptr,
                       e->d_fit_part.p);
    hipLaunchKernelGGL(qm::sums_final_kernel, dim3(4), dim3(BS), 0, s,
                       (const double *)e->d_fit_part.p, NB, 4, 1.0, val + 8);
    hipLaunchKernelGGL(qm::moments_partial_kernel<1>, dim3(NB), dim3(BS), 0, s,
                       (const double *)d_norm, nx, ny, nz, cov_thresh, node_spacing[0],
                       node_spacing[1], node_spacing[2], (const double *)(val + 8),
                       e->d_fit_part.p);
    hipLaunchKernelGGL(qm::sums_final_kernel, dim3(6), dim3(BS), 0, s,
                       (const double *)e->d_fit_part.p, NB, 6, 1.0, val + 12);
    hipLaunchKernelGGL(qm::moments_scale_kernel, dim3(1), dim3(64), 0, s, val + 12,
                       (const double *)(val + 8));

    // (4) the windows the Gaussian (7^3, smoothed map) and spline (5^3, normalised map) fits use
    hipLaunchKernelGGL(qm::window_kernel, dim3(2), dim3(256), 0, s, (const double *)d_smooth, nx,
                       ny, nz, 7, (const double *)(val + 7), e->d_fit_win.p);
    hipLaunchKernelGGL(qm::window_kernel, dim3(1), dim3(128), 0, s, (const double *)d_norm, nx,
                       ny, nz, 5, (const double *)(val + 1), e->d_fit_win.p + 343);
    QM_HIP(hipGetLastError());

    double h[32], w[343 + 125];
    QM_HIP(copy_back(h, val, sizeof(h), s));
    QM_HIP(copy_back(w, e->d_fit_win.p, sizeof(w), s));
    if (norm_map && !out_on_device)
        QM_HIP(copy_back(norm_map, d_norm, (size_t)n * sizeof(double), s));
    if (smoothed_map && !out_on_device)
        QM_HIP(copy_back(smoothed_map, d_smooth, (size_t)n * sizeof(double), s));
    QM_HIP(hipStreamSynchronize(s));
    if (h[1] < 0) return fail("qm_engine_locate_fits: the map holds no finite value");
    summary[0] = h[0];                      // nanmax of the input map
    summary[1] = h[1];                      // first argmax of the normalised map (flat index)
    summary[2] = h[6];                      // mean of the smoothed map
    summary[3] = h[7];                      // first argmax of the smoothed map
    summary[4] = h[8];                      // total weight above the threshold
    for (int k = 0; k < 3; ++k) summary[5 + k] = h[9 + k] / h[8];     // xe, ye, ze
    for (int k = 0; k < 6; ++k) summary[8 + k] = h[12 + k];
    summary[14] = h[2];
    summary[15] = h[4];
    std::memcpy(gau_window, w, 343 * sizeof(double));
    std::memcpy(spline_window, w + 343, 125 * sizeof(double));
    return 0;
}

int qm_engine_rbf_peak(qm_engine *e, const double *weights, int32_t n, int32_t upscale,
                       double *peak_value, int64_t *peak_index) {
    if (!e || !weights || !peak_value || !peak_index)
        return fail("qm_engine_rbf_peak: NULL argument");
    if (n < 2 || n > 9 || upscale < 1 || upscale > 64)
        return fail("qm_engine_rbf_peak: need 2 <= n <= 9 centres per axis and 1 <= upscale <= 64");
    DeviceGuard guard(e->device);
    const int m = (n - 1) * upscale + 1;
    const int64_t fine = (int64_t)m * m * m;
    constexpr int NB = qm::kFitBlocks, BS = qm::kFitBlock;
    if (e->d_fit_a.ensure((size_t)fine) || e->d_fit_win.ensure(9 * 9 * 9) ||
        e->d_fit_part.ensure((size_t)NB * 6) || e->d_fit_pidx.ensure(NB) || e->d_fit_val.ensure(32))
        return 1;
    hipStream_t s = e->stream;
    QM_HIP(copy_in(e->d_fit_win.p, weights, (size_t)n * n * n * sizeof(double), s));
    hipLaunchKernelGGL(qm::rbf_dense_kernel, dim3((unsigned)((fine + BS - 1) / BS)), dim3(BS), 0, s,
                       (const double *)e->d_fit_win.p, (int)n, m, (double)(n - 1) / (double)(m - 1),
                       e->d_fit_a.p);
    hipLaunchKernelGGL(qm::argmax_partial_kernel, dim3(NB), dim3(BS), 0, s,
                       (const double *)e->d_fit_a.p, fine, e->d_fit_part.p, e->d_fit_pidx.p);
    hipLaunchKernelGGL(qm::argmax_final_kernel, dim3(1), dim3(BS), 0, s, e->d_fit_part.p,
                       e->d_fit_pidx.p, NB, e->d_fit_val.p, e->d_fit_val.p + 1);
    QM_HIP(hipGetLastError());
    double h[2];
    QM_HIP(copy_back(h, e->d_fit_val.p, sizeof(h), s));
    QM_HIP(hipStreamSynchronize(s));
    if (h[1] < 0) return fail("qm_engine_rbf_peak: the interpolant holds no finite value");
    *peak_value = h[0];
    *peak_index = (int64_t)h[1];
    return 0;
}

}  // extern "C"
