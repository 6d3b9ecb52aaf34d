/*
 * qm_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement of the algorithm of the reference's native hot path
 * (QuakeMigrate v1.2.1).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker
 * or as the timed CPU baseline.  The shipped engine (quakemigrate_amd/csrc)
 * never links, loads or calls it.
 *
 * Parity status: PINNED.  oracle/make_golden.py runs the reference's own
 * quakemigrate/core/lib.py against a build of the reference C sources
 * (oracle/_ref/, see oracle/Makefile) and stores inputs + outputs under
 * tests/golden/; tests/test_oracle_golden.py checks this restatement against
 * those vectors (and against the known answers of the reference's
 * tests/test_onsets.py:27-35).
 *
 * What is restated (reference file:line):
 *   oq_stack      <- quakemigrate/core/src/migratelib.c:40-65   (migrate)
 *   oq_scan_max   <- quakemigrate/core/src/migratelib.c:85-111  (find_max_coa)
 *   oq_stalta_*   <- quakemigrate/core/src/onsetlib.c:35-59, 79-108, 126-148
 *
 * Build flags mirror the reference's (setup.py:119: -fopenmp -fPIC -Ofast) so
 * the floating-point behaviour (reciprocal multiply, libmvec exp) is the same
 * as the reference extension's.  All arithmetic is float64 / int32 / int64.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/*
 * Stack log-onset rows through the integer travel-time table.
 *
 *   log_onsets : [n_rows][row_len]  row_len = first + n_scan + last
 *   tt         : [n_nodes][n_rows]  samples of delay; negatives act as 0
 *   volume     : [n_nodes][n_scan]  accumulated INTO (caller zeroes it), then
 *                replaced by exp(sum / n_available)
 *
 * Rows are added in ascending row order for every (node, sample) -- that order
 * fixes the rounding of the float64 sums and therefore the argmax parity.
 */
void oq_stack(const double *log_onsets, const int32_t *tt, double *volume,
              int32_t first, int32_t last, int32_t n_scan, int32_t n_rows,
              int32_t n_available, int64_t n_nodes, int64_t n_threads)
{
    const int32_t row_len = first + last + n_scan;
    int64_t inode;

#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (inode = 0; inode < n_nodes; ++inode) {
        double *out = volume + inode * (int64_t)n_scan;
        const int32_t *delays = tt + inode * (int64_t)n_rows;
        for (int32_t r = 0; r < n_rows; ++r) {
            int32_t d = delays[r];
            if (d < 0) d = 0;                       /* migratelib.c:55 */
            /* int32 index arithmetic, as in the reference (:57-58) */
            const double *src = log_onsets + (r * row_len + d + first);
            for (int32_t k = 0; k < n_scan; ++k)
                out[k] += src[k];
        }
        for (int32_t k = 0; k < n_scan; ++k)
            out[k] = exp(out[k] / n_available);     /* migratelib.c:62 */
    }
}

/*
 * Per-sample scan over all nodes: maximum, index of the first node reaching
 * it (strict '>' => lowest index wins ties), and the normalised maximum
 * max * n_nodes / sum, the sum running sequentially in node order.
 */
void oq_scan_max(const double *volume, double *peak, double *peak_norm,
                 int64_t *peak_node, int32_t n_scan, int64_t n_nodes,
                 int64_t n_threads)
{
    int32_t k;

#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (k = 0; k < n_scan; ++k) {
        double best = volume[k];
        double total = best;
        int64_t where = 0;
        for (int64_t inode = 1; inode < n_nodes; ++inode) {
            const double v = volume[inode * (int64_t)n_scan + k];
            total += v;
            if (v > best) {
                best = v;
                where = inode;
            }
        }
        peak[k] = best;
        peak_norm[k] = best * n_nodes / total;     /* migratelib.c:108 */
        peak_node[k] = where;
    }
}

typedef struct {
    int n;
    int nsta;
    int nlta;
} oq_stalta_header;                                 /* qmlib.h:34-38 */

/* STA window is the tail of the LTA window; value sits on the last sample. */
void oq_stalta_overlapping(const double *x, const oq_stalta_header *h,
                           double *y)
{
    const int n = h->n, ns = h->nsta, nl = h->nlta;
    const double scale = (double)nl / (double)ns;
    double s_short = 0.0, s_long;
    int i;

    for (i = 0; i < ns; ++i)
        s_short += x[i];
    s_long = s_short;
    for (i = ns; i < nl; ++i) {
        s_long += x[i];
        s_short += x[i] - x[i - ns];
    }
    y[nl - 1] = s_short / s_long * scale;
    for (i = nl; i < n; ++i) {
        s_short += x[i] - x[i - ns];
        s_long += x[i] - x[i - nl];
        y[i] = s_short / s_long * scale;
    }
}

/* STA window follows the LTA window; value sits on the last LTA sample. */
void oq_stalta_centred(const double *x, const oq_stalta_header *h, double *y)
{
    const int n = h->n, ns = h->nsta, nl = h->nlta;
    const double scale = (double)nl / (double)ns;
    double s_short = 0.0, s_long = 0.0;
    int i;

    for (i = 0; i < nl; ++i)
        s_long += x[i];
    for (i = nl; i < nl + ns; ++i)
        s_short += x[i];
    y[nl - 1] = s_short / s_long * scale;
    for (i = nl; i < n - ns; ++i) {
        s_short += x[i + ns] - x[i];
        s_long += x[i] - x[i - nl];
        y[i] = (s_long > 0.0) ? s_short / s_long * scale : 1.0;
    }
}

/* Exponentially weighted STA and LTA; the first nlta outputs are nulled to 1. */
void oq_stalta_recursive(const double *x, const oq_stalta_header *h,
                         double *y)
{
    const int n = h->n, nl = h->nlta;
    const double a_short = 1.0 / (double)h->nsta;
    const double a_long = 1.0 / (double)nl;
    double s_short = 0.0, s_long = 0.0;
    int i;

    for (i = 1; i < n; ++i) {
        s_short = a_short * x[i] + (1 - a_short) * s_short;
        s_long = a_long * x[i] + (1 - a_long) * s_long;
        y[i] = s_short / s_long;
    }
    if (nl < n)
        for (i = 0; i < nl; ++i)
            y[i] = 1.0;
}
