// qm_stalta.cpp -- host implementations of the three STA/LTA onset symbols the reference binds
// at import (quakemigrate/core/lib.py:173,211,249; prototypes qmlib.h:40-44).
//
// They sit UPSTREAM of the migrate/find_max_coa hot path, are serial O(n) per trace in the
// reference (quakemigrate/core/src/onsetlib.c:35-59, 79-108, 126-148) and stay host code here;
// a drop-in qmlib must export them or `import quakemigrate.core` fails.  Running sums are
// updated in the same operation order as the reference so the outputs agree to rounding.
#include "../../include/qmhip.h"

namespace {

struct Windows {
    int n, ns, nl;
    double ratio;   // nlta / nsta : turns the ratio of sums into a ratio of means
    explicit Windows(const stalta_header *h)
        : n(h->n), ns(h->nsta), nl(h->nlta), ratio((double)h->nlta / (double)h->nsta) {}
};

inline double sum_range(const double *x, int a, int b) {
    double s = 0.0;
    for (int i = a; i < b; ++i) s += x[i];
    return s;
}

}  // namespace

extern "C" {

// short window = last nsta samples of the long window; value on the window's last sample
void overlapping_sta_lta(const double *signal, const stalta_header *head, double *onset) {
    const Windows w(head);
    double shortw = sum_range(signal, 0, w.ns);
    double longw = shortw;
    for (int i = w.ns; i < w.nl; ++i) {
        const double in = signal[i];
        longw += in;
        shortw += in - signal[i - w.ns];
    }
    onset[w.nl - 1] = shortw / longw * w.ratio;
    for (int i = w.nl; i < w.n; ++i) {
        const double in = signal[i];
        shortw += in - signal[i - w.ns];
        longw += in - signal[i - w.nl];
        onset[i] = shortw / longw * w.ratio;
    }
}

// short window starts where the long window ends; value on the long window's last sample
void centred_sta_lta(const double *signal, const stalta_header *head, double *onset) {
    const Windows w(head);
    double longw = sum_range(signal, 0, w.nl);
    double shortw = sum_range(signal, w.nl, w.nl + w.ns);
    onset[w.nl - 1] = shortw / longw * w.ratio;
    const int stop = w.n - w.ns;
    for (int i = w.nl; i < stop; ++i) {
        shortw += signal[i + w.ns] - signal[i];
        longw += signal[i] - signal[i - w.nl];
        onset[i] = longw > 0.0 ? shortw / longw * w.ratio : 1.0;
    }
}

// exponentially weighted averages; the first nlta values are nulled to 1
void recursive_sta_lta(const double *signal, const stalta_header *head, double *onset) {
    const Windows w(head);
    const double ks = 1.0 / (double)w.ns, kl = 1.0 / (double)w.nl;
    double shortw = 0.0, longw = 0.0;
    for (int i = 1; i < w.n; ++i) {
        const double in = signal[i];
        shortw = ks * in + (1 - ks) * shortw;
        longw = kl * in + (1 - kl) * longw;
        onset[i] = shortw / longw;
    }
    if (w.nl < w.n)
        for (int i = 0; i < w.nl; ++i) onset[i] = 1.0;
}

}  // extern "C"
