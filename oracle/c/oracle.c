/*
 * C twin of oracle/np_oracle.py -- TEST INFRASTRUCTURE ONLY (never linked into libprcore.so).
 *
 * Plain-C restatement of the two loops of the reference that are too slow in Python for
 * medium-size checks.  Pinned by tests/test_oracle_golden.py against the same golden vectors
 * as the NumPy oracle.
 *
 *   orc_nlms          clutter_removal.py:189-249  (NLMS_filter; complex64 state, sequential)
 *   orc_caf_segments  range_doppler_processing.py:81-86 with the boxcar decimator (:72):
 *                     y[j,k] = sum_{n=jq-ceil(q/2)}^{jq+floor(q/2)} w[n] ref[n] conj(srv[(n+R-k) mod N])
 *   orc_sosfilt       the serial recursion under signal.decimate(x, q) in find_channel_offset
 *                     (signal_utils.py:73-78), complex128
 */
#include <complex.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float complex c64;
typedef double complex c128;

/* u_k[i] = ref[T+k-i]; e = d - w^H u; w += mu*u*conj(e)/(u^H u); out[L+k] = e   (:211-215,:234-244) */
int orc_nlms(const c64* ref, const c64* srv, int64_t n, int L, int peek, float mu,
             const c64* taps_in, c64* out, c64* taps_out) {
    const int T = L + peek;
    c64* w = (c64*)calloc((size_t)T, sizeof(c64));
    if (!w) return -1;
    if (taps_in) memcpy(w, taps_in, sizeof(c64) * (size_t)T);
    memset(out, 0, sizeof(c64) * (size_t)n);
    for (int64_t k = 0; k < n - T; ++k) {
        const c64* top = ref + T + k;          /* u[i] = top[-i] */
        c64 y = 0;
        float en = 0.f;
        for (int i = 0; i < T; ++i) {
            const c64 u = top[-i];
            y += conjf(w[i]) * u;
            en += crealf(u) * crealf(u) + cimagf(u) * cimagf(u);
        }
        const c64 e = srv[k + L] - y;
        const c64 ec = conjf(e);
        for (int i = 0; i < T; ++i) {
            const c64 u = top[-i];
            w[i] += (mu * u) * ec / en;
        }
        out[L + k] = e;
    }
    if (taps_out) memcpy(taps_out, w, sizeof(c64) * (size_t)T);
    free(w);
    return 0;
}

/* y is [F][R+1] complex128, products rounded to complex64 like the reference's array arithmetic */
int orc_caf_segments(const c64* ref, const c64* srv, int64_t n, int R, int F, const double* window,
                     c128* y) {
    const int64_t q = n / F;
    if (q <= 0) return -1;
    const int64_t up = q / 2, dn = q - up;   /* window [jq-dn, jq+up] */
#pragma omp parallel for schedule(static)
    for (int k = 0; k <= R; ++k) {
        const int64_t ell = R - k;
        for (int j = 0; j < F; ++j) {
            int64_t lo = (int64_t)j * q - dn, hi = (int64_t)j * q + up;
            if (lo < 0) lo = 0;
            if (hi > n - 1) hi = n - 1;
            c128 acc = 0;
            for (int64_t m = lo; m <= hi; ++m) {
                int64_t s = m + ell;
                if (s >= n) s -= n;
                c64 p = ref[m] * conjf(srv[s]);
                if (window) p = (c64)((c128)p * window[m]);
                acc += (c128)p;
            }
            y[(int64_t)j * (R + 1) + k] = acc;
        }
    }
    return 0;
}

/* One direction of scipy.signal.sosfilt (transposed direct form II, _sosfilt.pyx) as sosfiltfilt runs it
 * inside signal.decimate(x, q) -- signal_utils.py:75-76.  sos: nsec x 6 (b0 b1 b2 1 a1 a2), x filtered
 * in place, zi: nsec x 2 state carried in and out. */
int orc_sosfilt(const double* sos, int nsec, c128* x, int64_t n, c128* zi) {
    for (int64_t i = 0; i < n; ++i) {
        c128 cur = x[i];
        for (int s = 0; s < nsec; ++s) {
            const double* c = sos + 6 * s;
            c128 nw = c[0] * cur + zi[2 * s];
            zi[2 * s] = c[1] * cur - c[4] * nw + zi[2 * s + 1];
            zi[2 * s + 1] = c[2] * cur - c[5] * nw;
            cur = nw;
        }
        x[i] = cur;
    }
    return 0;
}
