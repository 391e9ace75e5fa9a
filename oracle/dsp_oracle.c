/*
 * dsp_oracle.c -- CPU restatement of the texane/dsp compute routines.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
 * calls this file; only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py use it, and only as the checker / the timed CPU baseline.
 *
 * Every routine restates, in plain double-precision C, the arithmetic of one
 * routine of the reference, in the reference's own order of floating-point
 * operations (build with -ffp-contract=off so no FMA contraction changes it).
 *
 * Parity status:
 *   - FIR / convolution routines: PINNED.  Checked against the reference's own
 *     convolution/main.c compiled in place (oracle/_ref, see oracle/Makefile)
 *     and against that program's printed 9-point table.
 *   - FFT routines: PARITY UNPINNED.  The reference delegates every transform
 *     to FFTW3 (bare -lfftw3, no version pinned; fftc/do_build.sh:2), which is
 *     not installed here and whose source is not part of the reference.  The
 *     oracle is the DFT that FFTW documents and the reference's call sites
 *     rely on: unnormalised, forward exponent -2*pi*i*n*k/N, backward +,
 *     natural order, r2c returning bins 0..n/2, c2r(r2c(x)) == n*x
 *     (relied upon at util/filter/filter.c:102).  It is cross-checked against
 *     numpy's pocketfft in tests/, not against FFTW.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* FIR / convolution                                                   */
/* ------------------------------------------------------------------ */

/* Input-side ("scatter") convolution truncated to nx outputs.
 * Follows convolution/main.c:15-33: y is cleared, then every x[i] adds
 * x[i]*h[j] into y[i+j] for j < min(nh, nx-i); a given y[k] therefore
 * receives its terms in ascending i. */
void oracle_conv0(const double *x, unsigned nx, const double *h, unsigned nh,
                  double *y)
{
    unsigned i, j;
    memset(y, 0, (size_t)nx * sizeof *y);
    for (i = 0; i != nx; i++) {
        unsigned span = nx - i;
        if (nh < span)
            span = nh;
        for (j = 0; j != span; j++)
            y[i + j] += x[i] * h[j];
    }
}

/* Output-side ("gather") convolution, bug-compatible.
 * Follows convolution/main.c:36-53: the tap count is min(i, nh), so while
 * i < nh the j == i term (x[0]*h[i]) is never added. */
void oracle_conv1(const double *x, unsigned nx, const double *h, unsigned nh,
                  double *y)
{
    unsigned i, j;
    for (i = 0; i != nx; i++) {
        unsigned taps = i < nh ? i : nh;
        y[i] = 0;
        for (j = 0; j != taps; j++)
            y[i] += x[i - j] * h[j];
    }
}

/* Gather convolution into a caller-owned buffer of nxx outputs.
 * Follows fir/src/main.c:258-280 (== alsa/main.c:494-516): taps in
 * ascending j, terms whose input index falls outside [0, nx) skipped,
 * zero history before x[0]. */
void oracle_convolve(double *xx, unsigned nxx, const double *x, unsigned nx,
                     const double *h, unsigned nh)
{
    long i, j;
    for (i = 0; i < (long)nxx; i++) {
        xx[i] = 0;
        for (j = 0; j < (long)nh; j++) {
            long k = i - j;
            if (k < 0 || k >= (long)nx)
                continue;
            xx[i] += h[j] * x[k];
        }
    }
}

/* Allocating form, follows fft/src/main.c:180-205: the result has nx+nh
 * elements (one more than the mathematical nx+nh-1; the last is 0) and is
 * malloc'd for the caller to free. */
void oracle_convolve_alloc(double **xx, unsigned *nxx, const double *x,
                           unsigned nx, const double *h, unsigned nh)
{
    *nxx = nx + nh;
    *xx = (double *)malloc((size_t)*nxx * sizeof(double));
    oracle_convolve(*xx, *nxx, x, nx, h, nh);
}

/* Batched checker for the f32 device API: the same sum as oracle_convolve,
 * accumulated in double from f32 inputs, one channel after the other.
 * x is [nchan][x_stride], y is [nchan][y_stride] (strides in elements).
 * drop_first != 0 reproduces oracle_conv1 (x[0] of each channel never
 * contributes). */
void oracle_fir_batch_f32(double *y, size_t y_stride, unsigned nxx,
                          const float *x, size_t x_stride, unsigned nx,
                          const float *h, unsigned nh, unsigned nchan,
                          int drop_first)
{
    unsigned c;
    long i, j;
    for (c = 0; c != nchan; c++) {
        const float *xc = x + (size_t)c * x_stride;
        double *yc = y + (size_t)c * y_stride;
        for (i = 0; i < (long)nxx; i++) {
            double acc = 0;
            for (j = 0; j < (long)nh; j++) {
                long k = i - j;
                if (k < 0 || k >= (long)nx)
                    continue;
                if (drop_first && k == 0)
                    continue;
                acc += (double)h[j] * (double)xc[k];
            }
            yc[i] = acc;
        }
    }
}

/* Streaming variant: history carried between chunks.  hist holds the nh-1
 * samples preceding x[0] (oldest first).  Used to check the carried-halo
 * mode, which the reference does not have (alsa/main.c:530 restarts from a
 * zero state every period). */
void oracle_fir_stream_f32(double *y, const float *x, unsigned nx,
                           const float *hist, const float *h, unsigned nh)
{
    long i, j;
    for (i = 0; i < (long)nx; i++) {
        double acc = 0;
        for (j = 0; j < (long)nh; j++) {
            long k = i - j;
            double v = k >= 0 ? (double)x[k] : (double)hist[(long)nh - 1 + k];
            acc += (double)h[j] * v;
        }
        y[i] = acc;
    }
}

/* ------------------------------------------------------------------ */
/* DFT (see "PARITY UNPINNED" above)                                   */
/* ------------------------------------------------------------------ */

/* Direct O(n^2) DFT in long double: the ground truth for n <= a few
 * thousand.  in/out are interleaved re,im doubles (FFTW's fftw_complex
 * layout, fftc/main.c:140-144).  sign = -1 forward (FFTW_FORWARD),
 * +1 backward (FFTW_BACKWARD); no scaling either way
 * (fftc/main.c:31-43). */
void oracle_dft_direct(const double *in, unsigned n, double *out, int sign)
{
    unsigned k, t;
    long double *cs = (long double *)malloc((size_t)n * 2 * sizeof *cs);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (t = 0; t != n; t++) {
        long double a = two_pi * (long double)t / (long double)n;
        cs[2 * t] = cosl(a);
        cs[2 * t + 1] = sinl(a);
    }
    for (k = 0; k != n; k++) {
        long double sr = 0, si = 0;
        for (t = 0; t != n; t++) {
            unsigned idx = (unsigned)(((unsigned long long)k * t) % n);
            long double c = cs[2 * idx];
            long double s = sign < 0 ? -cs[2 * idx + 1] : cs[2 * idx + 1];
            long double xr = in[2 * t], xi = in[2 * t + 1];
            sr += xr * c - xi * s;
            si += xr * s + xi * c;
        }
        out[2 * k] = (double)sr;
        out[2 * k + 1] = (double)si;
    }
    free(cs);
}

static unsigned ilog2u(unsigned n)
{
    unsigned l = 0;
    while ((1u << l) < n)
        l++;
    return l;
}

/* Iterative radix-2 FFT in double, n a power of two: bit-reversal
 * reordering followed by log2(n) butterfly stages, the algorithm the
 * reference describes in fft/doc/notes:8-22.  Out of place, natural order
 * in and out, unnormalised.  This is also the timed CPU baseline for the
 * FFT workloads (kind "port": FFTW itself is not available). */
void oracle_fft_radix2(const double *in, unsigned n, double *out, int sign)
{
    unsigned bits = ilog2u(n), i, len;
    for (i = 0; i != n; i++) {
        unsigned r = 0, b;
        for (b = 0; b != bits; b++)
            r |= ((i >> b) & 1u) << (bits - 1 - b);
        out[2 * r] = in[2 * i];
        out[2 * r + 1] = in[2 * i + 1];
    }
    for (len = 2; len <= n; len <<= 1) {
        unsigned half = len >> 1, base, k;
        double step = (sign < 0 ? -2.0 : 2.0) * M_PI / (double)len;
        for (base = 0; base < n; base += len) {
            for (k = 0; k != half; k++) {
                double wr = cos(step * (double)k), wi = sin(step * (double)k);
                double *a = out + 2 * (base + k);
                double *b2 = out + 2 * (base + k + half);
                double tr = b2[0] * wr - b2[1] * wi;
                double ti = b2[0] * wi + b2[1] * wr;
                b2[0] = a[0] - tr;
                b2[1] = a[1] - ti;
                a[0] += tr;
                a[1] += ti;
            }
        }
    }
}

/* Same transform with the twiddles taken from a table built once
 * (n/2 entries of exp(sign*2*pi*i*k/n)); numerically equivalent to the
 * routine above up to the table's rounding, and several times faster:
 * this is the "fair" single-core CPU baseline. */
void oracle_fft_radix2_table(const double *in, unsigned n, double *out,
                             const double *tw /* n/2 pairs */)
{
    unsigned bits = ilog2u(n), i, len;
    for (i = 0; i != n; i++) {
        unsigned r = 0, b;
        for (b = 0; b != bits; b++)
            r |= ((i >> b) & 1u) << (bits - 1 - b);
        out[2 * r] = in[2 * i];
        out[2 * r + 1] = in[2 * i + 1];
    }
    for (len = 2; len <= n; len <<= 1) {
        unsigned half = len >> 1, stride = n / len, base, k;
        for (base = 0; base < n; base += len) {
            for (k = 0; k != half; k++) {
                double wr = tw[2 * k * stride], wi = tw[2 * k * stride + 1];
                double *a = out + 2 * (base + k);
                double *b2 = out + 2 * (base + k + half);
                double tr = b2[0] * wr - b2[1] * wi;
                double ti = b2[0] * wi + b2[1] * wr;
                b2[0] = a[0] - tr;
                b2[1] = a[1] - ti;
                a[0] += tr;
                a[1] += ti;
            }
        }
    }
}

void oracle_fft_make_table(double *tw, unsigned n, int sign)
{
    unsigned k;
    for (k = 0; k != n / 2; k++) {
        double a = (sign < 0 ? -2.0 : 2.0) * M_PI * (double)k / (double)n;
        tw[2 * k] = cos(a);
        tw[2 * k + 1] = sin(a);
    }
}

/* Complex DFT of any length: radix-2 when n is a power of two and large,
 * direct otherwise. */
void oracle_dft(const double *in, unsigned n, double *out, int sign)
{
    if (n >= 8 && (n & (n - 1)) == 0 && n > 4096)
        oracle_fft_radix2(in, n, out, sign);
    else
        oracle_dft_direct(in, n, out, sign);
}

/* Real-input forward transform, bins 0..n/2, interleaved re,im.
 * The contract of fftw_plan_dft_r2c_1d as used by fft/src/main.c:146-168
 * and util/fft/fft.c:99-121 (which copy the bins into xx[2*(n/2+1)]). */
void oracle_fft_r2c(double *xx, const double *x, unsigned n)
{
    unsigned i;
    double *cin = (double *)calloc((size_t)n * 2, sizeof *cin);
    double *cout = (double *)malloc((size_t)n * 2 * sizeof *cout);
    for (i = 0; i != n; i++)
        cin[2 * i] = x[i];
    oracle_dft(cin, n, cout, -1);
    memcpy(xx, cout, (size_t)(n / 2 + 1) * 2 * sizeof *xx);
    free(cin);
    free(cout);
}

/* Complex-to-real backward transform from bins 0..n/2 (Hermitian
 * extension), unnormalised: the contract of fftw_plan_dft_c2r_1d
 * (util/filter/filter.c:100-101).  Like FFTW, the imaginary parts of bin 0
 * (and of bin n/2 when n is even) are ignored. */
void oracle_fft_c2r(double *x, const double *bins, unsigned n)
{
    unsigned i;
    double *cin = (double *)malloc((size_t)n * 2 * sizeof *cin);
    double *cout = (double *)malloc((size_t)n * 2 * sizeof *cout);
    for (i = 0; i <= n / 2; i++) {
        cin[2 * i] = bins[2 * i];
        cin[2 * i + 1] = bins[2 * i + 1];
    }
    cin[1] = 0;
    if ((n & 1u) == 0)
        cin[2 * (n / 2) + 1] = 0;
    for (i = n / 2 + 1; i < n; i++) {
        cin[2 * i] = bins[2 * (n - i)];
        cin[2 * i + 1] = -bins[2 * (n - i) + 1];
    }
    oracle_dft(cin, n, cout, +1);
    for (i = 0; i != n; i++)
        x[i] = cout[2 * i];
    free(cin);
    free(cout);
}

/* Frequency-domain filter: r2c, multiply bin i by the real gain
 * coeffs[i], c2r, divide by n.  Follows util/filter/filter.c:73-105. */
void oracle_filter(double *xx, const double *x, unsigned n,
                   const double *coeffs)
{
    unsigned i, nb = n / 2 + 1;
    double *bins = (double *)malloc((size_t)nb * 2 * sizeof *bins);
    oracle_fft_r2c(bins, x, n);
    for (i = 0; i != nb; i++) {
        bins[2 * i] *= coeffs[i];
        bins[2 * i + 1] *= coeffs[i];
    }
    oracle_fft_c2r(xx, bins, n);
    for (i = 0; i != n; i++)
        xx[i] /= (double)n;
    free(bins);
}

/* ------------------------------------------------------------------ */
/* Spectrum post-ops                                                   */
/* ------------------------------------------------------------------ */

/* Magnitude spectrum normalised by its sum when the sum exceeds 1e-5.
 * Follows fft/src/main.c:120-141 (== util/fft/fft.c:73-94). */
void oracle_power_spectrum(double *ps, const double *xx, unsigned nxx)
{
    double sum = 0;
    unsigned i;
    for (i = 0; i != nxx; i++) {
        double re = xx[2 * i], im = xx[2 * i + 1];
        double p = sqrt(re * re + im * im);
        ps[i] = p;
        sum += p;
    }
    if (sum > 0.00001)
        for (i = 0; i != nxx; i++)
            ps[i] /= sum;
}

/* Absolute amplitude spectrum 2|X|/nx over nx/2 bins.
 * Follows fftc/main.c:80-105 and fir/src/main.c:246-251. */
void oracle_ps_abs(const double *x, unsigned nx, double *xx)
{
    unsigned i;
    for (i = 0; i != nx / 2; i++) {
        double re = x[2 * i], im = x[2 * i + 1];
        xx[i] = (2 * sqrt(re * re + im * im)) / (double)nx;
    }
}

/* re - im over nx/2 bins, divided by the sum when the sum >= 1e-4.
 * Follows fftc/main.c:46-77. */
void oracle_ps_percent(const double *x, unsigned nx, double *xx)
{
    unsigned i, nb = nx / 2;
    double sum = 0;
    for (i = 0; i != nb; i++) {
        xx[i] = x[2 * i] - x[2 * i + 1];
        sum += xx[i];
    }
    if (sum >= 0.0001)
        for (i = 0; i != nb; i++)
            xx[i] /= sum;
}

/* Phase atan(im/re) where |re| >= 1e-5, else 0.  Follows fftc/main.c:108-123. */
void oracle_compute_phi(const double *x, unsigned nx, double *phi)
{
    unsigned i;
    for (i = 0; i != nx / 2; i++) {
        if (fabs(x[2 * i]) >= 0.00001)
            phi[i] = atan(x[2 * i + 1] / x[2 * i]);
        else
            phi[i] = 0;
    }
}

/* |X| over nx = nsampl/2 bins divided by their sum, no zero guard.
 * Follows alsa/main.c:473-487. */
void oracle_alsa_power_spectrum(double *ps, const double *xx, unsigned nbin)
{
    double sum = 0;
    unsigned i;
    for (i = 0; i != nbin; i++) {
        double re = xx[2 * i], im = xx[2 * i + 1];
        ps[i] = sqrt(re * re + im * im);
        sum += ps[i];
    }
    for (i = 0; i != nbin; i++)
        ps[i] /= sum;
}

/* ------------------------------------------------------------------ */
/* Tone generator (input synthesis for the demos and the chain)        */
/* ------------------------------------------------------------------ */

#define ORACLE_TONEGEN_MAX 32

typedef struct oracle_tonegen {
    unsigned n;
    double a[ORACLE_TONEGEN_MAX];
    double phi[ORACLE_TONEGEN_MAX];
    double w[ORACLE_TONEGEN_MAX];
    double dw[ORACLE_TONEGEN_MAX];
} oracle_tonegen_t;

/* Follows tonegen/tonegen.c:5-23 (argument order: fsampl, freq, a, phi). */
void oracle_tonegen_init(oracle_tonegen_t *g) { g->n = 0; }

void oracle_tonegen_add(oracle_tonegen_t *g, double fsampl, double freq,
                        double a, double phi)
{
    unsigned i = g->n++;
    g->w[i] = 0.0;
    g->a[i] = a;
    g->phi[i] = phi;
    g->dw[i] = (2.0 * M_PI * freq) / fsampl;
}

/* Sum of a*cos(w+phi) with the running phase recurrence w += dw.
 * Follows tonegen/tonegen.c:25-40. */
void oracle_tonegen_read(oracle_tonegen_t *g, double *buf, unsigned n)
{
    unsigned i, j;
    for (i = 0; i != n; i++) {
        buf[i] = 0;
        for (j = 0; j != g->n; j++) {
            buf[i] += g->a[j] * cos(g->w[j] + g->phi[j]);
            g->w[j] += g->dw[j];
        }
    }
}

/* Complex form: real part as above, imaginary part 0.
 * Follows tonegen/tonegen.c:42-59. */
void oracle_tonegen_read_complex(oracle_tonegen_t *g, double *buf, unsigned n)
{
    unsigned i, j;
    for (i = 0; i != n; i++) {
        buf[2 * i] = 0;
        buf[2 * i + 1] = 0;
        for (j = 0; j != g->n; j++) {
            buf[2 * i] += g->a[j] * cos(g->w[j] + g->phi[j]);
            g->w[j] += g->dw[j];
        }
    }
}

/* Unit-amplitude sine variant with dw = 2*pi*freq/fsampl.
 * Follows fft/src/main.c:265-291. */
void oracle_sin_tonegen_read(double *w, const double *dw, unsigned ntone,
                             double *buf, unsigned n)
{
    unsigned i, j;
    for (i = 0; i != n; i++) {
        buf[i] = 0;
        for (j = 0; j != ntone; j++) {
            buf[i] += sin(w[j]);
            w[j] += dw[j];
        }
    }
}

/* ------------------------------------------------------------------ */
/* s16 stereo <-> mono helpers (alsa/main.c:464-468, 527-539)           */
/* ------------------------------------------------------------------ */

void oracle_s16x2_to_mono(double *mono, const short *buf, unsigned nsampl)
{
    unsigned i;
    for (i = 0; i != nsampl; i++)
        mono[i] = ((double)buf[2 * i] + (double)buf[2 * i + 1]) / 2;
}

void oracle_mono_to_s16x2(short *buf, const double *mono, unsigned nsampl)
{
    unsigned i;
    for (i = 0; i != nsampl; i++) {
        short v = (short)mono[i];
        buf[2 * i] = v;
        buf[2 * i + 1] = v;
    }
}

/* One capture block through the FIR stage: stereo s16 -> mono double ->
 * convolve over nsampl+nh outputs with zero initial state -> the first
 * nsampl results cast to s16 into both channels.  `out` may alias `in`.
 * Follows do_fir, alsa/main.c:518-540. */
void oracle_alsa_do_fir(short *out, const short *in, unsigned nsampl,
                        const double *h, unsigned nh)
{
    double *mono = (double *)malloc((size_t)nsampl * sizeof *mono);
    double *y = (double *)malloc(((size_t)nsampl + nh) * sizeof *y);
    oracle_s16x2_to_mono(mono, in, nsampl);
    oracle_convolve(y, nsampl + nh, mono, nsampl, h, nh);
    oracle_mono_to_s16x2(out, y, nsampl);
    free(mono);
    free(y);
}

/* One capture block through the spectrum stage: stereo s16 -> complex
 * (mono, 0) -> forward DFT of nsampl points -> |X| over nsampl/2 bins
 * divided by their sum.  Follows do_power_spectrum, alsa/main.c:453-492. */
void oracle_alsa_do_power_spectrum(double *ps, const short *in, unsigned nsampl)
{
    double *z = (double *)malloc((size_t)nsampl * 2 * sizeof *z);
    double *zz = (double *)malloc((size_t)nsampl * 2 * sizeof *zz);
    unsigned i;
    for (i = 0; i != nsampl; i++) {
        z[2 * i] = ((double)in[2 * i] + (double)in[2 * i + 1]) / 2;
        z[2 * i + 1] = 0;
    }
    oracle_dft(z, nsampl, zz, -1);
    oracle_alsa_power_spectrum(ps, zz, nsampl / 2);
    free(z);
    free(zz);
}

/* Generated samples -> stereo s16, the cast of tonegen/main.c:14-24. */
void oracle_double_to_s16x2(short *raw, const double *gen, unsigned n)
{
    oracle_mono_to_s16x2(raw, gen, n);
}

/* ------------------------------------------------------------------ */
/* Filter-kernel designer (fir/src/main.c:108-226)                      */
/* ------------------------------------------------------------------ */

/* Pass bands alternate at every cut frequency, starting in a stop band.
 * Follows make_multipass_fresp, fir/src/main.c:108-135. */
void oracle_make_multipass_fresp(double *x, unsigned nx, const double *fcuts,
                                 unsigned ncuts, double fsampl)
{
    const double fband = fsampl / (((double)nx - 1) * 2);
    double level = -1;
    unsigned i, j = 0;
    for (i = 0; i != nx; i++) {
        if (j < ncuts && (double)i * fband >= fcuts[j]) {
            level *= -1;
            j++;
        }
        x[i] = (1 + level) / 2;
    }
}

/* Follows make_blackman_window, fir/src/main.c:138-146 (denominator nx). */
void oracle_make_blackman_window(double *x, unsigned nx)
{
    unsigned i;
    for (i = 0; i != nx; i++)
        x[i] = 0.42 - 0.5 * cos((2.0 * M_PI * (double)i) / (double)nx)
             + 0.08 * cos((4 * M_PI * (double)i) / (double)nx);
}

/* Follows make_filter_kernel, fir/src/main.c:149-226: backward DFT of the
 * mirrored half response, first nk/2 samples mirrored into nk taps, window. */
void oracle_make_filter_kernel(const double *fresp, unsigned nx, double *kernel)
{
    const unsigned nxx = nx * 2;
    const unsigned nk = nxx < 32 ? nxx : 32;
    const unsigned nkk = nk / 2;
    double *in = (double *)calloc((size_t)nxx * 2, sizeof *in);
    double *out = (double *)malloc((size_t)nxx * 2 * sizeof *out);
    double *w = (double *)malloc((size_t)nk * sizeof *w);
    unsigned i;
    for (i = 0; i != nx; i++)
        in[2 * i] = fresp[i] / 2;
    for (i = 0; i != nx; i++) {
        in[2 * (nx + i)] = in[2 * (nx - i - 1)];
        in[2 * (nx + i) + 1] = in[2 * (nx - i - 1) + 1];
    }
    oracle_dft(in, nxx, out, +1);
    for (i = 0; i != nkk; i++) {
        kernel[2 * (nkk - i - 1)] = out[2 * i];
        kernel[2 * (nkk - i - 1) + 1] = out[2 * i + 1];
        kernel[2 * (nkk + i)] = out[2 * i];
        kernel[2 * (nkk + i) + 1] = out[2 * i + 1];
    }
    for (i = nk; i != nxx; i++)
        kernel[2 * i] = kernel[2 * i + 1] = 0;
    oracle_make_blackman_window(w, nk);
    for (i = 0; i != nk; i++) {
        kernel[2 * i] *= w[i];
        kernel[2 * i + 1] *= w[i];
    }
    free(in);
    free(out);
    free(w);
}

/* ------------------------------------------------------------------ */
/* CPU-baseline timing loops (bench.py cpu_baseline leg)                */
/* ------------------------------------------------------------------ */

/* nframes transforms of length n, in -> out, frames back to back; one
 * thread.  with_table selects the table variant. */
void oracle_fft_batch(const double *in, double *out, unsigned n,
                      unsigned nframes, int sign, int with_table)
{
    unsigned f;
    if (with_table) {
        double *tw = (double *)malloc((size_t)n * sizeof *tw);
        oracle_fft_make_table(tw, n, sign);
        /* frames are independent: with -fopenmp (the _fast build) they are
         * spread over oracle_set_threads() threads for the all-cores baseline */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (f = 0; f < nframes; f++)
            oracle_fft_radix2_table(in + (size_t)f * 2 * n, n,
                                    out + (size_t)f * 2 * n, tw);
        free(tw);
    } else {
        for (f = 0; f != nframes; f++)
            oracle_fft_radix2(in + (size_t)f * 2 * n, n,
                              out + (size_t)f * 2 * n, sign);
    }
}

/* nchan channels of oracle_convolve, channel-major, one thread. */
void oracle_convolve_batch(double *xx, unsigned nxx, const double *x,
                           unsigned nx, const double *h, unsigned nh,
                           unsigned nchan)
{
    unsigned c;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (c = 0; c < nchan; c++)
        oracle_convolve(xx + (size_t)c * nxx, nxx, x + (size_t)c * nx, nx, h,
                        nh);
}

/* Threads the two batch loops above use in an OpenMP build (1 otherwise). */
#ifdef _OPENMP
#include <omp.h>
int oracle_set_threads(int n)
{
    if (n > 0)
        omp_set_num_threads(n);
    return omp_get_max_threads();
}
#else
int oracle_set_threads(int n)
{
    (void)n;
    return 1;
}
#endif
