/* The reference's own FFT wrappers, verbatim in shape, compiled against
 * dspmi_fftw3.h instead of <fftw3.h>: plan / execute / destroy per call as
 * fftc/main.c:31-43 does, the r2c copy loop of fft/src/main.c:146-168, and the
 * r2c -> gains -> c2r -> /n sequence of util/filter/filter.c:73-105.
 * Prints the 480-point spectrum, then a filtered 4096-sample block. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "dspmi_fftw3.h"
#include "tones.h"

static void dft(fftw_complex *in, unsigned int nx, fftw_complex *out)
{
    fftw_plan plan = fftw_plan_dft_1d(nx, in, out, FFTW_FORWARD, FFTW_ESTIMATE);
    fftw_execute(plan);
    fftw_destroy_plan(plan);
}

static void idft(fftw_complex *in, unsigned int nx, fftw_complex *out)
{
    fftw_plan plan = fftw_plan_dft_1d(nx, in, out, FFTW_BACKWARD, FFTW_ESTIMATE);
    fftw_execute(plan);
    fftw_destroy_plan(plan);
}

static void filter(double *xx, const double *x, unsigned int n, const double *coeffs)
{
    fftw_complex *out = fftw_malloc((n / 2 + 1) * sizeof(fftw_complex));
    fftw_plan plan = fftw_plan_dft_r2c_1d(n, (double *)x, out, FFTW_ESTIMATE);
    unsigned int i;
    fftw_execute(plan);
    fftw_destroy_plan(plan);
    for (i = 0; i != n / 2 + 1; ++i) {
        out[i][0] *= coeffs[i];
        out[i][1] *= coeffs[i];
    }
    plan = fftw_plan_dft_c2r_1d(n, out, xx, FFTW_ESTIMATE);
    fftw_execute(plan);
    for (i = 0; i != n; ++i) xx[i] /= (double)n;
    fftw_destroy_plan(plan);
    fftw_free(out);
}

int main(void)
{
    const unsigned int nx = 480, n = 4096;
    fftw_complex *x = fftw_malloc(nx * sizeof(fftw_complex));
    fftw_complex *y = fftw_malloc(nx * sizeof(fftw_complex));
    fftw_complex *z = fftw_malloc(nx * sizeof(fftw_complex));
    double *s = malloc(n * sizeof(double)), *f = malloc(n * sizeof(double)), *g = malloc((n / 2 + 1) * sizeof(double));
    unsigned int i;
    tones_t gen;

    tones_init(&gen);
    tones_add(&gen, 48000, 2000, 100, 0);
    tones_add(&gen, 48000, 4000, 100, 0.666);
    tones_add(&gen, 48000, 6000, 100, 0);
    tones_read_complex(&gen, x, nx);
    dft(x, nx, y);
    idft(y, nx, z);
    for (i = 0; i < nx; ++i) printf("%u %lf %lf %lf %lf\n", i, x[i][0], z[i][0] / (double)nx, y[i][0], y[i][1]);

    tones_init(&gen);
    tones_add(&gen, 48000, 1000, 1.0, 0);
    tones_add(&gen, 48000, 12000, 0.5, 0.3);
    tones_read(&gen, s, n);
    for (i = 0; i < n / 2 + 1; ++i) g[i] = (double)i * 48000.0 / n < 3000 ? 1.0 : 0.0;
    filter(f, s, n, g);
    for (i = 0; i < n; ++i) printf("%u %lf %lf\n", i, s[i], f[i]);
    fftw_free(x); fftw_free(y); fftw_free(z); free(s); free(f); free(g);
    return 0;
}
