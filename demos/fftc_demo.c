/* The reference's complex-DFT demo (fftc/main.c:125-189): three tones, forward and
 * backward transform of 480 points (not a power of two), phase and amplitude
 * spectrum, printed as "f amplitude re/nbands -im/nbands phi".  The tone arguments
 * are passed exactly as the reference passes them (fftc/main.c:151-153: frequency
 * first, sampling rate second -- swapped with respect to tonegen_add's parameter
 * order, so every tone lands on a multiple of 2*pi and the signal is constant). */
#define DSP_DROPIN_MAGNITUDE_FFT
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "dsp_dropin.h"
#include "tones.h"

int main(void)
{
    const double fsampl = 48000;
    double fband = 100;
    unsigned int nbands = (unsigned int)(fsampl / (2 * fband));
    unsigned int nx, nw, i;
    fftw_complex *x, *y, *z;
    double *w, *phi;
    tones_t gen;

    if (nbands & 1) {
        ++nbands;
        fband = fsampl / (2 * (double)nbands);
    }
    nx = nbands * 2;
    nw = nbands;
    x = malloc(nx * sizeof(fftw_complex));
    y = malloc(nx * sizeof(fftw_complex));
    z = malloc(nx * sizeof(fftw_complex));
    w = malloc(nw * sizeof(double));
    phi = malloc(nw * sizeof(double));

    tones_init(&gen);
    tones_add(&gen, 2000, fsampl, 100, 0);
    tones_add(&gen, 4000, fsampl, 100, 0.666);
    tones_add(&gen, 6000, fsampl, 100, 0);
    tones_read_complex(&gen, x, nx);

    dft(x, nx, y);
    idft(y, nx, z);
    /* fftc/main.c:108-123 (compute_phi) and :80-105 (ps_abs), on the host as there */
    for (i = 0; i < nw; ++i) phi[i] = fabs(y[i][0]) >= 0.00001 ? atan(y[i][1] / y[i][0]) : 0;
    for (i = 0; i < nw; ++i) w[i] = (2 * sqrt(y[i][0] * y[i][0] + y[i][1] * y[i][1])) / (double)nx;

    for (i = 0; i < nw; ++i)
        printf("%lf %lf %lf %lf %lf\n", fband * (double)i, w[i], y[i][0] / (double)nbands,
               -1 * y[i][1] / (double)nbands, phi[i]);
    /* the round trip the reference computes and discards (fftc/main.c:157,164) */
    {
        double err = 0;
        for (i = 0; i < nx; ++i) err = fmax(err, fabs(z[i][0] / (double)nx - x[i][0]));
        fprintf(stderr, "round trip max error %g\n", err);
    }
    free(x); free(y); free(z); free(w); free(phi);
    return 0;
}
