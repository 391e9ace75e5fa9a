/* The reference's filter-design demo (fir/src/main.c:283-340): a multi-band magnitude
 * response on 100 Hz bands at 48 kHz -> filter kernel (backward DFT of 482 points, 32 taps,
 * Blackman window) -> the kernel's own magnitude response; one row per band:
 * "frequency wanted kernel.re kernel.im achieved". */
#define DSP_DROPIN_MAGNITUDE_FFT
#include <stdio.h>
#include <stdlib.h>

#include "dsp_dropin.h"

/* 0 below the first edge, then toggling between 1 and 0 at every edge (fir/src/main.c:108-135) */
static void multiband_response(double *resp, unsigned int n, const double *edges, unsigned int nedges, double fsampl)
{
    const double hz_per_band = fsampl / (((double)n - 1) * 2);
    double sign = -1;
    unsigned int band, next = 0;
    for (band = 0; band < n; ++band) {
        if (next < nedges && (double)band * hz_per_band >= edges[next]) {
            sign = -sign;
            ++next;
        }
        resp[band] = (1 + sign) / 2;
    }
}

int main(void)
{
    const double fsampl = 48000, fband = 100;
    const double edges[] = {1000, 2500, 3000, 3200, 6000, 10000, 14000, 20000};
    const unsigned int nbands = 1 + fsampl / (2 * fband); /* 241 */
    double *wanted = malloc(nbands * sizeof(double));
    double *achieved = malloc(nbands * sizeof(double));
    fftw_complex *kernel = malloc(nbands * 2 * sizeof(fftw_complex));
    unsigned int i;

    multiband_response(wanted, nbands, edges, sizeof(edges) / sizeof(edges[0]), fsampl);
    make_filter_kernel(wanted, nbands, kernel);
    fft(kernel, nbands * 2, achieved);
    for (i = 0; i < nbands; ++i)
        printf("%lf %lf %lf %lf %lf\n", (double)i * fband, wanted[i], kernel[i][0], kernel[i][1], achieved[i]);

    free(wanted);
    free(achieved);
    free(kernel);
    return 0;
}
