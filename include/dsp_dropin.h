/*
 * dsp_dropin.h -- the reference's own routine names over libdspmi.so.
 *
 * Include this instead of keeping the static compute routines in a demo's
 * main.c; the mains then compile unchanged.  Two names are defined twice in the
 * reference with different signatures, so the including file picks one:
 *
 *   convolve:  default = caller-allocated form (fir/src/main.c:258-263,
 *              alsa/main.c:494-499); define DSP_DROPIN_ALLOCATING_CONVOLVE
 *              before including for fft/src/main.c:180-185.
 *   fft:       default = r2c form (fft/src/main.c:146-147, util/fft/fft.c:99-100);
 *              define DSP_DROPIN_MAGNITUDE_FFT for fir/src/main.c:229-230.
 */
#ifndef DSP_DROPIN_H_INCLUDED
#define DSP_DROPIN_H_INCLUDED

#include "dspmi.h"

#ifndef DSP_DROPIN_HAVE_FFTW_COMPLEX
typedef double fftw_complex[2]; /* FFTW's layout, as used at fftc/main.c:140-144 */
#endif

/* convolution/main.c:15-53 */
static inline void conv0(const double *x, unsigned int nx, const double *h, unsigned int nh, double *y)
{
    dsp_conv0(x, nx, h, nh, y);
}
static inline void conv1(const double *x, unsigned int nx, const double *h, unsigned int nh, double *y)
{
    dsp_conv1(x, nx, h, nh, y);
}

#ifdef DSP_DROPIN_ALLOCATING_CONVOLVE
/* fft/src/main.c:180-244 */
static inline void convolve(double **xx, unsigned int *nxx, const double *x, unsigned int nx, const double *h,
                            unsigned int nh)
{
    dsp_convolve_alloc(xx, nxx, x, nx, h, nh);
}
static inline void fir(double **xx, unsigned int *nxx, const double *x, unsigned int nx)
{
    dsp_fir(xx, nxx, x, nx);
}
#else
/* fir/src/main.c:258-280, alsa/main.c:494-516 */
static inline void convolve(double *xx, unsigned int nxx, const double *x, unsigned int nx, const double *h,
                            unsigned int nh)
{
    dsp_convolve(xx, nxx, x, nx, h, nh);
}
#endif

/* fftc/main.c:31-43 */
static inline void dft(fftw_complex *in, unsigned int nx, fftw_complex *out) { dsp_dft(in, nx, out); }
static inline void idft(fftw_complex *in, unsigned int nx, fftw_complex *out) { dsp_idft(in, nx, out); }

/* fftc/main.c:46-123: read-outs of a transform (nx/2 values each) */
static inline void ps_percent(fftw_complex *x, unsigned int nx, double *xx) { dsp_ps_percent(x, nx, xx); }
static inline void ps_abs(fftw_complex *x, unsigned int nx, double *xx) { dsp_ps_abs(x, nx, xx); }
static inline void compute_phi(fftw_complex *x, unsigned int nx, double *phi) { dsp_compute_phi(x, nx, phi); }

#ifdef DSP_DROPIN_MAGNITUDE_FFT
/* fir/src/main.c:229-255 */
static inline void fft(fftw_complex *x, unsigned int nx, double *xx) { dsp_fft_mag(x, nx, xx); }
#else
/* fft/src/main.c:146-168, util/fft/fft.c:99-121 */
static inline void fft(double *xx, const double *x, unsigned int n) { dsp_fft_r2c(xx, x, n); }
#endif

/* fir/src/main.c:138-226 */
static inline void make_blackman_window(double *x, unsigned int nx) { dsp_make_blackman_window(x, nx); }
static inline void make_filter_kernel(const double *fresp, unsigned int nx, fftw_complex *kernel)
{
    dsp_make_filter_kernel(fresp, nx, kernel);
}

/* util/filter/filter.c:73-105 */
static inline void filter(double *xx, const double *x, unsigned int n, const double *coeffs)
{
    dsp_filter(xx, x, n, coeffs);
}

#endif /* DSP_DROPIN_H_INCLUDED */
