/*
 * dspmi_fftw3.h -- the subset of the FFTW3 API the reference calls, over libdspmi.so.
 *
 * Every FFT in the reference is a call into libfftw3; these are all the entry points it
 * uses (fftc/main.c:33-42,140-144; fir/src/main.c:164-182,232-244; fft/src/main.c:155-167;
 * util/fft/fft.c:108-120; util/filter/filter.c:82-104; alsa/main.c:419-426,444-446,471):
 *
 *   fftw_malloc, fftw_free, fftw_plan_dft_1d, fftw_plan_dft_r2c_1d, fftw_plan_dft_c2r_1d,
 *   fftw_execute, fftw_destroy_plan, FFTW_FORWARD, FFTW_BACKWARD, FFTW_ESTIMATE,
 *   fftw_complex (= double[2]).
 *
 * Include this in place of <fftw3.h> and link -ldspmi instead of -lfftw3: the reference's
 * own FFT wrappers then run unchanged.  A plan owns the transform's twiddles and a pair of
 * device buffers; fftw_execute copies the plan's host input to the GPU, runs the double
 * precision transform and copies the result to the plan's host output before it returns.
 * Like FFTW: transforms are unnormalised, r2c yields bins 0..n/2, c2r may be given the same
 * array layout FFTW expects.  Unlike FFTW's c2r, the input array is left intact.
 * The symbols are exported with a dspmi_ prefix and mapped by macros, so a program that also
 * links the real libfftw3 does not see duplicate definitions.
 */
#ifndef DSPMI_FFTW3_H_INCLUDED
#define DSPMI_FFTW3_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double fftw_complex[2];
#define DSP_DROPIN_HAVE_FFTW_COMPLEX 1
typedef struct dspmi_fftw_plan_s *fftw_plan;

#define FFTW_FORWARD (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_MEASURE (0U)
#define FFTW_ESTIMATE (1U << 6)

void *dspmi_fftw_malloc(size_t n);
void dspmi_fftw_free(void *p);
fftw_plan dspmi_fftw_plan_dft_1d(int n, fftw_complex *in, fftw_complex *out, int sign, unsigned flags);
fftw_plan dspmi_fftw_plan_dft_r2c_1d(int n, double *in, fftw_complex *out, unsigned flags);
fftw_plan dspmi_fftw_plan_dft_c2r_1d(int n, fftw_complex *in, double *out, unsigned flags);
void dspmi_fftw_execute(const fftw_plan plan);
void dspmi_fftw_destroy_plan(fftw_plan plan);

#define fftw_malloc dspmi_fftw_malloc
#define fftw_free dspmi_fftw_free
#define fftw_plan_dft_1d dspmi_fftw_plan_dft_1d
#define fftw_plan_dft_r2c_1d dspmi_fftw_plan_dft_r2c_1d
#define fftw_plan_dft_c2r_1d dspmi_fftw_plan_dft_c2r_1d
#define fftw_execute dspmi_fftw_execute
#define fftw_destroy_plan dspmi_fftw_destroy_plan

#ifdef __cplusplus
}
#endif

#endif /* DSPMI_FFTW3_H_INCLUDED */
