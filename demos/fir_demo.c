/* The reference's tone -> FIR demo (fft/src/main.c:318-397, time-domain branch):
 * 100 Hz + 12 kHz unit sines at 48 kHz, 4096 samples, low-passed by `fir`, first
 * 1000 rows "t filtered original" printed.  The tap table is the 16-tap set of
 * alsa/main.c:389-407 (the reference's /tmp/bar.h needs gmeteor, which is absent). */
#define DSP_DROPIN_ALLOCATING_CONVOLVE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "dsp_dropin.h"
#include "tones.h"

int main(void)
{
    const double fsampl = 48000.0;
    const double ftones[] = {100.0, 12000.0};
    double fband = 20.0;
    unsigned int nsampl = (unsigned int)(fsampl / fband);
    unsigned int lg = (unsigned int)log2(nsampl);
    unsigned int i, nxx;
    double *x, *xx = NULL;
    tones_t gen;

    if (nsampl != (1u << lg)) nsampl = 1u << (lg + 1); /* 2400 -> 4096, as fft/src/main.c:343-349 */
    x = malloc(nsampl * sizeof(double));
    tones_init(&gen);
    for (i = 0; i < sizeof(ftones) / sizeof(ftones[0]); ++i) tones_add(&gen, fsampl, ftones[i], 1.0, 0.0);
    tones_read_sin(&gen, x, nsampl);

    fir(&xx, &nxx, x, nsampl);
    for (i = 0; i < (nsampl < 1000 ? nsampl : 1000); ++i) printf("%lf %lf %lf\n", (double)i / fsampl, xx[i], x[i]);

    free(x);
    free(xx); /* allocated by the library with malloc, as fft/src/main.c:191,392 */
    return 0;
}
