/* The call shape of util/filter/filter.c:339-370 without its CSV/argv front end:
 * a 4096-sample two-tone block, a per-bin gain table built from (flo, fhi, gain)
 * bands, `filter`, and the "index x filtered" columns. */
#include <stdio.h>
#include <stdlib.h>

#include "dsp_dropin.h"
#include "tones.h"

int main(void)
{
    const double fsampl = 48000.0;
    const unsigned int n = 4096, nbin = n / 2 + 1;
    const double bands[][3] = {{0, 3000, 1.0}, {3000, 3500, 0.5}};
    double *x = malloc(n * sizeof(double)), *xx = malloc(n * sizeof(double));
    double *coeffs = calloc(nbin, sizeof(double));
    const double fres = fsampl / (double)n;
    unsigned int i, b;
    tones_t gen;

    tones_init(&gen);
    tones_add(&gen, fsampl, 1000.0, 1.0, 0.0);
    tones_add(&gen, fsampl, 12000.0, 0.5, 0.3);
    tones_read(&gen, x, n);
    for (b = 0; b < sizeof(bands) / sizeof(bands[0]); ++b)
        for (i = 0; i < nbin; ++i)
            if ((double)i * fres >= bands[b][0] && (double)i * fres < bands[b][1]) coeffs[i] = bands[b][2];

    filter(xx, x, n, coeffs);
    for (i = 0; i < n; ++i) printf("%u %lf %lf\n", i, x[i], xx[i]);
    free(x); free(xx); free(coeffs);
    return 0;
}
