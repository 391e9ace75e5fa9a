/* The reference's spectrum tool (util/fft/fft.c:246-363): one column of a text table, a time
 * window of it, real FFT, normalised magnitude spectrum, printed as "frequency share".
 *   fft_tool -ifile table -icol 1 -fsampl 48000 -tsampl_lo 0 -tsampl_hi 0.01
 * Options come in "-key value" pairs; -ifile, -icol and -fsampl are required.  As in the
 * reference, a missing -tsampl_hi defaults to fsampl * rows (seconds), which no table
 * satisfies, so give it.  Exit status 255 on any error, nothing printed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "columns.h"
#include "dsp_dropin.h"

int main(int ac, char **av)
{
    const char *ifile = NULL;
    double fsampl = 0, t_lo = 0, t_hi = 0;
    int have_col = 0, have_fs = 0, have_hi = 0;
    size_t icol = 0, first, n, nbin, i;
    columns_t table;
    double *x, *bins, *ps, sum = 0, fband;
    int k;

    if ((ac - 1) & 1) return -1;
    for (k = 1; k < ac; k += 2) {
        const char *key = av[k], *val = av[k + 1];
        if (!strcmp(key, "-fsampl")) { fsampl = columns_number(val); have_fs = 1; }
        else if (!strcmp(key, "-tsampl_lo")) t_lo = columns_number(val);
        else if (!strcmp(key, "-tsampl_hi")) { t_hi = columns_number(val); have_hi = 1; }
        else if (!strcmp(key, "-ifile")) ifile = val;
        else if (!strcmp(key, "-ofile")) { /* accepted and unused, as in the reference */ }
        else if (!strcmp(key, "-icol")) { icol = (size_t)columns_number(val); have_col = 1; }
        else return -1;
    }
    if (!ifile || columns_load(&table, ifile)) return -1;
    if (!have_col || !have_fs) goto fail;
    if (!have_hi) t_hi = fsampl * (double)table.nrow;

    first = (size_t)floor(t_lo * fsampl);
    if (first >= table.nrow) goto fail;
    n = (size_t)ceil((t_hi - t_lo) * fsampl);
    if (first + n > table.nrow) goto fail;
    fband = fsampl / (double)(unsigned int)n;
    nbin = n / 2 + 1;

    x = columns_get(&table, icol);
    bins = malloc(nbin * 2 * sizeof(double));
    ps = malloc(nbin * sizeof(double));
    if (!bins || !ps) goto fail;
    fft(bins, x + first, (unsigned int)n);
    /* |X| per bin, divided by the total when that is not vanishing (util/fft/fft.c:73-94) */
    for (i = 0; i < nbin; ++i) {
        ps[i] = sqrt(bins[2 * i] * bins[2 * i] + bins[2 * i + 1] * bins[2 * i + 1]);
        sum += ps[i];
    }
    if (sum > 0.00001)
        for (i = 0; i < nbin; ++i) ps[i] /= sum;
    for (i = 0; i < nbin; ++i) printf("%lf %lf\n", fband * (double)(unsigned int)i, ps[i]);
    free(bins);
    free(ps);
    columns_free(&table);
    return 0;
fail:
    columns_free(&table);
    return -1;
}
