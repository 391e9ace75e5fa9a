/* The reference's frequency-domain filter tool (util/filter/filter.c:255-388): one column of a
 * text table, a time window of it, per-bin gains set band by band, forward real FFT x gains ->
 * inverse, printed as "index input filtered".
 *   filter_tool -ifile table -icol 1 -fsampl 48000 -tsampl 0:0.01: -filter 3000:24000:0:
 * -tsampl is lo:hi in seconds, -filter is flo:fhi:gain in Hz (several allowed; the LAST one
 * given is applied first, as the reference's list is built by prepending).  Exit status 255
 * on any error. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "columns.h"
#include "dsp_dropin.h"

#define MAX_BANDS 64

/* "a:b:c" -> up to n numbers; every number is the text up to the next ':' */
static void split_numbers(const char *s, double *out, size_t n)
{
    size_t i;
    for (i = 0; i < n; ++i) {
        out[i] = columns_number(s);
        s = strchr(s, ':');
        if (!s) return;
        ++s;
    }
}

int main(int ac, char **av)
{
    const char *ifile = NULL;
    double fsampl = 0, window[2] = {0, 0}, bands[MAX_BANDS][3];
    int have_col = 0, have_fs = 0, have_window = 0, nbands = 0, k, b;
    size_t icol = 0, first, n, nbin, j;
    columns_t table;
    double *x, *gains = NULL, *y = NULL, fband;

    if ((ac - 1) & 1) return -1;
    for (k = 1; k < ac; k += 2) {
        const char *key = av[k], *val = av[k + 1];
        if (!strcmp(key, "-fsampl")) { fsampl = columns_number(val); have_fs = 1; }
        else if (!strcmp(key, "-tsampl")) { split_numbers(val, window, 2); have_window = 1; }
        else if (!strcmp(key, "-filter")) {
            if (nbands == MAX_BANDS) return -1;
            bands[nbands][0] = bands[nbands][1] = bands[nbands][2] = 0;
            split_numbers(val, bands[nbands++], 3);
        }
        else if (!strcmp(key, "-ifile")) ifile = val;
        else if (!strcmp(key, "-ofile")) { /* accepted and unused, as in the reference */ }
        else if (!strcmp(key, "-icol")) { icol = (size_t)columns_number(val); have_col = 1; }
        else return -1;
    }
    if (!ifile || columns_load(&table, ifile)) return -1;
    if (!have_col || !have_fs) goto fail;
    if (!have_window) {
        window[0] = 0.0;
        window[1] = fsampl * (double)table.nrow;
    }
    first = (size_t)floor(window[0] * fsampl);
    if (first >= table.nrow) goto fail;
    n = (size_t)ceil((window[1] - window[0]) * fsampl);
    if (first + n > table.nrow) goto fail;
    fband = fsampl / (double)(unsigned int)n;
    nbin = n / 2 + 1;

    gains = malloc(nbin * sizeof(double));
    if (!gains) goto fail;
    for (j = 0; j < nbin; ++j) gains[j] = 1.0;
    for (b = nbands - 1; b >= 0; --b) {
        size_t lo, hi;
        if (bands[b][0] < 0 || bands[b][1] > (double)nbin * fband || bands[b][0] > bands[b][1]) goto fail;
        lo = (size_t)floor(bands[b][0] / fband);
        hi = (size_t)ceil(bands[b][1] / fband);
        if (hi > nbin) hi = nbin;
        for (; lo < hi; ++lo) gains[lo] = bands[b][2];
    }

    x = columns_get(&table, icol);
    y = malloc(n * sizeof(double));
    if (!y) goto fail;
    filter(y, x + first, (unsigned int)n, gains);
    for (j = 0; j < n; ++j) printf("%zu %lf %lf\n", j, x[first + j], y[j]);
    free(y);
    free(gains);
    columns_free(&table);
    return 0;
fail:
    free(gains);
    columns_free(&table);
    return -1;
}
