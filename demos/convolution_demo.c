/* The reference's convolution demo (convolution/main.c:56-98) with its two static
 * routines replaced by libdspmi: same 9-point vectors, same printed columns. */
#include <stdio.h>

#include "dsp_dropin.h"

int main(void)
{
    enum { N = 9 };
    double x[N] = {1, -0.5, -0.25, 0};
    double h[N] = {0, -1, -1.25, 2, 1.3, 1.3, 0.75, 0, -0.75};
    double y0[N], y1[N];
    unsigned int i;

    conv0(x, N, h, N, y0);
    conv1(x, N, h, N, y1);
    for (i = 0; i < N; ++i) printf("%u %lf %lf %lf\n", i, x[i], y0[i], y1[i]);
    return 0;
}
