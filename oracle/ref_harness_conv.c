/*
 * ref_harness_conv.c -- exposes the reference's own conv0/conv1 to the tests.
 *
 * TEST INFRASTRUCTURE ONLY.  The two routines are `static` inside the
 * reference's convolution/main.c, so this translation unit includes that file
 * where it lies (it is never copied into this repository) with its main()
 * renamed, and forwards two exported symbols to the static functions.  It can
 * only be built where the reference tree is mounted; the resulting library
 * lives in oracle/_ref/ and is git-ignored.
 */
#define main ref_convolution_main
#include REF_CONVOLUTION_MAIN_C
#undef main

void ref_conv0(const double *x, unsigned nx, const double *h, unsigned nh,
               double *y)
{
    conv0(x, nx, h, nh, y);
}

void ref_conv1(const double *x, unsigned nx, const double *h, unsigned nh,
               double *y)
{
    conv1(x, nx, h, nh, y);
}
