/*
 * One node, several GPUs, plain C: the three ways the FIR / FFT path shards, each through
 * include/dspmi.h alone -- one host thread per shard, a shard's plan created on its device with
 * the `device` argument of dspmi_*_create, no collective anywhere on the data path.
 *
 *   channels  batched FIR (convolve of fir/src/main.c:258-280 per channel): the channels are cut
 *             into contiguous blocks, one per shard;
 *   frames    batched c2c transform (dft of fftc/main.c:31-36 per frame): the frames are cut the
 *             same way;
 *   split     ONE long signal through a long filter (the sum of fft/src/main.c:193-204): its
 *             outputs are cut into one segment per shard; a shard is handed its segment's samples
 *             together with the nh-1 samples before them (the halo), which go in as `hist`, so
 *             shards never exchange samples; putting the segments back together is a copy.
 *
 * Shards go to devices round robin (shard s -> device s mod ndev), so the program also runs with
 * more shards than GPUs -- two shards on one GPU is how the test exercises it.
 *
 *   multi_gpu_demo <channels|frames|split> <nshard> <out.bin> [size...]
 *     channels: [nchan nx nh]   frames: [nframe n]   split: [nx nh]
 *   out.bin: the input (f32) followed by the output (f32), for checking against the CPU path.
 *
 * Inputs are synthetic (a fixed linear congruential generator, uniform in [-1, 1)); the filter is
 * the Hamming-windowed low-pass of the benchmarks.
 */
#define __HIP_PLATFORM_AMD__
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dspmi.h"

#define CHECK_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return (void *)1;                                                                    \
        }                                                                                        \
    } while (0)
#define CHECK_DSP(call)                                                                          \
    do {                                                                                         \
        int r_ = (call);                                                                         \
        if (r_ != DSPMI_OK) {                                                                    \
            fprintf(stderr, "%s:%d: %s failed (%d): %s\n", __FILE__, __LINE__, #call, r_, dspmi_last_error()); \
            return (void *)1;                                                                    \
        }                                                                                        \
    } while (0)

enum mode { CHANNELS, FRAMES, SPLIT };

struct shard {
    enum mode mode;
    int device;
    const float *taps;
    unsigned nh;
    /* channels / frames: rows [row_lo, row_hi) of `len` samples (complex points: 2 floats each) */
    size_t row_lo, row_hi, len;
    /* split: outputs [out_lo, out_hi) of the signal, computed from samples [in_lo, out_hi) */
    size_t out_lo, out_hi, in_lo;
    const float *in;   /* whole host input */
    float *out;        /* whole host output */
};

static void fill_uniform(float *x, size_t n, uint32_t seed)
{
    uint32_t s = seed;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        x[i] = (float)(s >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
}

static void lowpass_taps(float *h, unsigned nh)
{
    double sum = 0.0, *t = malloc(nh * sizeof *t);
    for (unsigned i = 0; i < nh; i++) {
        const double k = (double)i - (nh - 1) / 2.0, a = 2.0 * 6000.0 / 48000.0 * k;
        const double sinc = a == 0.0 ? 1.0 : sin(M_PI * a) / (M_PI * a);
        t[i] = sinc * (0.54 - 0.46 * cos(2.0 * M_PI * i / (nh - 1)));
        sum += t[i];
    }
    for (unsigned i = 0; i < nh; i++) h[i] = (float)(t[i] / sum);
    free(t);
}

/* One shard, start to finish, on its own device and stream. */
static void *run_shard(void *arg)
{
    struct shard *s = arg;
    hipStream_t stream;
    CHECK_HIP(hipSetDevice(s->device));
    CHECK_HIP(hipStreamCreate(&stream));
    if (s->mode == CHANNELS) {
        const size_t rows = s->row_hi - s->row_lo, bytes = rows * s->len * sizeof(float);
        float *dx, *dy;
        dspmi_fir_t fir;
        if (rows == 0) return NULL;
        CHECK_HIP(hipMalloc((void **)&dx, bytes));
        CHECK_HIP(hipMalloc((void **)&dy, bytes));
        CHECK_DSP(dspmi_fir_create_f32(&fir, s->taps, s->nh, s->device));
        CHECK_HIP(hipMemcpyAsync(dx, s->in + s->row_lo * s->len, bytes, hipMemcpyHostToDevice, stream));
        CHECK_DSP(dspmi_fir_exec_f32(fir, dy, s->len, (unsigned)s->len, dx, s->len, (unsigned)s->len, (unsigned)rows, NULL, 0, 0, stream));
        CHECK_HIP(hipMemcpyAsync(s->out + s->row_lo * s->len, dy, bytes, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_DSP(dspmi_fir_destroy(fir));
        CHECK_HIP(hipFree(dx));
        CHECK_HIP(hipFree(dy));
    } else if (s->mode == FRAMES) {
        const size_t rows = s->row_hi - s->row_lo, bytes = rows * s->len * 2 * sizeof(float);
        float *din, *dout;
        dspmi_fft_t fft;
        if (rows == 0) return NULL;
        CHECK_HIP(hipMalloc((void **)&din, bytes));
        CHECK_HIP(hipMalloc((void **)&dout, bytes));
        CHECK_DSP(dspmi_fft_create(&fft, (unsigned)s->len, DSPMI_F32, s->device));
        CHECK_HIP(hipMemcpyAsync(din, s->in + s->row_lo * s->len * 2, bytes, hipMemcpyHostToDevice, stream));
        CHECK_DSP(dspmi_fft_exec_c2c(fft, din, dout, rows, DSPMI_FORWARD, stream));
        CHECK_HIP(hipMemcpyAsync(s->out + s->row_lo * s->len * 2, dout, bytes, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_DSP(dspmi_fft_destroy(fft));
        CHECK_HIP(hipFree(din));
        CHECK_HIP(hipFree(dout));
    } else {
        /* the slice holds the halo first: hist = its start, x = what follows */
        const size_t halo = s->out_lo - s->in_lo, nout = s->out_hi - s->out_lo;
        float *dslice, *dy;
        dspmi_ols_t ols;
        if (nout == 0) return NULL;
        CHECK_HIP(hipMalloc((void **)&dslice, (halo + nout) * sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&dy, nout * sizeof(float)));
        CHECK_DSP(dspmi_ols_create_f32(&ols, s->taps, s->nh, 0, s->device));
        CHECK_HIP(hipMemcpyAsync(dslice, s->in + s->in_lo, (halo + nout) * sizeof(float), hipMemcpyHostToDevice, stream));
        CHECK_DSP(dspmi_ols_exec_f32(ols, dy, nout, dslice + halo, nout, (unsigned)nout, 1, halo ? dslice : NULL, halo, stream));
        CHECK_HIP(hipMemcpyAsync(s->out + s->out_lo, dy, nout * sizeof(float), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_DSP(dspmi_ols_destroy(ols));
        CHECK_HIP(hipFree(dslice));
        CHECK_HIP(hipFree(dy));
    }
    CHECK_HIP(hipStreamDestroy(stream));
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <channels|frames|split> <nshard> <out.bin> [sizes...]\n", argv[0]);
        return 255;
    }
    const enum mode mode = !strcmp(argv[1], "channels") ? CHANNELS : !strcmp(argv[1], "frames") ? FRAMES : SPLIT;
    const int nshard = atoi(argv[2]);
    const int ndev = dspmi_device_count();
    if (nshard < 1 || nshard > 64 || ndev < 1) {
        fprintf(stderr, "need 1..64 shards and at least one device (found %d)\n", ndev);
        return 255;
    }
    size_t a1 = argc > 4 ? strtoull(argv[4], NULL, 0) : 0, a2 = argc > 5 ? strtoull(argv[5], NULL, 0) : 0;
    size_t a3 = argc > 6 ? strtoull(argv[6], NULL, 0) : 0;
    size_t nrow, len, n_in, n_out;
    unsigned nh = 0;
    if (mode == CHANNELS) {
        nrow = a1 ? a1 : 64, len = a2 ? a2 : 65536, nh = a3 ? (unsigned)a3 : 127;
        n_in = n_out = nrow * len;
    } else if (mode == FRAMES) {
        nrow = a1 ? a1 : 4096, len = a2 ? a2 : 1024;
        n_in = n_out = nrow * len * 2;
    } else {
        nrow = 1, len = a1 ? a1 : (size_t)1 << 22, nh = a2 ? (unsigned)a2 : 8191;
        n_in = n_out = len;
        if (len < (size_t)nshard) {
            fprintf(stderr, "fewer samples than shards\n");
            return 255;
        }
    }
    float *in = malloc(n_in * sizeof *in), *out = calloc(n_out, sizeof *out), *taps = nh ? malloc(nh * sizeof *taps) : NULL;
    if (!in || !out || (nh && !taps)) return 255;
    fill_uniform(in, n_in, 0xD5B0u + (unsigned)mode);
    if (nh) lowpass_taps(taps, nh);

    struct shard sh[64];
    pthread_t th[64];
    for (int s = 0; s < nshard; s++) {
        memset(&sh[s], 0, sizeof sh[s]);
        sh[s].mode = mode;
        sh[s].device = s % ndev;
        sh[s].taps = taps;
        sh[s].nh = nh;
        sh[s].in = in;
        sh[s].out = out;
        sh[s].len = len;
        if (mode == SPLIT) {
            /* equal output segments (the last takes the remainder); the input slice reaches nh-1 back */
            const size_t seg = len / nshard;
            sh[s].out_lo = seg * s;
            sh[s].out_hi = s == nshard - 1 ? len : seg * (s + 1);
            sh[s].in_lo = sh[s].out_lo >= nh - 1 ? sh[s].out_lo - (nh - 1) : 0;
            /* a first segment shorter than the filter: its halo is what exists, zero state before it */
            if (sh[s].out_lo - sh[s].in_lo != nh - 1 && sh[s].out_lo != 0) {
                fprintf(stderr, "segments shorter than the filter's reach are not cut here\n");
                return 255;
            }
        } else {
            sh[s].row_lo = nrow * s / nshard;
            sh[s].row_hi = nrow * (s + 1) / nshard;
        }
    }
    int failed = 0;
    for (int s = 0; s < nshard; s++)
        if (pthread_create(&th[s], NULL, run_shard, &sh[s])) return 255;
    for (int s = 0; s < nshard; s++) {
        void *rv = NULL;
        pthread_join(th[s], &rv);
        failed |= rv != NULL;
    }
    if (failed) return 1;

    FILE *f = fopen(argv[3], "wb");
    if (!f || fwrite(in, sizeof *in, n_in, f) != n_in || fwrite(out, sizeof *out, n_out, f) != n_out || fclose(f)) {
        fprintf(stderr, "cannot write %s\n", argv[3]);
        return 255;
    }
    printf("%s: %d shards on %d device(s), %zu rows of %zu, %u taps\n", argv[1], nshard, ndev, nrow, len, nh);
    for (int s = 0; s < nshard; s++) {
        if (mode == SPLIT)
            printf("  shard %d -> device %d: outputs [%zu, %zu) from samples [%zu, %zu)\n", s, sh[s].device, sh[s].out_lo,
                   sh[s].out_hi, sh[s].in_lo, sh[s].out_hi);
        else
            printf("  shard %d -> device %d: rows [%zu, %zu)\n", s, sh[s].device, sh[s].row_lo, sh[s].row_hi);
    }
    free(in);
    free(out);
    free(taps);
    return 0;
}
