/*
 * dspmi.h -- C ABI of libdspmi.so, the MI355X (gfx950) implementation of the
 * texane/dsp FIR / convolution / FFT hot path.
 *
 * Two levels:
 *
 *  1. Device-resident batched API (dspmi_*): plain pointers to DEVICE memory,
 *     sizes, an opaque plan handle and a HIP stream passed as void*.  Returns
 *     0 on success or a negative errno-style code (the reference's own
 *     convention for fallible calls, util/common/csv.c:147-216,
 *     util/filter/filter.c get_cmdline_info: -1 on error); dspmi_last_error()
 *     gives the text.  This is the measured path.
 *
 *  2. Host drop-in wrappers (dsp_*): the exact C signatures of the reference's
 *     static compute routines, taking caller-owned HOST double buffers,
 *     blocking until the result is back in host memory.  The reference's
 *     routines return void and check nothing; these do the same on success and
 *     print the failing HIP call to stderr and abort() on any device error --
 *     there is no CPU fallback anywhere in the library.  They compute in
 *     double precision on the device.  include/dsp_dropin.h maps the
 *     reference's unprefixed names (conv0, convolve, fir, dft, ...) onto them.
 *
 * No C++ or torch types appear in any signature.
 *
 * Threads: any number of threads may create, execute and destroy DIFFERENT plans
 * concurrently.  One plan must not be executed from two threads (or on two
 * streams) at once: FFT and overlap-save plans own scratch buffers that an
 * execute may grow and that consecutive kernels of one execute share -- and an
 * f32 FIR plan of 306 taps or more (or created with DSPMI_FIR_VARIANT=ols) owns
 * an overlap-save plan, so the same rule holds for it.  FIR plans without one
 * (f64 plans, f32 plans of up to 305 taps) hold nothing but their taps
 * and may be shared freely.  Every execute is asynchronous on the stream given;
 * destroying a plan while its work is still queued is the caller's error, as with
 * fftw_destroy_plan.
 */
#ifndef DSPMI_H_INCLUDED
#define DSPMI_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (negative errno values) -------------------------------- */
#define DSPMI_OK       0
#define DSPMI_EIO     (-5)   /* a HIP runtime call failed */
#define DSPMI_ENOMEM  (-12)  /* host or device allocation failed */
#define DSPMI_EINVAL  (-22)  /* bad argument */
#define DSPMI_ENOSYS  (-38)  /* size / mode not supported by any kernel */

/* Text of the calling thread's most recent error ("" if none). */
const char *dspmi_last_error(void);

/* ABI version of the loaded library (this header: 1). */
int dspmi_abi_version(void);

/* Number of visible HIP devices, or a negative error code. */
int dspmi_device_count(void);

/* ------------------------------------------------------------------------ */
/* FIR / direct convolution                                                  */
/* ------------------------------------------------------------------------ */

/* A FIR plan holds the taps on one device (uploaded once, zero-padded to the
 * kernel's tap-block size), so tap staging is off the timed path -- the role
 * the compiled-in tap table plays in the reference (fft/src/main.c:213-243,
 * alsa/main.c:389-407). */
typedef struct dspmi_fir_plan *dspmi_fir_t;

/* taps: nh HOST values, h[0] first.  device < 0 means the current device.
 * An f32 plan picks its kernel by tap count, every choice within the same 1e-6 (relative L2 per
 * 4096 outputs) of the double sum:
 *   up to 16 taps   f32 multiply-accumulate chains straight from global memory, four outputs per
 *                   lane (flags == 0, nxx == nx, rows on 16-byte boundaries; the next line's
 *                   kernel otherwise; also 17 .. 24 taps on rows of fewer than 4096 samples and
 *                   25 .. 32 taps on rows of fewer than 2048);
 *   up to 32 taps   f32 multiply-accumulate chains on the matrix unit;
 *   33 .. 305       the same sum through the f16 matrix pipe: samples and taps as two f16 parts
 *                   each, under one power-of-two scale per 4096-output tile (the tile's window of
 *                   4096 + up to 304 samples), accumulated in f32.  What that changes against f32
 *                   products: a sample more than 2^39 below the largest of its tile is not carried --
 *                   every output is within 2^-37 sqrt(nh) max|h| of its tile's largest sample, whatever
 *                   the samples under its own taps.  A tile whose window holds an infinity or a NaN
 *                   is summed tap by tap in f32 instead, so exactly the outputs the reference would
 *                   make non-finite are (fft/src/main.c:197-203);
 *   306 .. 16384    overlap-save (the frequency-domain convolution doc/notes:81-82 names) when
 *                   flags == 0, nxx == nx and the rows hold enough work (up to 899 taps: two
 *                   transforms of four times the tap count; from 900 taps: nx * nh >= 4e5, in
 *                   blocks of 16384 samples, of 32768 from rows of 65536 samples); any other
 *                   call, and longer filters, run the f32 kernel.  306 .. 1009 taps on short rows
 *                   run the f16 kernel of the line above instead, which is faster there (rows
 *                   shorter than 2^19 samples up to 401 taps, 2^17 up to 529, 32768 up to 625, 16384
 *                   up to 817, 12288 up to 899, 8192 beyond; also the calls overlap-save does not take: tails past
 *                   the input, rows that start from zero state); its contract applies to those calls.
 * DSPMI_FIR_VARIANT = mfma | f16 | bf16 | ols in the environment at creation pins the f32 kernel,
 * the f16 kernel, its three-part bf16 sibling (no scale, full f32 exponent range) or overlap-save. */
int dspmi_fir_create_f32(dspmi_fir_t *plan, const float *taps, unsigned nh, int device);
/* The same with options:
 *   DSPMI_FIR_FULL_RANGE    never the f16 kernel: 33 .. 305 taps run its sibling with three bf16 parts
 *                           per operand and no scale -- products exact to 2^-24 over the whole f32
 *                           exponent range, about 20 % slower at 127 taps; 306 .. 1009 taps keep
 *                           overlap-save and the f32 kernel on every row length;
 *   DSPMI_FIR_F32_PRODUCTS  every tap count on the f32 matrix kernel: f32 products, f32 sums.
 * In the two matrix kernels these options select, a non-finite sample makes the outputs of the
 * 16-output blocks around its own nh non-finite as well (up to 15 before, up to 46 after). */
#define DSPMI_FIR_FULL_RANGE   1u
#define DSPMI_FIR_F32_PRODUCTS 2u
int dspmi_fir_create_f32_opts(dspmi_fir_t *plan, const float *taps, unsigned nh, int device, unsigned options);
/* An f64 plan sums in double with every product and every sum rounded on its own, in the
 * reference's order (bit for bit conv0 / conv1 / convolve); filters of up to 12 taps (more on
 * rows shorter than 2048 samples) take the same tile-free kernel as short f32 filters. */
int dspmi_fir_create_f64(dspmi_fir_t *plan, const double *taps, unsigned nh, int device);
int dspmi_fir_destroy(dspmi_fir_t plan);

/* flags for dspmi_fir_exec_* */
#define DSPMI_FIR_DROP_FIRST 1u /* conv1's off-by-one (convolution/main.c:48):
                                   x[0] of every channel never contributes */

/*
 * Batched gather convolution, one independent channel per row:
 *
 *   y[c][i] = sum_{j < nh, 0 <= i-j < nx} h[j] * x[c][i-j],   0 <= i < nxx
 *
 * which is `convolve` of fir/src/main.c:258-280 / alsa/main.c:494-516 (any
 * nxx), the allocating `convolve` of fft/src/main.c:180-205 (nxx = nx+nh) and
 * conv0 of convolution/main.c:15-33 (nxx = nx).  Taps before x[c][0] read the
 * zero history the reference assumes, or, when hist != NULL, the nh-1 samples
 * hist[c][0..nh-2] that preceded x[c][0] (oldest first; hist_stride elements
 * between channels) -- the carried state a streaming caller needs and the
 * reference lacks (alsa/main.c:530 restarts from zero every period).
 *
 * x, y, hist: DEVICE pointers; x_stride / y_stride: elements between
 * consecutive channels (>= nx / >= nxx).  No output row may share an element
 * with an input row (DSPMI_EINVAL): workgroups read the samples before their
 * own tile, so filtering in place would race.
 * stream: a hipStream_t (NULL = the default stream).  Asynchronous.
 */
int dspmi_fir_exec_f32(dspmi_fir_t plan, float *y, size_t y_stride, unsigned nxx,
                       const float *x, size_t x_stride, unsigned nx,
                       unsigned nchan, const float *hist, size_t hist_stride,
                       unsigned flags, void *stream);
int dspmi_fir_exec_f64(dspmi_fir_t plan, double *y, size_t y_stride, unsigned nxx,
                       const double *x, size_t x_stride, unsigned nx,
                       unsigned nchan, const double *hist, size_t hist_stride,
                       unsigned flags, void *stream);

/* Capture blocks of interleaved 16-bit stereo frames through the FIR stage, the whole of
 * do_fir (alsa/main.c:518-540) per block: a frame enters as ((double)left + (double)right) / 2,
 * the block is convolved from zero initial state (no history is carried between blocks,
 * alsa/main.c:530), and result i < nsampl leaves as (int16_t)value in both channels.
 * in/out: device, [nblock] blocks of nsampl frames, strides in frames (4 bytes each); they
 * must not overlap.  plan: a double-precision plan -- products and sums are rounded one by one
 * in the reference's order, so the 16-bit results are the reference's exactly.  A value outside
 * the 32-bit range converts to 0, as the cast does in an x86 build of the reference. */
int dspmi_fir_exec_s16x2(dspmi_fir_t plan, int16_t *out, size_t out_stride, const int16_t *in,
                         size_t in_stride, unsigned nsampl, unsigned nblock, void *stream);

/* Name of the kernel variant the plan dispatches to (for profiles / logs). */
const char *dspmi_fir_variant(dspmi_fir_t plan);

/* ------------------------------------------------------------------------ */
/* FFT                                                                       */
/* ------------------------------------------------------------------------ */

/* Interleaved re,im -- the layout of FFTW's fftw_complex that every
 * reference FFT call site uses (fftc/main.c:140-144). */
typedef struct { float re, im; } dspmi_c32;
typedef struct { double re, im; } dspmi_c64;

#define DSPMI_FORWARD  (-1) /* FFTW_FORWARD:  exponent -2*pi*i*n*k/N */
#define DSPMI_BACKWARD (+1) /* FFTW_BACKWARD: exponent +, unnormalised */

#define DSPMI_F32 0
#define DSPMI_F64 1

/* An FFT plan holds the twiddle table (computed in double on the host, rounded
 * once) and the kernel choice for one transform length on one device; it is
 * the counterpart of fftw_plan_dft_1d / _r2c_1d / _c2r_1d + fftw_destroy_plan
 * (fftc/main.c:33-35, fft/src/main.c:156-166, util/filter/filter.c:86-103). */
typedef struct dspmi_fft_plan *dspmi_fft_t;

/* n: transform length (complex points, or real samples for r2c/c2r): any length up to 2^24
 * points, 2^22 when it is not a power of two (the reference's tools transform whatever window
 * their input holds, util/fft/fft.c:321-352); DSPMI_ENOSYS beyond.  Powers of two up to 32768
 * (f32) / 8192 (f64) run as one kernel, longer ones as two batched passes of those kernels
 * (n = n1 * n2), other lengths through a chirp-z convolution of a power-of-two length. */
int dspmi_fft_create(dspmi_fft_t *plan, unsigned n, int precision, int device);
int dspmi_fft_destroy(dspmi_fft_t plan);

/* batch independent transforms, frame f at in + f*n / out + f*n (elements of
 * dspmi_c32 or dspmi_c64 according to the plan's precision).  Unnormalised in
 * both directions, natural order in and out.  in == out is allowed; buffers that
 * overlap without being the same are refused (DSPMI_EINVAL). */
int dspmi_fft_exec_c2c(dspmi_fft_t plan, const void *in, void *out, size_t batch,
                       int sign, void *stream);

/* dspmi_fft_exec_c2c for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in
 * the reference.  Same arguments and results as dspmi_fft_exec_c2c; the difference is where
 * the multi-kernel transforms keep their work frames (the any-length transform in double
 * precision where it is still several kernels -- M = the power of two >= 2n-1 up to 256:
 * batch * M * 16 bytes; M = 512 .. 16384 is one kernel and needs none --; powers of two beyond the
 * longest single kernel: n points per frame -- f32 up to 2^21 points, f64 up to 2^20 --, frames
 * going through in groups that fit):
 * not in the plan's own buffer, which an execute may have to grow, but in a static device
 * array of DSPMI_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the
 * process ends.  The call performs no hipMalloc / hipFree.  Work frames that exceed the array
 * are refused with DSPMI_EINVAL (a length that needs both kinds of work space at once with
 * DSPMI_ENOSYS).  All calls on one GPU share the array: calls that overlap in time on one GPU
 * (two streams, two threads) race, whatever plans they use. */
#define DSPMI_NOALLOC_SCRATCH_BYTES (16u << 20)
int dspmi_fft_exec_c2c_noalloc(dspmi_fft_t plan, const void *in, void *out, size_t batch,
                               int sign, void *stream);

/* Real-input forward transform: in [batch][n] reals, out [batch][n/2+1]
 * complex (bins 0..n/2), the fftw_plan_dft_r2c_1d contract
 * (fft/src/main.c:155-158).  Even n with a power-of-two half runs the fused real-input
 * kernel; any other n goes through the complex transform.  out may overlap in
 * (a slower two-kernel route that reads everything first). */
int dspmi_fft_exec_r2c(dspmi_fft_t plan, const void *in, void *out, size_t batch,
                       void *stream);

/* Frequency-domain filtering of real frames, `filter` of util/filter/filter.c:73-105 per frame:
 * forward real transform, every bin k <= n/2 scaled by the real gain gains[k], inverse real
 * transform, divided by n.  in/out: [batch][n] reals of the plan's precision (may be the same
 * buffer), gains: n/2+1 device reals. */
int dspmi_fft_exec_filter(dspmi_fft_t plan, const void *in, void *out, const void *gains,
                          size_t batch, void *stream);

/* Backward transform of a Hermitian half spectrum: in [batch][n/2+1] complex,
 * out [batch][n] reals, unnormalised (fftw_plan_dft_c2r_1d,
 * util/filter/filter.c:100-101).  The input is left intact unless the output
 * buffer overlaps it (allowed; f32 frames then take a slower two-kernel route).
 * The imaginary parts of bins 0 and n/2 are ignored, as FFTW ignores them. */
int dspmi_fft_exec_c2r(dspmi_fft_t plan, const void *in, void *out, size_t batch,
                       void *stream);

const char *dspmi_fft_variant(dspmi_fft_t plan);

/* ------------------------------------------------------------------------ */
/* Overlap-save fast convolution (long filters)                              */
/* ------------------------------------------------------------------------ */

/* The FFT-domain form of the same FIR sum, for filters long enough that 2*nh
 * flops per output cost more than two transforms: forward FFT, multiply by the
 * taps' spectrum, backward FFT, divide by n -- the single-block scheme of
 * util/filter/filter.c:73-105 (named as the scalable alternative in
 * doc/notes:81-82) applied to overlapping blocks.  A plan holds H = FFT(taps)/N. */
typedef struct dspmi_ols_plan *dspmi_ols_t;

/* fft_n: block length in samples (power of two >= 2*nh, <= 32768), or 0 to choose.  Blocks of up
 * to 8192 samples ride in pairs, two real blocks per fft_n-point complex transform; blocks of 16384
 * and 32768 samples ride one each on a transform of half their length (even samples as real parts,
 * odd samples as imaginary parts) -- 32768-sample blocks are the default beyond 2048 taps.
 * DSPMI_OLS_REAL=0 in the environment at creation selects the kernels that run pairs of such
 * blocks through full-length transforms instead. */
int dspmi_ols_create_f32(dspmi_ols_t *plan, const float *taps, unsigned nh, unsigned fft_n, int device);
int dspmi_ols_destroy(dspmi_ols_t plan);
unsigned dspmi_ols_fft_length(dspmi_ols_t plan);

/* y[c][i] = sum_j h[j] * x[c][i-j], 0 <= i < nx: the same result as
 * dspmi_fir_exec_f32 with nxx = nx (same history / stride arguments). */
int dspmi_ols_exec_f32(dspmi_ols_t plan, float *y, size_t y_stride, const float *x, size_t x_stride,
                       unsigned nx, unsigned nchan, const float *hist, size_t hist_stride, void *stream);

/* ------------------------------------------------------------------------ */
/* Stages either side of the FIR / FFT in the reference's chain              */
/* ------------------------------------------------------------------------ */

#define DSPMI_PS_NORM_GUARDED 0 /* |X|/sum if sum > 1e-5 else |X|: fft/src/main.c:120-141, util/fft/fft.c:73-94 */
#define DSPMI_PS_NORM         1 /* |X|/sum, no guard: alsa/main.c:473-487 */
#define DSPMI_PS_ABS          2 /* 2|X|/scale: fftc/main.c:80-105, fir/src/main.c:246-251 */
#define DSPMI_PS_PERCENT      3 /* (re - im)/sum if sum >= 1e-4 else re - im: ps_percent, fftc/main.c:46-77 */

/* ps[f][k] for k < nbin from bins[f][k] (dspmi_c32, in_stride complex elements between
 * frames, so the first nbin bins of a longer spectrum can be used in place). */
int dspmi_power_spectrum_f32(float *ps, const void *bins, unsigned nbin, unsigned in_stride,
                             size_t batch, int mode, float scale, void *stream);

/* phi[f][k] = atan(im/re) of bins[f][k] where |re| >= 1e-5, else 0: compute_phi of
 * fftc/main.c:108-123, same layout as dspmi_power_spectrum_f32. */
int dspmi_phase_f32(float *phi, const void *bins, unsigned nbin, unsigned in_stride, size_t batch,
                    void *stream);

/* Real frames straight to their magnitude spectrum (r2c + dspmi_power_spectrum_f32 in one
 * pass over memory): in [batch][n] reals, ps [batch][nbin] with nbin = n/2+1 (fft/src) or n/2
 * (alsa).  f32 plans; the fused kernel covers even n with a power-of-two half. */
int dspmi_fft_exec_r2c_spectrum(dspmi_fft_t plan, const void *in, float *ps, size_t batch,
                                unsigned nbin, int mode, float scale, void *stream);

/* The same from capture blocks of interleaved 16-bit stereo frames, the whole of
 * do_power_spectrum (alsa/main.c:453-492) per block with nbin = n/2 and DSPMI_PS_NORM: in
 * [batch][n] frames (8-byte aligned), each entering as (left + right) / 2. */
int dspmi_fft_exec_r2c_spectrum_s16x2(dspmi_fft_t plan, const int16_t *in, float *ps, size_t batch,
                                      unsigned nbin, int mode, float scale, void *stream);

/* The filter and spectrum stages of the capture loop in one pass (alsa/main.c:542-568: do_fir,
 * then do_power_spectrum on the filtered samples), for callers that want the spectra only:
 * ps[c * (nx / 4096) + f][k], k < nbin, is the normalised magnitude spectrum (mode, scale as
 * above; nbin = 2048 or 2049) of filtered samples 4096 f .. 4096 f + 4095 of channel c, the
 * filtered samples being those of dspmi_fir_exec_f32(fir, ..., nxx = nx, hist) -- they are never
 * written to memory.  fir: an f32 plan of up to 305 taps (the filter runs on the f16 matrix pipe
 * here whatever kernel the plan's own FIR calls take, with that kernel's properties); fft:
 * an f32 plan of 4096 points; nx a multiple of 4096.  DSPMI_ENOSYS for any other pair of plans:
 * the caller then makes the two calls.  A frame whose window of samples (its 4096 and the up to
 * 304 before them that the kernel stages) holds an infinity or a NaN comes out as a spectrum of
 * NaNs -- as the reference's transform of the filtered frame would, except when the non-finite
 * sample lies in the few staged positions beyond the filter's reach. */
int dspmi_fir_spectrum_f32(dspmi_fir_t fir, dspmi_fft_t fft, float *ps, unsigned nbin, int mode,
                           float scale, const float *x, size_t x_stride, unsigned nx, unsigned nchan,
                           const float *hist, size_t hist_stride, void *stream);

/* The whole chain tone generator -> filter -> spectrum in one pass (tonegen_read,
 * tonegen/tonegen.c:25-40, feeding alsa/main.c:542-568): the same spectra as dspmi_tonegen_f32
 * into a buffer x followed by dspmi_fir_spectrum_f32(fir, fft, ps, ..., x, ..., hist), but neither
 * the generated nor the filtered samples reach memory -- a tile's window of samples is computed
 * in double precision straight into the filter's staging registers and rounded to f32 once, as
 * the generator's own output is.  The running angle at every tile's first sample is the exact
 * result of the reference's rounded additions (as in dspmi_tonegen_f32); inside a tile a tone
 * advances by the one fixed increment those additions perform there, applied as a rotation by
 * dw and a second-order correction for the difference: the double-precision sums agree with the
 * separately generated ones to ~1e-10 of the amplitudes, so the f32 samples are the same except
 * where a sum sits on a rounding boundary; in the cosine form the rounding of angle + phase is
 * not reproduced per sample (<= half a unit of the angle, ~1e-10 rad at 1e6).
 * Tiles in which an angle crosses a power of two above 64, and rows whose angles or increments
 * are negative, are generated by dspmi_tonegen_f32's kernel into the work space and read back.
 *   ntone: 1..4 (DSPMI_ENOSYS beyond: generate, then call dspmi_fir_spectrum_f32);
 *   w, dw, amp, phi, use_sin: as for dspmi_tonegen_f32; w is advanced by nx samples;
 *   hist: the nh-1 samples before this call's first (NULL: zeros); hist_out (may be NULL): receives
 *   the last nh-1 generated samples of every row, the next call's hist; the two must not overlap;
 *   work: device scratch of dspmi_tone_fir_spectrum_work_bytes(nx, nchan, ntone) bytes, 256-byte
 *   aligned (tables of a few MB, and room for the rows of the tiles generated separately);
 *   everything else as for dspmi_fir_spectrum_f32. */
size_t dspmi_tone_fir_spectrum_work_bytes(unsigned nx, unsigned nchan, unsigned ntone);
int dspmi_tone_fir_spectrum_f32(dspmi_fir_t fir, dspmi_fft_t fft, float *ps, unsigned nbin, int mode, float scale,
                                unsigned nx, unsigned nchan, unsigned ntone, double *w, const double *dw,
                                const double *amp, const double *phi, int use_sin, const float *hist,
                                size_t hist_stride, float *hist_out, size_t hist_out_stride, void *work,
                                size_t work_bytes, void *stream);

/* Tone generator of tonegen/tonegen.c:25-40 on the device, one generator per channel:
 * x[c][i] = sum_j amp[c][j]*cos(w[c][j] + phi[c][j]), then w[c][j] += dw[c][j], evaluated in
 * double and stored as f32; w (device, [nchan][ntone] doubles) carries the running phases from
 * call to call.  use_sin != 0 selects the unit-amplitude sine form of fft/src/main.c:281-290
 * (amp and phi unused).  ntone <= 32, the reference's limit (tonegen/tonegen.h:14-19).
 * The call is parallel in time as well as over channels: the running angle at the start of
 * every 16384-sample range is computed exactly (the rounded additions inside one binade of the
 * angle are a fixed integer step, so they are applied in one go; additions that cross a binade
 * are performed), then the ranges are generated independently from the same angles, bit for
 * bit, as the sequential recurrence; inside a range cos / sin advance by rotation, which agrees
 * with evaluating every sample to ~1e-14.  DSPMI_TONEGEN_SEQUENTIAL walks each channel with
 * one lane instead (the recurrence as written; for checking). */
#define DSPMI_TONEGEN_SEQUENTIAL 1u
int dspmi_tonegen_f32(float *x, size_t x_stride, unsigned n, unsigned nchan, unsigned ntone,
                      double *w, const double *dw, const double *amp, const double *phi,
                      int use_sin, unsigned flags, void *stream);
/* Same generator, samples stored as complex points with a zero imaginary part -- the input
 * tonegen_read_complex (tonegen/tonegen.c:42-59) hands the c2c transform in fftc/main.c:146-158.
 * x: dspmi_c32 elements, x_stride in complex elements, 8-byte aligned. */
int dspmi_tonegen_c32(void *x, size_t x_stride, unsigned n, unsigned nchan, unsigned ntone,
                      double *w, const double *dw, const double *amp, const double *phi,
                      int use_sin, unsigned flags, void *stream);
/* Same generator, samples stored as interleaved 16-bit stereo frames: (int16_t)value in both
 * channels (convert_double_to_s16x2, tonegen/main.c:14-24).  x_stride in frames.  The sum is
 * formed in double as above; its cosines may differ from the host libm's in the last bit, so a
 * sample whose exact value lies within ~1e-12 of an integer can come out one step apart. */
int dspmi_tonegen_s16x2(int16_t *x, size_t x_stride, unsigned n, unsigned nchan, unsigned ntone,
                        double *w, const double *dw, const double *amp, const double *phi,
                        int use_sin, unsigned flags, void *stream);

/* ------------------------------------------------------------------------ */
/* Host drop-in wrappers (reference signatures; host double buffers)         */
/* ------------------------------------------------------------------------ */

/* The reference's callers run these routines once per period with the same taps and window
 * length (alsa/main.c:494-516 in the capture loop, fft/src/main.c:343-349), so the wrappers keep
 * what a call set up: the last 4 tap sets and 6 transform lengths (up to 2^20 points) per device
 * as plans, and staging buffers that only grow (over 256 MiB: given back when the call ends).
 * One mutex serialises them.  dspmi_dropin_release has no counterpart in the reference: it
 * frees everything the wrappers keep (call it with no wrapper running, e.g. before exit). */
void dspmi_dropin_release(void);

/* convolution/main.c:15-33 */
void dsp_conv0(const double *x, unsigned int nx, const double *h, unsigned int nh,
               double *y);
/* convolution/main.c:36-53 (bug-compatible: drops the j == i term) */
void dsp_conv1(const double *x, unsigned int nx, const double *h, unsigned int nh,
               double *y);
/* fir/src/main.c:258-280 == alsa/main.c:494-516 (caller-owned xx[nxx]) */
void dsp_convolve(double *xx, unsigned int nxx, const double *x, unsigned int nx,
                  const double *h, unsigned int nh);
/* fft/src/main.c:180-205: *xx = malloc((nx+nh) doubles), the caller free()s it */
void dsp_convolve_alloc(double **xx, unsigned int *nxx, const double *x,
                        unsigned int nx, const double *h, unsigned int nh);
/* fft/src/main.c:207-244: `fir` with its compiled-in table; the table is bound
 * here at run time instead (default: the 16 taps of alsa/main.c:389-407). */
void dsp_fir_set_taps(const double *h, unsigned int nh);
void dsp_fir(double **xx, unsigned int *nxx, const double *x, unsigned int nx);

/* fftc/main.c:31-43; in/out are fftw_complex arrays (double[2]) */
void dsp_dft(double (*in)[2], unsigned int nx, double (*out)[2]);
void dsp_idft(double (*in)[2], unsigned int nx, double (*out)[2]);
/* fft/src/main.c:146-168 == util/fft/fft.c:99-121: xx[2*(n/2+1)] <- r2c(x[n]) */
void dsp_fft_r2c(double *xx, const double *x, unsigned int n);
/* fir/src/main.c:229-255: xx[i] = 2|X[i]|/nx for i < nx/2 */
void dsp_fft_mag(double (*x)[2], unsigned int nx, double *xx);

/* The spectrum read-outs of fftc/main.c over a transform x of nx points, nx/2 values each:
 * ps_percent (:46-77), ps_abs (:80-105), compute_phi (:108-123).  ps_percent's sum runs in
 * another order than the reference's loop: results agree to a few ulp, not bit for bit. */
void dsp_ps_percent(double (*x)[2], unsigned int nx, double *xx);
void dsp_ps_abs(double (*x)[2], unsigned int nx, double *xx);
void dsp_compute_phi(double (*x)[2], unsigned int nx, double *phi);
/* make_blackman_window (fir/src/main.c:138-146): 0.42 - 0.5 cos(2 pi i/nx) + 0.08 cos(4 pi i/nx). */
void dsp_make_blackman_window(double *x, unsigned int nx);

/* make_filter_kernel (fir/src/main.c:149-226): filter kernel from a magnitude response.
 * fresp[nx] -> unnormalised backward DFT of the 2*nx-point mirrored spectrum fresp/2 -> the
 * first min(2*nx, 32)/2 samples mirrored into a min(2*nx, 32)-tap kernel, Blackman-windowed,
 * the remaining kernel[.. 2*nx) zeroed.  kernel: 2*nx complex values, caller-owned. */
void dsp_make_filter_kernel(const double *fresp, unsigned int nx, double (*kernel)[2]);

/* util/filter/filter.c:73-105 */
void dsp_filter(double *xx, const double *x, unsigned int n, const double *coeffs);

#ifdef __cplusplus
}
#endif

#endif /* DSPMI_H_INCLUDED */
