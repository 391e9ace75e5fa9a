// Internal interface between the FFT kernels (fft_stockham.hip) and the plan /
// real-transform / Bluestein layer (fft_plan.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace dspmi {

// Launches batch transforms of one power-of-two length: in/out/tw are device pointers to
// interleaved complex data of the kernel's precision, tw = exp(-2 pi i k / n), k < n.
typedef int (*fft_launch_fn)(const void *in, void *out, const void *tw, size_t batch, int backward, int num_cu,
                             hipStream_t s);

// Real input through this N-point complex kernel: in = batch frames of 2N reals; epilogue 1
// writes bins 0..N (N+1 complex per frame), epilogue 2 writes the magnitude spectrum over nbin
// (N or N+1) bins normalised per `mode` (DSPMI_PS_*).  tw2n = exp(-2 pi i k / 2N), k < N.
// epilogue | kR2cSourceS16x2: `in` holds 2N interleaved 16-bit stereo frames per transform
// instead of 2N floats; a frame enters as (left + right) / 2.
constexpr int kR2cSourceS16x2 = 0x100;
typedef int (*fft_r2c_launch_fn)(const void *in, void *out, const void *tw, const void *tw2n, size_t batch, int epilogue,
                                 unsigned nbin, int mode, float scale, int num_cu, hipStream_t s);

// Real output through this N-point complex kernel: in = batch frames of N+1 bins, out = batch
// frames of 2N reals (unnormalised); in and out must not overlap.
typedef int (*fft_c2r_launch_fn)(const void *in, void *out, const void *tw, const void *tw2n, size_t batch, int num_cu,
                                 hipStream_t s);

struct FftKernel {
    unsigned n;
    fft_launch_fn fn;         // default cache policy
    fft_launch_fn fn_stream;  // nontemporal loads and stores: batches far larger than the caches
    const char *name;
    fft_r2c_launch_fn fn_r2c;  // fused real-input form, or nullptr
    fft_c2r_launch_fn fn_c2r;  // fused real-output form, or nullptr
};

// Kernel for a power-of-two length, or nullptr.
const FftKernel *fft_find(unsigned n, int precision);

// One pass of a long transform done in two (n = n1 n2) or three (n = t1 t2 t3), N = 128 .. 1024 points
// per column or row, nframe frames.  tw = exp(-2 pi i k / N); twa, twb, shift: the two-table form of W_n.
//   mode 1: `in` seen as [N][ncol], the N-point transforms of its columns, times W_n^(column . k), left
//           in `out` as tiles [k / 16][column / 16][k % 16][column % 16];
//   mode 2: `in` in those tiles (ncol values of k, N columns), the N-point transforms over the
//           columns, stored as columns of `out` seen as [N][ncol];
//   mode 3: as mode 1 with the factors W_n^((column - column % p0) . k), result in place ([k][column]);
//   mode 4: as mode 1 with the factors W_n^(column . (p0 k + frame % p0)), result in place;
//   mode 5: frames seen as [p1][p0][N]: the N-point transforms of rows (kc, kb), stored as columns
//           p1 kb + kc of `out` seen as [N][ncol], ncol = p0 p1.
typedef int (*fft_bundle_fn)(const void *in, void *out, const void *tw, size_t nframe, unsigned ncol, const void *twa,
                             const void *twb, unsigned shift, int mode, int backward, unsigned p0, unsigned p1, hipStream_t s);
fft_bundle_fn fft_bundle_find(unsigned n, int precision);

// Fused overlap-save (f32): blocks of N samples overlapping by nh-1, two real blocks per
// complex transform, H = FFT_N(taps)/N (device, N complex).  All pointers are device pointers.
struct OlsLaunch {
    const float *x, *hist;
    float *y;
    const void *H;
    size_t x_stride, y_stride, hist_stride;
    unsigned nx, nh, nchan;
};
typedef int (*ols_launch_fn)(const OlsLaunch &o, const void *tw, int num_cu, hipStream_t s);
ols_launch_fn ols_find(unsigned n);
// The same with one real block of `block` samples per block/2-point complex transform (32768 or
// 16384: 512-lane workgroups, two or three per CU), the transform as R = block/1024 rows of 512
// points.  H = block/2 entries of two complex values, {Hev / (block/2), (i/2) Hod / (block/2)}
// (the transforms of the even and odd taps), stored row by row: entry k1 + R k2 at [k1][k2];
// tw = exp(-2 pi i k / 512), k < 512, then exp(-2 pi i k1 n2 / (block/2)) as [R][512], then
// exp(-2 pi i (k1 + R l) / (block/2)) as [R][64].
ols_launch_fn ols_find_real(unsigned block);

// Any-length transform in one kernel (Bluestein over an M-point convolution, f32): in/out =
// batch frames of n complex points, chirp[m] = exp(-i pi m^2 / n) (n entries), filt =
// FFT_M(conj(chirp) laid out circularly) / M, tw = exp(-2 pi i k / M).  in == out is allowed.
typedef int (*bluestein_launch_fn)(const void *in, void *out, const void *chirp, const void *filt, const void *tw, unsigned n,
                                   size_t batch, int backward, int num_cu, hipStream_t s);
bluestein_launch_fn bluestein_find(unsigned m, int precision);

// FIR (the f16 matrix-pipe kernel of a FIR plan) -> 4096-point real transform -> normalised
// magnitude spectrum in one kernel.  x: nchan rows of nx samples, nx a multiple of 4096; ps:
// nchan * nx / 4096 rows of nbin (2048 or 2049) floats.  tap_images / tap_unscale / slices come
// from the FIR plan, tw_half = exp(-2 pi i k / 2048), tw_full = exp(-2 pi i k / 4096) from a
// 4096-point FFT plan.  All pointers are device pointers.
struct FirSpectrumLaunch {
    const float *x, *hist;
    size_t x_stride, hist_stride;
    unsigned nx, nh, nchan;
    const void *tap_images;
    const float *taps;   // the plan's taps as they are (nh floats)
    float tap_unscale;
    int slices;
    const void *tw_half, *tw_full;
    float *ps;
    unsigned nbin;
    int mode;
    float scale;
    // tone_slots > 0 (2 or 4): the rows are generated inside the kernel from these tables
    // (tone_chain_prepare, spectrum.hip); x then holds the generator's own output for the tiles
    // the tables mark as hard, and hist_out receives every row's last nh-1 samples
    int tone_slots;
    const double *lane_tab, *tile_tab;
    const unsigned *hard;
    float *hist_out;
    size_t hist_out_stride;
    // the generators' running angles w[nseq], advanced by nx increments dw[nseq] by the same launch
    double *w;
    const double *dw;
    unsigned nseq;
};
int fir_spectrum_launch(const FirSpectrumLaunch &a, hipStream_t s);


// Tables for the tone source of the fused FIR + spectrum kernel (layouts in fir_half.h), one
// call's worth: running angles w[nchan][ntone] (read, not advanced), increments dw, amplitudes
// and phases (cosine form) or unit sines (use_sin).  A window covers samples
// [4096 tile - lead, 4096 tile + 4096) of a row of nx samples.  Tiles marked hard are generated
// into spill[chan][nx] by the stand-alone generator.  slots: 2 or 4 (>= ntone).
struct ToneChainTables {
    double *lane_tab, *tile_tab;
    unsigned *hard;
    float *spill;
};
size_t tone_chain_work_bytes(unsigned nx, unsigned nchan, int slots);
ToneChainTables tone_chain_carve(void *work, unsigned nx, unsigned nchan, int slots);
int tone_chain_prepare(const ToneChainTables &tab, unsigned nx, unsigned nchan, unsigned ntone, int slots, unsigned lead,
                       const double *w, const double *dw, const double *amp, const double *phi, int use_sin, hipStream_t s);

}  // namespace dspmi
