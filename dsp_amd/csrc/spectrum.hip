// Spectrum post-ops and the tone generator: the cheap stages either side of the FIR / FFT
// kernels in the reference's real-time chain (alsa/main.c:542-568: spectrum, FIR, spectrum).
//
//  * power spectrum: |X[k]| per bin and one sum per frame, then the reference's
//    normalisation -- fft_to_power_spectrum (fft/src/main.c:120-141: divide by the sum when
//    it exceeds 1e-5), do_power_spectrum (alsa/main.c:473-487: always divide),
//    ps_abs (fftc/main.c:80-105: 2|X|/n, no sum).  One wave per frame; the bins stay in
//    registers between the magnitude pass and the divide.
//  * tone generator: sum of a*cos(w+phi) with the running-phase recurrence w += dw of
//    tonegen/tonegen.c:25-40, kept in double so a stream continues exactly where the last
//    call stopped.  A sequential one-lane-per-channel kernel is the recurrence as written;
//    the default kernel is parallel in time: it jumps to the exact angle of any sample
//    (advance) and steps cos / sin by rotation where the angles form an exact arithmetic
//    sequence.
#include "common.h"
#include "fft_internal.h"
#include "tone_math.h"

#include <cstdint>
#include <type_traits>

#include <cmath>

namespace {

struct alignas(8) C32 {
    float re, im;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// mode 0: ps = |X| / sum if sum > 1e-5 else |X|      (fft/src/main.c:120-141)
// mode 1: ps = |X| / sum                             (alsa/main.c:473-487)
// mode 2: ps = 2|X| / scale                          (fftc/main.c:80-105)
// mode 3: ps = (re - im) / sum if sum >= 1e-4        (ps_percent, fftc/main.c:46-77)
template <typename T>
__device__ __forceinline__ T bin_value(T re, T im, int mode)
{
    return mode == 3 ? re - im : sqrt(re * re + im * im);
}

// The divisor and factor a frame's values are scaled by, from the frame's sum.
template <typename T>
__device__ __forceinline__ void frame_scale(int mode, T sum, T scale, T &mul, T &div)
{
    mul = T(1);
    div = T(1);
    if (mode == 2) {
        mul = T(2);
        div = scale;
    } else if (mode == 3) {
        if (sum >= T(0.0001)) div = sum;
    } else if (mode == 1 || sum > T(0.00001)) {
        div = sum;
    }
}

template <int PER_LANE>
__global__ __launch_bounds__(256) void power_spectrum_kernel(float *__restrict__ ps, const C32 *__restrict__ bins,
                                                             unsigned nbin, unsigned in_stride, size_t batch, int mode,
                                                             float scale)
{
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t f = wave; f < batch; f += nwaves) {
        const C32 *src = bins + f * in_stride;
        float mag[PER_LANE];
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < PER_LANE; q++) {
            const unsigned k = lane + 64 * q;
            float m = 0.f;
            if (k < nbin) {
                const C32 v = src[k];
                m = bin_value(v.re, v.im, mode);
            }
            mag[q] = m;
            part += m;
        }
        float div, mul;
        frame_scale(mode, mode == 2 ? 0.f : wave_sum(part), scale, mul, div);
#pragma unroll
        for (int q = 0; q < PER_LANE; q++) {
            const unsigned k = lane + 64 * q;
            if (k < nbin) ps[f * nbin + k] = (mul * mag[q]) / div;
        }
    }
}

// Frames too long for the register path, and the double-precision form behind the host
// wrappers: two passes over the bins.
template <typename T>
__device__ __forceinline__ T wave_sum_t(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void power_spectrum_long_kernel(T *__restrict__ ps, const T *__restrict__ bins, unsigned nbin,
                                                                  unsigned in_stride, size_t batch, int mode, T scale)
{
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t f = wave; f < batch; f += nwaves) {
        const T *src = bins + 2 * f * in_stride;  // interleaved re, im
        T part = T(0);
        for (unsigned k = lane; k < nbin; k += 64) part += bin_value(src[2 * k], src[2 * k + 1], mode);
        T div, mul;
        frame_scale(mode, mode == 2 ? T(0) : wave_sum_t(part), scale, mul, div);
        for (unsigned k = lane; k < nbin; k += 64) ps[f * nbin + k] = (mul * bin_value(src[2 * k], src[2 * k + 1], mode)) / div;
    }
}

// phi = atan(im / re) where |re| >= 1e-5, else 0 (compute_phi, fftc/main.c:108-123)
template <typename T>
__global__ __launch_bounds__(256) void phase_kernel(T *__restrict__ phi, const T *__restrict__ bins, unsigned nbin,
                                                    unsigned in_stride, size_t batch)
{
    const size_t total = batch * (size_t)nbin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / nbin;
        const size_t k = i - f * nbin;
        const T re = bins[2 * (f * in_stride + k)], im = bins[2 * (f * in_stride + k) + 1];
        phi[i] = fabs(re) >= T(0.00001) ? atan(im / re) : T(0);
    }
}

// state: [nchan][ntone] running angles (double, updated); dw, amp, phi: [nchan][ntone].
// 64 channels per workgroup of 64 lanes; each lane advances its channel 64 samples, the
// tile is transposed through LDS and written as 64 rows of 256 contiguous bytes.
// O = float: samples as floats; O = int: one interleaved 16-bit stereo frame per sample, both
// channels (int16_t)value (tonegen/main.c:14-24); O = C32: complex samples with zero imaginary
// part (tonegen_read_complex, tonegen/tonegen.c:42-59).
template <typename O>
__device__ __forceinline__ O tone_out(double v)
{
    if constexpr (std::is_floating_point<O>::value) {
        return (O)v;
    } else if constexpr (std::is_same<O, C32>::value) {
        return C32{(float)v, 0.f};  // tonegen_read_complex: the imaginary part is always 0
    } else {
        // the cast as an x86 build performs it: truncate to 32 bits (out of range -> 0x80000000), low half
        const unsigned h = (v > -2147483649.0 && v < 2147483648.0) ? ((unsigned)(int)v & 0xffffu) : 0u;
        return (O)(h | (h << 16));
    }
}

template <typename O, int MAXT>
__global__ __launch_bounds__(64) void tonegen_kernel(O *__restrict__ x, size_t x_stride, unsigned n, unsigned nchan,
                                                     unsigned ntone, double *__restrict__ w, const double *__restrict__ dw,
                                                     const double *__restrict__ amp, const double *__restrict__ phi,
                                                     int use_sin)
{
    __shared__ O tile[64][65];
    const unsigned lane = threadIdx.x;
    const unsigned c = blockIdx.x * 64 + lane;
    const bool live = c < nchan;
    double wl[MAXT], dwl[MAXT], al[MAXT], pl[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; j++) {
        const bool on = live && j < (int)ntone;
        wl[j] = on ? w[(size_t)c * ntone + j] : 0.0;
        dwl[j] = on ? dw[(size_t)c * ntone + j] : 0.0;
        al[j] = on ? amp[(size_t)c * ntone + j] : 0.0;
        pl[j] = on ? phi[(size_t)c * ntone + j] : 0.0;
    }
    for (unsigned base = 0; base < n; base += 64) {
        for (unsigned i = 0; i < 64; i++) {
            // buf[i] = 0; buf[i] += a*cos(w + phi); w += dw  -- tones in order, in double
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < MAXT; j++) {
                if (j < (int)ntone) {
                    v += use_sin ? tone_sin(wl[j]) : al[j] * tone_cos(wl[j] + pl[j]);
                    if (base + i < n) wl[j] += dwl[j];
                }
            }
            tile[lane][i] = tone_out<O>(v);
        }
        __syncthreads();
        for (unsigned r = 0; r < 64; r++) {
            const unsigned cc = blockIdx.x * 64 + r;
            if (cc < nchan && base + lane < n) x[(size_t)cc * x_stride + base + lane] = tile[r][lane];
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < MAXT; j++)
        if (live && j < (int)ntone) w[(size_t)c * ntone + j] = wl[j];
}

// running angles after a call: one lane per (channel, tone)
__global__ __launch_bounds__(64) void tonegen_finish_kernel(double *__restrict__ w, const double *__restrict__ dw, unsigned n,
                                                            unsigned nseq)
{
    const unsigned s = blockIdx.x * 64 + threadIdx.x;
    if (s < nseq) w[s] = advance(w[s], dw[s], n);
}

// One wave per range of 16384 samples of one channel; lane l produces samples l, l + 64, ...,
// so every store is 256 contiguous bytes.  The range is walked in segments inside which every
// tone's angles stay on one binade's grid, a_k = a + k*step exactly (usually the whole range is
// one segment; an angle crossing a power of two ends a segment, and the crossing addition is a
// segment of one sample).  In a segment a lane evaluates cos / sin once, at its first sample,
// and then turns them by the fixed angle 64*step per sample: four multiply-adds per sample and
// tone, ~3e-14 of accumulated rounding over at most 256 turns.  The reference rounds
// angle + phase before taking the cosine; that rounding is recovered exactly (two-sum) and
// applied as a first-order correction.
constexpr unsigned kToneSteps = 256;                // samples per lane
constexpr unsigned kToneRange = 64 * kToneSteps;    // samples per wave
constexpr unsigned kToneStepsGrouped = 32;          // samples per lane when the tones go through in groups

// MAXT: tones the instantiation holds at a time (the loops over tones are unrolled to it).
// STEPS: samples per lane, 64 * STEPS per wave.  GROUPS: more than MAXT tones (the reference's
// generator holds up to 32, tonegen/tonegen.h:14-19) go through in groups of MAXT; a sample's
// running sum waits in LDS between groups, in double, so every sample still adds its tones in
// the reference's order (tonegen/tonegen.c:31-37) and rounds once.
template <typename O, int MAXT, int STEPS, bool GROUPS>
__device__ __forceinline__ void tone_range(O *__restrict__ x, size_t x_stride, unsigned ntone, const double *__restrict__ w,
                                           const double *__restrict__ dw, const double *__restrict__ amp,
                                           const double *__restrict__ phi, int use_sin, unsigned chan, size_t first, unsigned len,
                                           double *partial)
{
    const unsigned lane = threadIdx.x & 63;
    O *xrow = x + (size_t)chan * x_stride + first;

    for (unsigned j0 = 0; j0 < (GROUPS ? ntone : 1u); j0 += MAXT) {
        const int nt = GROUPS ? (int)min((unsigned)MAXT, ntone - j0) : (int)ntone;  // tones of this group
        const bool first_group = j0 == 0, last_group = !GROUPS || j0 + MAXT >= ntone;
        double A[MAXT], dl[MAXT], al[MAXT], pl[MAXT];
#pragma unroll
        for (int j = 0; j < MAXT; j++) {
            const bool on = j < nt;
            const size_t s = (size_t)chan * ntone + j0 + j;
            dl[j] = on ? dw[s] : 0.0;
            A[j] = on ? advance(w[s], dl[j], first) : 0.0;  // angle at the range's first sample: the same for every lane
            al[j] = on ? amp[s] : 0.0;
            pl[j] = (on && !use_sin) ? phi[s] : 0.0;
        }

        unsigned pos = 0;
        while (pos < len) {
            // the segment: as far as every tone's run reaches (one sample where a tone needs a real addition)
            unsigned seg = len - pos;
            double step[MAXT];
            bool real_add[MAXT];
#pragma unroll
            for (int j = 0; j < MAXT; j++) {
                step[j] = 0.0;
                real_add[j] = false;
                if (j < nt) {
                    const unsigned long long run = (A[j] >= 0.0 && dl[j] > 0.0) ? fixed_run(A[j], dl[j], step[j]) : 0;
                    real_add[j] = run == 0;
                    if (run < seg) seg = run == 0 ? 1 : (unsigned)run;
                }
            }
            // this lane's samples in the segment: k = k0, k0 + 64, ... < pos + seg
            const unsigned off = (lane - pos) & 63;  // k0 - pos
            double a0[MAXT], st64[MAXT], cs[MAXT], sn[MAXT], cd[MAXT], sd[MAXT];
            if (off < seg) {
#pragma unroll
                for (int j = 0; j < MAXT; j++) {
                    a0[j] = fma((double)off, step[j], A[j]);  // exact
                    st64[j] = 64.0 * step[j];
                    // cos / sin of the exact angle a + phase: of the rounded sum, corrected by its residual
                    const double sum = a0[j] + pl[j];
                    const double bb = sum - a0[j];
                    const double err = (a0[j] - (sum - bb)) + (pl[j] - bb);  // a + phase = sum + err
                    const double c0 = tone_cos(sum), s0 = tone_sin(sum);
                    cs[j] = fma(-err, s0, c0);
                    sn[j] = fma(err, c0, s0);
                    if (seg > 64) {
                        cd[j] = tone_cos(st64[j]);
                        sd[j] = tone_sin(st64[j]);
                    } else {
                        cd[j] = 1.0;  // a single sample per lane: never turned
                        sd[j] = 0.0;
                    }
                }
                unsigned i = 0;
                if (!GROUPS && use_sin && nt == MAXT) {
                    // The common case (every slot a unit sine, fft/src/main.c:281-290) without the
                    // per-tone conditions of the general loop: the same operations in the same
                    // order, 14 vector instructions per sample instead of 27 and no branches (the
                    // kernel is bound by its double-precision instructions).
#pragma unroll 4
                    for (unsigned k = pos + off; k < pos + seg; k += 64) {
                        double v = 0.0;
#pragma unroll
                        for (int j = 0; j < MAXT; j++) {
                            v += sn[j];
                            const double cn = fma(cs[j], cd[j], -(sn[j] * sd[j]));
                            sn[j] = fma(sn[j], cd[j], cs[j] * sd[j]);
                            cs[j] = cn;
                        }
                        xrow[k] = tone_out<O>(v);
                    }
                } else
                for (unsigned k = pos + off; k < pos + seg; k += 64, i++) {
                    double v = 0.0;
                    if constexpr (GROUPS) {
                        if (!first_group) v = partial[k];  // the lane's own slot: k = lane (mod 64) in every group
                    }
#pragma unroll
                    for (int j = 0; j < MAXT; j++) {
                        if (j < nt) {  // an idle slot must not add anything, not even +0.0 to a -0.0 sum
                            if (use_sin) {
                                v += sn[j];
                            } else {
                                double c = cs[j];
                                if (pl[j] != 0.0) {
                                    // the reference's angle is fl(a_k + phase) = (a_k + phase) - err
                                    const double ak = fma((double)i, st64[j], a0[j]);
                                    const double sum = ak + pl[j];
                                    const double bb = sum - ak;
                                    const double err = (ak - (sum - bb)) + (pl[j] - bb);
                                    c = fma(err, sn[j], c);
                                }
                                v += al[j] * c;
                            }
                        }
                        const double cn = fma(cs[j], cd[j], -(sn[j] * sd[j]));
                        sn[j] = fma(sn[j], cd[j], cs[j] * sd[j]);
                        cs[j] = cn;
                    }
                    if (last_group) xrow[k] = tone_out<O>(v);
                    else if constexpr (GROUPS) partial[k] = v;
                }
            }
            // every lane moves the angles to the next segment's first sample
#pragma unroll
            for (int j = 0; j < MAXT; j++) {
                if (j < nt) A[j] = real_add[j] ? A[j] + dl[j] : fma((double)seg, step[j], A[j]);
            }
            pos += seg;
        }
    }
}

template <typename O, int MAXT, int STEPS, bool GROUPS>
__global__ __launch_bounds__(64) void tonegen_range_kernel(O *__restrict__ x, size_t x_stride, unsigned n, unsigned ntone,
                                                           const double *__restrict__ w, const double *__restrict__ dw,
                                                           const double *__restrict__ amp, const double *__restrict__ phi,
                                                           int use_sin)
{
    constexpr unsigned RANGE = 64 * STEPS;
    __shared__ double partial[GROUPS ? RANGE : 1];
    const unsigned nrange = (n + RANGE - 1) / RANGE;
    const unsigned chan = blockIdx.x / nrange;
    const size_t first = (size_t)(blockIdx.x - chan * nrange) * RANGE;
    const unsigned len = (unsigned)min((size_t)RANGE, (size_t)n - first);
    tone_range<O, MAXT, STEPS, GROUPS>(x, x_stride, ntone, w, dw, amp, phi, use_sin, chan, first, len, partial);
}


// ---- tables for the generator inside the fused FIR + spectrum kernel (layouts in fir_half.h) ----
// One launch prepares everything the fused kernel's tone source reads.
//
// Blocks below 4 * nchan * slots: the lane tables cos, sin (4 t dw), t < 256, a quarter of one
// (channel, tone slot) each.  Idle slots hold zeros (their p, q are zero as well).
//
// The other blocks, one lane per (channel, tile): every tone's value p and quadrature q at the
// window's first position (4096 tile - lead; before the row in its first tile, where the pair is
// turned back from sample 0), the increment's deviation from dw inside the tile, and cos, sin of
// 1024, 1, 2 and 3 increments; or the flag that marks the tile as hard.  A tile is computed in
// place when, for every tone, the increment is positive (or zero) and the angle non-negative,
// and either the angles stay below 64 (the rounded additions then follow a + k dw to 3e-11 over
// the window) or every addition of the window stays on one binade's grid, where a_k = a + k step
// exactly.  The wave then generates the windows of its hard tiles into the spill rows with the
// stand-alone generator's routine, all lanes on one tile at a time.
template <int MAXT>
__global__ __launch_bounds__(64) void tone_chain_tables_kernel(double *__restrict__ lane_tab, double *__restrict__ tile_tab,
                                                               unsigned *__restrict__ hard, float *__restrict__ spill, unsigned nx,
                                                               unsigned nchan, unsigned ntone, int slots, unsigned lead,
                                                               const double *__restrict__ w, const double *__restrict__ dw,
                                                               const double *__restrict__ amp, const double *__restrict__ phi,
                                                               int use_sin)
{
    const unsigned lane = threadIdx.x;
    const unsigned lane_blocks = nchan * (unsigned)slots * 4;
    if (blockIdx.x < lane_blocks) {
        const unsigned cs = blockIdx.x >> 2, t = (blockIdx.x & 3) * 64 + lane;
        const unsigned chan = cs / slots, j = cs - chan * slots;
        const bool on = j < ntone;
        const double d = on ? dw[(size_t)chan * ntone + j] : 0.0;
        double *lt = lane_tab + ((size_t)cs * 256 + t) * 2;
        const double a = (double)(4 * t) * d;
        lt[0] = on ? tone_cos(a) : 0.0;
        lt[1] = on ? tone_sin(a) : 0.0;
        return;
    }
    const unsigned ntile = nx / 4096;
    const size_t base = (size_t)(blockIdx.x - lane_blocks) * 64;
    const size_t id = base + lane;
    bool is_hard = false;
    if (id < (size_t)nchan * ntile) {
        const unsigned chan = (unsigned)(id / ntile), tile = (unsigned)(id - (size_t)chan * ntile);
        const unsigned long long start = (unsigned long long)tile * 4096;
        const unsigned long long k0 = start >= lead ? start - lead : 0;   // first sample of the window that exists
        const unsigned back = (unsigned)(k0 + lead - start);              // window positions before it
        const unsigned long long nadd = start + 4095 - k0;                // additions performed inside the window
        double *tt = tile_tab + id * (size_t)slots * 12;
        for (int j = 0; j < slots; j++) {
            double p = 0.0, q = 0.0, delta = 0.0, cg = 0.0, sg = 0.0, ce[3] = {0.0, 0.0, 0.0}, se[3] = {0.0, 0.0, 0.0};
            if (j < (int)ntone && !is_hard) {
                const size_t s = (size_t)chan * ntone + j;
                const double d = dw[s], w0 = w[s];
                double a = w0;
                if (d == 0.0) {
                    if (!(fabs(w0) <= 1.7e308)) is_hard = true;
                } else if (!(d > 0.0) || !(w0 >= 0.0) || !(w0 <= 1.7e308)) {
                    is_hard = true;
                } else {
                    a = advance(w0, d, k0);
                    if (!(a + (double)(nadd + 1) * d < 64.0)) {
                        double step;
                        const unsigned long long run = fixed_run(a, d, step);
                        if (run >= nadd) delta = step - d;  // both within a factor of two of each other: exact
                        else is_hard = true;
                        if (fabs(delta) * 4400.0 > 2e-3) is_hard = true;  // second-order drift correction good to 1.3e-9
                    }
                }
                if (!is_hard) {
                    const double inc = d + delta;
                    const double ang = use_sin ? a : a + phi[s];
                    const double s0 = tone_sin(ang), c0 = tone_cos(ang);
                    const double am = use_sin ? 1.0 : amp[s];
                    p = use_sin ? s0 : am * c0;
                    q = use_sin ? c0 : -am * s0;
                    if (back) {
                        const double b = (double)back * inc;
                        const double cb = tone_cos(b), sb = tone_sin(b);
                        const double pb = fma(-q, sb, p * cb);
                        q = fma(p, sb, q * cb);
                        p = pb;
                    }
                    cg = tone_cos(1024.0 * inc);
                    sg = tone_sin(1024.0 * inc);
                    for (int e = 0; e < 3; e++) {
                        ce[e] = tone_cos((double)(e + 1) * inc);
                        se[e] = tone_sin((double)(e + 1) * inc);
                    }
                }
            }
            tt[12 * j] = p;
            tt[12 * j + 1] = q;
            tt[12 * j + 2] = delta;
            tt[12 * j + 3] = cg;
            tt[12 * j + 4] = sg;
            tt[12 * j + 5] = 0.0;
            for (int e = 0; e < 3; e++) {
                tt[12 * j + 6 + 2 * e] = ce[e];
                tt[12 * j + 7 + 2 * e] = se[e];
            }
        }
        hard[id] = is_hard ? 1u : 0u;
    }
    unsigned long long todo = __ballot(is_hard);
    while (todo) {
        const int bit = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const size_t hid = base + (unsigned)bit;
        const unsigned chan = (unsigned)(hid / ntile), tile = (unsigned)(hid - (size_t)chan * ntile);
        const size_t start = (size_t)tile * 4096;
        const size_t first = start >= lead ? start - lead : 0;
        tone_range<float, MAXT, kToneSteps, false>(spill, nx, ntone, w, dw, amp, phi, use_sin, chan, first,
                                                   (unsigned)(start + 4096 - first), nullptr);
    }
}

}  // namespace

namespace dspmi {

static size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t tone_chain_work_bytes(unsigned nx, unsigned nchan, int slots)
{
    const size_t tiles = (size_t)nchan * (nx / 4096);
    return up256((size_t)nchan * slots * 512 * sizeof(double)) + up256(tiles * slots * 12 * sizeof(double)) +
           up256(tiles * sizeof(unsigned)) + up256((size_t)nchan * nx * sizeof(float));
}

ToneChainTables tone_chain_carve(void *work, unsigned nx, unsigned nchan, int slots)
{
    const size_t tiles = (size_t)nchan * (nx / 4096);
    char *b = static_cast<char *>(work);
    ToneChainTables t;
    t.lane_tab = reinterpret_cast<double *>(b);
    b += up256((size_t)nchan * slots * 512 * sizeof(double));
    t.tile_tab = reinterpret_cast<double *>(b);
    b += up256(tiles * slots * 12 * sizeof(double));
    t.hard = reinterpret_cast<unsigned *>(b);
    b += up256(tiles * sizeof(unsigned));
    t.spill = reinterpret_cast<float *>(b);
    return t;
}

int tone_chain_prepare(const ToneChainTables &tab, unsigned nx, unsigned nchan, unsigned ntone, int slots, unsigned lead,
                       const double *w, const double *dw, const double *amp, const double *phi, int use_sin, hipStream_t s)
{
    const size_t tiles = (size_t)nchan * (nx / 4096);
    if (use_sin && (!amp || !phi)) amp = phi = dw;  // never read in the sine form
    const size_t blocks = (size_t)nchan * slots * 4 + (tiles + 63) / 64;
    if (blocks > 0x7fffffffull) return fail(DSPMI_EINVAL, "tone_fir_spectrum: too many tiles for one launch");
    auto kern = ntone == 1 ? tone_chain_tables_kernel<1> : ntone == 2 ? tone_chain_tables_kernel<2> : tone_chain_tables_kernel<4>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), 0, s, tab.lane_tab, tab.tile_tab, tab.hard, tab.spill, nx, nchan,
                       ntone, slots, lead, w, dw, amp, phi, use_sin);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

}  // namespace dspmi

extern "C" {

int dspmi_power_spectrum_f32(float *ps, const void *bins, unsigned nbin, unsigned in_stride, size_t batch, int mode,
                             float scale, void *stream)
{
    if (batch == 0 || nbin == 0) return DSPMI_OK;
    if (!ps || !bins) return dspmi::fail(DSPMI_EINVAL, "power_spectrum: null buffer");
    if (mode < 0 || mode > 3) return dspmi::fail(DSPMI_EINVAL, "power_spectrum: mode must be 0..3");
    if (in_stride < nbin) return dspmi::fail(DSPMI_EINVAL, "power_spectrum: in_stride < nbin");
    hipStream_t s = static_cast<hipStream_t>(stream);
    size_t blocks = (batch + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const C32 *b = static_cast<const C32 *>(bins);
    if (nbin <= 64 * 8)
        hipLaunchKernelGGL(power_spectrum_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, ps, b, nbin, in_stride, batch, mode, scale);
    else if (nbin <= 64 * 33)
        hipLaunchKernelGGL(power_spectrum_kernel<33>, dim3((unsigned)blocks), dim3(256), 0, s, ps, b, nbin, in_stride, batch, mode, scale);
    else
        hipLaunchKernelGGL(power_spectrum_long_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, ps,
                           static_cast<const float *>(bins), nbin, in_stride, batch, mode, scale);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

int dspmi_phase_f32(float *phi, const void *bins, unsigned nbin, unsigned in_stride, size_t batch, void *stream)
{
    if (batch == 0 || nbin == 0) return DSPMI_OK;
    if (!phi || !bins) return dspmi::fail(DSPMI_EINVAL, "phase: null buffer");
    if (in_stride < nbin) return dspmi::fail(DSPMI_EINVAL, "phase: in_stride < nbin");
    size_t blocks = (batch * (size_t)nbin + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(phase_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), phi,
                       static_cast<const float *>(bins), nbin, in_stride, batch);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

}  // extern "C"

namespace {

template <typename O>
int tonegen_run(O *x, size_t x_stride, unsigned n, unsigned nchan, unsigned ntone, double *w, const double *dw,
                const double *amp, const double *phi, int use_sin, unsigned flags, void *stream)
{
    if (n == 0 || nchan == 0) return DSPMI_OK;
    if (!x || !w || !dw || (!use_sin && (!amp || !phi))) return dspmi::fail(DSPMI_EINVAL, "tonegen: null buffer");
    if (ntone == 0 || ntone > 32)
        return dspmi::fail(DSPMI_ENOSYS, "tonegen: 1..32 tones per channel, as the reference's generator holds (got %u)", ntone);
    if (x_stride < n) return dspmi::fail(DSPMI_EINVAL, "tonegen: x_stride < n");
    if (use_sin && (!amp || !phi)) amp = phi = dw;  // never read in the sine form
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (flags & DSPMI_TONEGEN_SEQUENTIAL) {
        auto seq = ntone <= 8 ? tonegen_kernel<O, 8> : tonegen_kernel<O, 32>;
        hipLaunchKernelGGL(seq, dim3((nchan + 63) / 64), dim3(64), 0, s, x, x_stride, n, nchan, ntone, w, dw, amp, phi, use_sin);
        DSPMI_HIP(hipGetLastError());
        return DSPMI_OK;
    }
    const unsigned nseq = nchan * ntone;
    // up to 8 tones: 16384 samples per wave; more: groups of 8 tones over 2048-sample ranges
    // (a range's running sums wait in 16 KB of LDS between groups)
    const unsigned range = ntone <= 8 ? kToneRange : 64 * kToneStepsGrouped;
    const unsigned long long nrange = (n + range - 1) / range;
    if (nrange * nchan > 0x7fffffffull) return dspmi::fail(DSPMI_EINVAL, "tonegen: too many sample ranges for one launch");
    auto kern = ntone == 1 ? tonegen_range_kernel<O, 1, kToneSteps, false> : ntone == 2 ? tonegen_range_kernel<O, 2, kToneSteps, false>
              : ntone <= 4 ? tonegen_range_kernel<O, 4, kToneSteps, false>
              : ntone <= 8 ? tonegen_range_kernel<O, 8, kToneSteps, false>
                           : tonegen_range_kernel<O, 8, kToneStepsGrouped, true>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nrange * nchan)), dim3(64), 0, s, x, x_stride, n, ntone, w, dw, amp, phi, use_sin);
    hipLaunchKernelGGL(tonegen_finish_kernel, dim3((nseq + 63) / 64), dim3(64), 0, s, w, dw, n, nseq);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

}  // namespace

extern "C" {

int dspmi_tonegen_f32(float *x, size_t x_stride, unsigned n, unsigned nchan, unsigned ntone, double *w, const double *dw,
                      const double *amp, const double *phi, int use_sin, unsigned flags, void *stream)
{
    return tonegen_run<float>(x, x_stride, n, nchan, ntone, w, dw, amp, phi, use_sin, flags, stream);
}

int dspmi_tonegen_c32(void *x, size_t x_stride, unsigned n, unsigned nchan, unsigned ntone, double *w, const double *dw,
                      const double *amp, const double *phi, int use_sin, unsigned flags, void *stream)
{
    if (reinterpret_cast<uintptr_t>(x) & 7) return dspmi::fail(DSPMI_EINVAL, "tonegen: output must be 8-byte aligned");
    return tonegen_run<C32>(static_cast<C32 *>(x), x_stride, n, nchan, ntone, w, dw, amp, phi, use_sin, flags, stream);
}

int dspmi_tonegen_s16x2(int16_t *x, size_t x_stride, unsigned n, unsigned nchan, unsigned ntone, double *w, const double *dw,
                        const double *amp, const double *phi, int use_sin, unsigned flags, void *stream)
{
    if (reinterpret_cast<uintptr_t>(x) & 3) return dspmi::fail(DSPMI_EINVAL, "tonegen: output must be 4-byte aligned");
    return tonegen_run<int>(reinterpret_cast<int *>(x), x_stride, n, nchan, ntone, w, dw, amp, phi, use_sin, flags, stream);
}

}  // extern "C"

// Double-precision forms used by the host wrappers (one frame per call there).
namespace dspmi {

int power_spectrum_f64(double *ps, const double *bins, unsigned nbin, unsigned in_stride, size_t batch, int mode, double scale,
                       hipStream_t s)
{
    size_t blocks = (batch + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(power_spectrum_long_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, ps, bins, nbin, in_stride,
                       batch, mode, scale);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

int phase_f64(double *phi, const double *bins, unsigned nbin, unsigned in_stride, size_t batch, hipStream_t s)
{
    size_t blocks = (batch * (size_t)nbin + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(phase_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, phi, bins, nbin, in_stride, batch);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

}  // namespace dspmi
