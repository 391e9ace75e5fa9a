// FFT plans: the counterpart of fftw_plan_dft_1d / _r2c_1d / _c2r_1d as the
// reference uses them (fftc/main.c:33-42, fft/src/main.c:156-166,
// util/filter/filter.c:86-103), over the batched kernels of fft_stockham.hip.
//
//  * power-of-two lengths run the Stockham kernels directly, up to the longest single-workgroup
//    kernel; longer ones are split n = n1 * n2 and run as two batched passes of those kernels
//    around transposes (the "four-step" scheme), so every length plans;
//  * any other length (the demos use 480 and 482: fftc/main.c:137, fir/src/main.c:154)
//    runs Bluestein's chirp-z: x[n]*c[n] convolved with conj(c) through two
//    power-of-two transforms of length M >= 2n-1, c[n] = exp(-i*pi*n^2/N) -- in one fused
//    kernel for f32 (fft_stockham.hip), as five kernels for f64;
//  * real transforms of even length n use one complex transform of n/2 points (a power-of-two
//    kernel, or a Bluestein plan of its own) plus an O(n) split / merge step -- inside the
//    transform kernel where that exists; odd lengths go through the full complex transform.
//
// All transforms are unnormalised in both directions (FFTW's convention).
#include "common.h"
#include "fft_internal.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

template <typename T>
struct alignas(2 * sizeof(T)) C {
    T re, im;
};

template <typename T> __device__ __forceinline__ C<T> cmul(C<T> a, C<T> b)
{
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

#define GRID_STRIDE(i, total) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (size_t)gridDim.x * blockDim.x)

// ---- real <-> half-complex, n even ------------------------------------------------

// Z = FFT_{n/2}(x[2m] + i x[2m+1]);  X[k] = E[k] + W_n^k O[k], k = 0..n/2, with
// E[k] = (Z[k] + conj Z[h-k])/2, O[k] = -i (Z[k] - conj Z[h-k])/2, h = n/2, Z[h] = Z[0].
template <typename T>
__global__ __launch_bounds__(256) void r2c_split_kernel(const C<T> *__restrict__ z, C<T> *__restrict__ xout,
                                                        const C<T> *__restrict__ twn, unsigned n, size_t batch)
{
    const unsigned h = n / 2;
    GRID_STRIDE(i, batch * (size_t)(h + 1)) {
        const size_t f = i / (h + 1);
        const unsigned k = (unsigned)(i - f * (h + 1));
        const C<T> *zf = z + f * h;
        const C<T> a = zf[k == h ? 0 : k];
        const C<T> b0 = zf[(h - k) % h];
        const C<T> b = {b0.re, -b0.im};
        const C<T> e = {(a.re + b.re) * T(0.5), (a.im + b.im) * T(0.5)};
        const C<T> d = {(a.re - b.re) * T(0.5), (a.im - b.im) * T(0.5)};
        const C<T> o = {d.im, -d.re};                            // -i * d
        const C<T> w = (k == h) ? C<T>{T(-1), T(0)} : twn[k];    // exp(-2 pi i k / n)
        const C<T> wo = cmul(w, o);
        xout[f * (h + 1) + k] = {e.re + wo.re, e.im + wo.im};
    }
}

// Unnormalised c2r: n * x equals the n/2-point backward transform of
// Z[k] = (X[k] + conj X[h-k]) + i W_n^{-k} (X[k] - conj X[h-k]).
template <typename T>
__global__ __launch_bounds__(256) void c2r_merge_kernel(const C<T> *__restrict__ xin, C<T> *__restrict__ z,
                                                        const C<T> *__restrict__ twn, unsigned n, size_t batch)
{
    const unsigned h = n / 2;
    GRID_STRIDE(i, batch * (size_t)h) {
        const size_t f = i / h;
        const unsigned k = (unsigned)(i - f * h);
        const C<T> *xf = xin + f * (h + 1);
        C<T> a = xf[k];
        C<T> b0 = xf[h - k];
        // like FFTW's c2r, ignore the imaginary parts of the DC and Nyquist bins
        if (k == 0) { a.im = T(0); b0.im = T(0); }
        const C<T> b = {b0.re, -b0.im};
        const C<T> s = {a.re + b.re, a.im + b.im};
        const C<T> d = {a.re - b.re, a.im - b.im};
        const C<T> w0 = twn[k];
        const C<T> w = {w0.re, -w0.im};  // exp(+2 pi i k / n)
        const C<T> wd = cmul(w, d);
        z[f * h + k] = {s.re - wd.im, s.im + wd.re};  // s + i * wd
    }
}

// ---- real transforms of any length through the complex transform -------------------

template <typename T>
__global__ __launch_bounds__(256) void real_to_complex_kernel(const T *__restrict__ x, C<T> *__restrict__ z, size_t total)
{
    GRID_STRIDE(i, total) z[i] = {x[i], T(0)};
}

template <typename T>
__global__ __launch_bounds__(256) void take_bins_kernel(const C<T> *__restrict__ full, C<T> *__restrict__ bins, unsigned n,
                                                        size_t batch)
{
    const unsigned nb = n / 2 + 1;
    GRID_STRIDE(i, batch * (size_t)nb) {
        const size_t f = i / nb;
        bins[i] = full[f * n + (i - f * nb)];
    }
}

// Hermitian extension of bins 0..n/2 to n points (imaginary parts of DC / Nyquist dropped).
template <typename T>
__global__ __launch_bounds__(256) void hermitian_extend_kernel(const C<T> *__restrict__ bins, C<T> *__restrict__ full,
                                                               unsigned n, size_t batch)
{
    const unsigned nb = n / 2 + 1;
    GRID_STRIDE(i, batch * (size_t)n) {
        const size_t f = i / n;
        const unsigned k = (unsigned)(i - f * n);
        C<T> v;
        if (k < nb) {
            v = bins[f * nb + k];
            if (k == 0 || (n % 2 == 0 && k == n / 2)) v.im = T(0);
        } else {
            v = bins[f * nb + (n - k)];
            v.im = -v.im;
        }
        full[i] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void take_real_kernel(const C<T> *__restrict__ z, T *__restrict__ x, size_t total)
{
    GRID_STRIDE(i, total) x[i] = z[i].re;
}

// ---- long power-of-two lengths: n = n1 * n2 ------------------------------------------

// out[f][c][r] = in[f][r][c] (* W_n^(r c) when TWIDDLE), in: [rows][cols] per frame; 32 x 32
// tiles through LDS so that reads and writes are both contiguous.  W_n^e is the product of two
// table entries, A[e >> shift] = W_n^(e with its low bits cleared) and B[e & mask] (about
// sqrt(n) entries each, computed in long double): one extra rounding, no table of n entries.
// conj_tw conjugates the factor (backward transform).
template <typename T, bool TWIDDLE>
__global__ __launch_bounds__(256) void transpose_kernel(const C<T> *__restrict__ in, C<T> *__restrict__ out, unsigned rows,
                                                        unsigned cols, const C<T> *__restrict__ twa,
                                                        const C<T> *__restrict__ twb, unsigned shift, int conj_tw)
{
    __shared__ C<T> tile[32][33];
    const size_t frame = blockIdx.z;
    const C<T> *src = in + frame * (size_t)rows * cols;
    C<T> *dst = out + frame * (size_t)rows * cols;
    const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const unsigned c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned r = r0 + ty + 8 * i, c = c0 + tx;
        C<T> v = src[(size_t)r * cols + c];
        if constexpr (TWIDDLE) {
            const unsigned long long e = (unsigned long long)r * c;  // < n
            C<T> w = cmul(twa[e >> shift], twb[e & ((1ull << shift) - 1)]);
            if (conj_tw) w.im = -w.im;
            v = cmul(v, w);
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned c = c0 + ty + 8 * i, r = r0 + tx;
        dst[(size_t)c * rows + r] = tile[tx][ty + 8 * i];
    }
}

// ---- Bluestein ------------------------------------------------------------------------

// work[f][m] = x[f][m] * c[m] for m < n, 0 up to M; conj_in conjugates x (backward transform)
template <typename T>
__global__ __launch_bounds__(256) void bluestein_pre_kernel(const C<T> *__restrict__ x, const C<T> *__restrict__ chirp,
                                                            C<T> *__restrict__ work, unsigned n, unsigned M, size_t batch,
                                                            int conj_in)
{
    GRID_STRIDE(i, batch * (size_t)M) {
        const size_t f = i / M;
        const unsigned m = (unsigned)(i - f * M);
        C<T> v = {T(0), T(0)};
        if (m < n) {
            v = x[f * n + m];
            if (conj_in) v.im = -v.im;
            v = cmul(v, chirp[m]);
        }
        work[i] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bluestein_mul_kernel(C<T> *__restrict__ work, const C<T> *__restrict__ filt,
                                                            unsigned M, size_t batch)
{
    GRID_STRIDE(i, batch * (size_t)M) work[i] = cmul(work[i], filt[i % M]);
}

template <typename T>
__global__ __launch_bounds__(256) void bluestein_post_kernel(const C<T> *__restrict__ work, const C<T> *__restrict__ chirp,
                                                             C<T> *__restrict__ out, unsigned n, unsigned M, size_t batch,
                                                             int conj_out)
{
    GRID_STRIDE(i, batch * (size_t)n) {
        const size_t f = i / n;
        const unsigned k = (unsigned)(i - f * n);
        C<T> v = cmul(work[f * M + k], chirp[k]);
        if (conj_out) v.im = -v.im;
        out[i] = v;
    }
}

// ---- element-wise helpers for the host wrappers -------------------------------------

// bins[f][k] *= g[k]  (real gain per bin; util/filter/filter.c:92-96)
template <typename T>
__global__ __launch_bounds__(256) void bin_gain_kernel(C<T> *__restrict__ bins, const T *__restrict__ g, unsigned nbin,
                                                       size_t batch)
{
    GRID_STRIDE(i, batch * (size_t)nbin) {
        const T gg = g[i % nbin];
        C<T> v = bins[i];
        v.re *= gg;
        v.im *= gg;
        bins[i] = v;
    }
}

// out[f][i] = 2*|bins[f][i]| / scale, i < nout (fftc/main.c:80-105, fir/src/main.c:246-251)
template <typename T>
__global__ __launch_bounds__(256) void ps_abs_kernel(const C<T> *__restrict__ bins, T *__restrict__ out, unsigned nout,
                                                     T scale, size_t batch, unsigned in_stride)
{
    GRID_STRIDE(i, batch * (size_t)nout) {
        const size_t f = i / nout;
        const C<T> v = bins[f * in_stride + (i - f * nout)];
        out[i] = (T(2) * sqrt(v.re * v.re + v.im * v.im)) / scale;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T *__restrict__ x, T divisor, size_t total)
{
    GRID_STRIDE(i, total) x[i] = x[i] / divisor;
}

inline unsigned ew_grid(size_t total, int num_cu)
{
    size_t g = (total + 255) / 256;
    const size_t cap = (size_t)num_cu * 8;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

template <typename T>
std::vector<C<T>> unit_circle(unsigned n, unsigned denom_mult)
{
    // exp(-2 pi i k / (n * denom_mult)) ... used with denom_mult = 1 for twiddles
    std::vector<C<T>> tw(n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (unsigned k = 0; k < n; k++) {
        const long double a = -two_pi * (long double)k / ((long double)n * denom_mult);
        tw[k].re = (T)cosl(a);
        tw[k].im = (T)sinl(a);
    }
    return tw;
}

template <typename T>
int upload(void **dev, const std::vector<C<T>> &v)
{
    DSPMI_HIP(hipMalloc(dev, v.size() * sizeof(C<T>)));
    DSPMI_HIP(hipMemcpy(*dev, v.data(), v.size() * sizeof(C<T>), hipMemcpyHostToDevice));
    return DSPMI_OK;
}

int upload_twiddles(void **dev, unsigned n, int precision)
{
    return precision == DSPMI_F64 ? upload<double>(dev, unit_circle<double>(n, 1)) : upload<float>(dev, unit_circle<float>(n, 1));
}

// chirp c[m] = exp(-i pi m^2 / n), the angle reduced exactly: m^2 mod 2n
template <typename T>
std::vector<C<T>> chirp(unsigned n)
{
    std::vector<C<T>> c(n);
    const long double pi = 3.141592653589793238462643383279502884L;
    for (unsigned m = 0; m < n; m++) {
        const unsigned long long r = ((unsigned long long)m * m) % (2ull * n);
        const long double a = -pi * (long double)r / (long double)n;
        c[m].re = (T)cosl(a);
        c[m].im = (T)sinl(a);
    }
    return c;
}

// interleaved 16-bit stereo frames -> mono floats, (left + right) / 2 (alsa/main.c:464-468)
__global__ __launch_bounds__(256) void s16x2_to_mono_kernel(float *__restrict__ mono, const int *__restrict__ frames, size_t total)
{
    GRID_STRIDE(i, total) {
        const int f = frames[i];
        mono[i] = (float)((short)(f & 0xffff) + (f >> 16)) * 0.5f;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// scratch of the entry point that allocates nothing
// ---------------------------------------------------------------------------

// dspmi_fft_exec_c2c_noalloc runs the multi-kernel any-length transform with its M-point work
// frames in this array instead of a buffer from hipMalloc.  Every device the library runs on
// holds its own copy, at its own address, until the process ends.  In host code the array's
// name is only a handle for hipGetSymbolAddress: the address kernels are given is the one
// that call returns for the current device, kept in g_noalloc_ptr.
constexpr size_t kNoAllocScratchBytes = (size_t)DSPMI_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

namespace {

constexpr int kMaxDevices = 64;
std::atomic<void *> g_noalloc_ptr[kMaxDevices];

// Device address of `scratch` on `device`, which must be the current device.  Resolved once per
// device (at plan creation, else on the entry point's first call there).
int noalloc_scratch(int device, void **out)
{
    if (device < 0 || device >= kMaxDevices) return dspmi::fail(DSPMI_EINVAL, "noalloc scratch: device %d out of range", device);
    void *p = g_noalloc_ptr[device].load(std::memory_order_acquire);
    if (!p) {
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return dspmi::fail(DSPMI_EIO, "noalloc scratch: hipGetSymbolAddress on device %d -> %s", device, hipGetErrorString(e));
        g_noalloc_ptr[device].store(p, std::memory_order_release);
    }
    *out = p;
    return DSPMI_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------

struct dspmi_fft_plan {
    int device = 0;
    int precision = 0;
    unsigned n = 0;
    int num_cu = 256;
    size_t esz = 8;        // bytes per complex element
    void *tw_n = nullptr;  // exp(-2 pi i k / n), k < n (c2c twiddles; also the r2c split factors)
    void *tw_half = nullptr;
    const dspmi::FftKernel *kern = nullptr;       // power-of-two n
    const dspmi::FftKernel *kern_half = nullptr;  // n even with a power-of-two half
    // Bluestein (any other n)
    unsigned M = 0;
    const dspmi::FftKernel *kern_M = nullptr;
    void *tw_M = nullptr, *chirp = nullptr, *filt = nullptr;
    dspmi::bluestein_launch_fn blu = nullptr;  // fused form, or nullptr
    // even n whose half is not a power of two: real transforms run the n/2-point complex plan
    // (itself Bluestein) around the same split / merge kernels
    dspmi_fft_plan *half_plan = nullptr;
    // power-of-two n beyond the longest single kernel: n = n1 * n2, two batched passes
    unsigned n1 = 0, n2 = 0, tw_shift = 0;
    unsigned t1 = 0, t2 = 0, t3 = 0;  // n = t1 t2 t3 beyond 2^20 points: three column passes
    void *tw_t1 = nullptr, *tw_t2 = nullptr, *tw_t3 = nullptr;
    const dspmi::FftKernel *kern_n1 = nullptr, *kern_n2 = nullptr;
    void *tw_n1 = nullptr, *tw_n2 = nullptr, *tw_hi = nullptr, *tw_lo = nullptr;
    // Bluestein whose convolution length M is itself beyond a single kernel
    dspmi_fft_plan *plan_M = nullptr;
    // scratch, grown on demand (allocation happens in exec only when a batch outgrows it)
    // 0: Bluestein, 1: real transforms, 2: spectrum bins, 3: converted input, 4: long power-of-two lengths
    void *work[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t work_bytes[5] = {0, 0, 0, 0, 0};
    char name[64] = "";
};

namespace {

int ensure_work(dspmi_fft_plan *pl, int which, size_t bytes)
{
    if (bytes <= pl->work_bytes[which]) return DSPMI_OK;
    if (pl->work[which]) DSPMI_HIP(hipFree(pl->work[which]));
    pl->work[which] = nullptr;
    pl->work_bytes[which] = 0;
    DSPMI_HIP(hipMalloc(&pl->work[which], bytes));
    pl->work_bytes[which] = bytes;
    return DSPMI_OK;
}

int run_pow2(const dspmi::FftKernel *k, const void *in, void *out, const void *tw, size_t batch, int backward, size_t esz,
             int num_cu, hipStream_t s)
{
    // a batch that cannot stay in the 256 MiB last-level cache is a read-once / write-once
    // stream: bypass the caches' allocation with nontemporal accesses
    const bool stream_policy = batch * (size_t)k->n * esz >= ((size_t)256 << 20);
    return (stream_policy ? k->fn_stream : k->fn)(in, out, tw, batch, backward, num_cu, s);
}

// n = n1 * n2 with x[a + n1 b] and X[n2 k1 + k2]:
//   X[n2 k1 + k2] = sum_a W_n1^(a k1) . W_n^(a k2) . sum_b x[a + n1 b] W_n2^(b k2)
// as transpose, n1 batched n2-point transforms, transpose with the W_n^(a k2) factors, n2 batched
// n1-point transforms, transpose.  Frames go through in groups that keep the two work images
// within `cap` bytes: 128 MiB of plan-owned scratch on the allocating path (256 MiB on the two-pass
// route; kept until the plan is destroyed), or the caller's fixed work space.
template <typename T>
int four_step(dspmi_fft_plan *pl, const C<T> *in, C<T> *out, size_t batch, int backward, hipStream_t s, void *fixed_work,
              size_t fixed_bytes)
{
    const size_t n = pl->n;
    size_t cap = fixed_work ? fixed_bytes : ((size_t)128 << 20);
    // Both factors 128 .. 1024 points (n up to 2^20): two passes, the transposes folded into the
    // transforms' loads and stores (fft_c2c_colreg_kernel); beyond, three factors of 128 or 256
    // points and three such passes.  One work frame per frame, in groups of 256 MiB (larger
    // launches count for more than what stays in the last-level cache between the passes).
    // DSPMI_FFT_FIVE_PASS=1 keeps the five-pass route below for A/B.
    {
        const int prec = sizeof(T) == 8 ? DSPMI_F64 : DSPMI_F32;
        const dspmi::fft_bundle_fn pass1 = dspmi::fft_bundle_find(pl->n2, prec), pass2 = dspmi::fft_bundle_find(pl->n1, prec);
        const dspmi::fft_bundle_fn pa = pl->t3 ? dspmi::fft_bundle_find(pl->t3, prec) : nullptr,
                                   pb = pl->t3 ? dspmi::fft_bundle_find(pl->t2, prec) : nullptr,
                                   pc = pl->t3 ? dspmi::fft_bundle_find(pl->t1, prec) : nullptr;
        const bool two = pass1 && pass2, three = !two && pa && pb && pc;
        if ((two || three) && !getenv("DSPMI_FFT_FIVE_PASS")) {
            if (!fixed_work) cap = (size_t)256 << 20;
            const size_t fb = n * sizeof(C<T>);
            if (fixed_work && fb > cap)
                return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c_noalloc: one frame of n = %u needs %zu bytes of work space, the static scratch holds %zu",
                                   pl->n, fb, cap);
            size_t group = cap / fb;
            if (group < 1) group = 1;
            if (group > batch) group = batch;
            C<T> *w = static_cast<C<T> *>(fixed_work);
            if (!fixed_work) {
                int rc = ensure_work(pl, 4, group * fb);
                if (rc != DSPMI_OK) return rc;
                w = static_cast<C<T> *>(pl->work[4]);
            }
            for (size_t done = 0; done < batch; done += group) {
                const size_t g = batch - done < group ? batch - done : group;
                int rc;
                if (two) {
                    rc = pass1(in + done * n, w, pl->tw_n2, g, pl->n1, pl->tw_hi, pl->tw_lo, pl->tw_shift, 1, backward, 0, 0, s);
                    if (rc != DSPMI_OK) return rc;
                    rc = pass2(w, out + done * n, pl->tw_n1, g, pl->n2, pl->tw_hi, pl->tw_lo, pl->tw_shift, 2, backward, 0, 0, s);
                    if (rc != DSPMI_OK) return rc;
                    continue;
                }
                // x[a + t1 (b + t2 c)]: over c, over b (one [t2][t1] frame per kc), over a
                rc = pa(in + done * n, w, pl->tw_t3, g, pl->t1 * pl->t2, pl->tw_hi, pl->tw_lo, pl->tw_shift, 3, backward, pl->t1, 0, s);
                if (rc != DSPMI_OK) return rc;
                rc = pb(w, w, pl->tw_t2, g * pl->t3, pl->t1, pl->tw_hi, pl->tw_lo, pl->tw_shift, 4, backward, pl->t3, 0, s);
                if (rc != DSPMI_OK) return rc;
                rc = pc(w, out + done * n, pl->tw_t1, g, pl->t2 * pl->t3, pl->tw_hi, pl->tw_lo, pl->tw_shift, 5, backward, pl->t2, pl->t3, s);
                if (rc != DSPMI_OK) return rc;
            }
            return DSPMI_OK;
        }
    }
    const size_t frame_bytes = 2 * n * sizeof(C<T>);
    if (fixed_work && frame_bytes > cap)
        return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c_noalloc: one frame of n = %u needs %zu bytes of work space, the static scratch holds %zu",
                           pl->n, frame_bytes, cap);
    size_t group = cap / frame_bytes;
    if (group < 1) group = 1;
    if (group > batch) group = batch;
    if (group > 65535) group = 65535;  // grid z
    C<T> *w0;
    if (fixed_work) {
        w0 = static_cast<C<T> *>(fixed_work);
    } else {
        int rc = ensure_work(pl, 4, group * frame_bytes);
        if (rc != DSPMI_OK) return rc;
        w0 = static_cast<C<T> *>(pl->work[4]);
    }
    C<T> *w1 = w0 + group * n;
    const C<T> *hi = static_cast<const C<T> *>(pl->tw_hi), *lo = static_cast<const C<T> *>(pl->tw_lo);
    for (size_t done = 0; done < batch; done += group) {
        const size_t g = batch - done < group ? batch - done : group;
        const C<T> *src = in + done * n;
        C<T> *dst = out + done * n;
        // x as [n2][n1] -> [n1][n2]
        hipLaunchKernelGGL((transpose_kernel<T, false>), dim3(pl->n1 / 32, pl->n2 / 32, (unsigned)g), dim3(256), 0, s, src, w0, pl->n2,
                           pl->n1, hi, lo, pl->tw_shift, 0);
        DSPMI_HIP(hipGetLastError());
        int rc = run_pow2(pl->kern_n2, w0, w0, pl->tw_n2, g * pl->n1, backward, pl->esz, pl->num_cu, s);
        if (rc != DSPMI_OK) return rc;
        // [n1][n2] -> [n2][n1], times W_n^(a k2)
        hipLaunchKernelGGL((transpose_kernel<T, true>), dim3(pl->n2 / 32, pl->n1 / 32, (unsigned)g), dim3(256), 0, s, w0, w1, pl->n1,
                           pl->n2, hi, lo, pl->tw_shift, backward);
        DSPMI_HIP(hipGetLastError());
        rc = run_pow2(pl->kern_n1, w1, w1, pl->tw_n1, g * pl->n2, backward, pl->esz, pl->num_cu, s);
        if (rc != DSPMI_OK) return rc;
        // [n2][n1] (k2, k1) -> [n1][n2]: X[n2 k1 + k2] in natural order
        hipLaunchKernelGGL((transpose_kernel<T, false>), dim3(pl->n1 / 32, pl->n2 / 32, (unsigned)g), dim3(256), 0, s, w1, dst, pl->n2,
                           pl->n1, hi, lo, pl->tw_shift, 0);
        DSPMI_HIP(hipGetLastError());
    }
    return DSPMI_OK;
}

// fixed_work: work frames supplied by the caller (fixed_bytes of them) instead of the plan's
// own buffer, which an execute may have to grow -- nothing is allocated or freed then.
template <typename T>
int c2c_typed(dspmi_fft_plan *pl, const void *in, void *out, size_t batch, int backward, hipStream_t s,
              void *fixed_work = nullptr, size_t fixed_bytes = 0)
{
    if (pl->kern) return run_pow2(pl->kern, in, out, pl->tw_n, batch, backward, pl->esz, pl->num_cu, s);
    if (pl->kern_n1)
        return four_step<T>(pl, static_cast<const C<T> *>(in), static_cast<C<T> *>(out), batch, backward, s, fixed_work, fixed_bytes);
    // Bluestein: one kernel where it exists (DSPMI_BLUESTEIN_COMPOSED=1: the five separate ones)
    if (pl->blu && !getenv("DSPMI_BLUESTEIN_COMPOSED"))
        return pl->blu(in, out, pl->chirp, pl->filt, pl->tw_M, pl->n, batch, backward, pl->num_cu, s);
    const unsigned n = pl->n, M = pl->M;
    int rc = DSPMI_OK;
    C<T> *w;
    if (fixed_work) {
        if (pl->plan_M)
            return dspmi::fail(DSPMI_ENOSYS, "fft_exec_c2c_noalloc: n = %u runs over a %u-point convolution that needs two work "
                               "buffers; use dspmi_fft_exec_c2c", n, M);
        if (batch > fixed_bytes / ((size_t)M * sizeof(C<T>)))
            return dspmi::fail(DSPMI_EINVAL,
                               "fft_exec_c2c_noalloc: %zu frames of n = %u need %zu work frames of %u points, the static "
                               "scratch of %zu bytes holds %zu",
                               batch, n, batch, M, fixed_bytes, fixed_bytes / ((size_t)M * sizeof(C<T>)));
        w = static_cast<C<T> *>(fixed_work);
    } else {
        rc = ensure_work(pl, 0, batch * (size_t)M * sizeof(C<T>));
        if (rc != DSPMI_OK) return rc;
        w = static_cast<C<T> *>(pl->work[0]);
    }
    auto transform_M = [&](int back) {
        return pl->kern_M ? run_pow2(pl->kern_M, w, w, pl->tw_M, batch, back, pl->esz, pl->num_cu, s)
                          : c2c_typed<T>(pl->plan_M, w, w, batch, back, s);
    };
    hipLaunchKernelGGL(bluestein_pre_kernel<T>, dim3(ew_grid(batch * M, pl->num_cu)), dim3(256), 0, s,
                       static_cast<const C<T> *>(in), static_cast<const C<T> *>(pl->chirp), w, n, M, batch, backward);
    DSPMI_HIP(hipGetLastError());
    rc = transform_M(0);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(bluestein_mul_kernel<T>, dim3(ew_grid(batch * M, pl->num_cu)), dim3(256), 0, s, w,
                       static_cast<const C<T> *>(pl->filt), M, batch);
    DSPMI_HIP(hipGetLastError());
    rc = transform_M(1);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(bluestein_post_kernel<T>, dim3(ew_grid(batch * n, pl->num_cu)), dim3(256), 0, s, w,
                       static_cast<const C<T> *>(pl->chirp), static_cast<C<T> *>(out), n, M, batch, backward);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

// filt = FFT_M(b) / M with b[m] = conj(c[m]) for |m| < n laid out circularly
template <typename T>
int setup_bluestein(dspmi_fft_plan *pl)
{
    const unsigned n = pl->n, M = pl->M;
    std::vector<C<T>> c = chirp<T>(n);
    std::vector<C<T>> b(M, C<T>{T(0), T(0)});
    for (unsigned m = 0; m < n; m++) {
        // chirp in full precision again for the filter, scaled by 1/M
        const long double pi = 3.141592653589793238462643383279502884L;
        const unsigned long long r = ((unsigned long long)m * m) % (2ull * n);
        const long double a = pi * (long double)r / (long double)n;
        const C<T> v = {(T)(cosl(a) / (long double)M), (T)(sinl(a) / (long double)M)};
        b[m] = v;
        if (m) b[M - m] = v;
    }
    int rc = upload<T>(&pl->chirp, c);
    if (rc == DSPMI_OK) rc = upload<T>(&pl->filt, b);
    if (rc == DSPMI_OK && pl->kern_M) rc = upload_twiddles(&pl->tw_M, M, pl->precision);
    if (rc == DSPMI_OK)
        rc = pl->kern_M ? pl->kern_M->fn(pl->filt, pl->filt, pl->tw_M, 1, 0, pl->num_cu, nullptr)
                        : c2c_typed<T>(pl->plan_M, pl->filt, pl->filt, 1, 0, nullptr);
    if (rc == DSPMI_OK) DSPMI_HIP(hipStreamSynchronize(nullptr));
    return rc;
}

// Whether real transforms of this plan run the fused kernels (real <-> half-complex step inside
// the transform kernel; they exist from 128 complex points up, tools/real_probe.py).
// DSPMI_FFT_REAL_COMPOSED=1 selects the separate split / merge kernels everywhere (A/B knob).
bool fused_real(const dspmi_fft_plan *pl)
{
    return pl->kern_half && pl->kern_half->fn_r2c && pl->kern_half->fn_c2r && !getenv("DSPMI_FFT_REAL_COMPOSED");
}

bool ranges_overlap(const void *a, size_t a_bytes, const void *b, size_t b_bytes)
{
    const char *p = static_cast<const char *>(a), *q = static_cast<const char *>(b);
    return p < q + b_bytes && q < p + a_bytes;
}

template <typename T>
int r2c_typed(dspmi_fft_plan *pl, const void *in, void *out, size_t batch, hipStream_t s)
{
    const unsigned n = pl->n;
    if (n == 1) {
        hipLaunchKernelGGL(real_to_complex_kernel<T>, dim3(ew_grid(batch, pl->num_cu)), dim3(256), 0, s,
                           static_cast<const T *>(in), static_cast<C<T> *>(out), batch);
        DSPMI_HIP(hipGetLastError());
        return DSPMI_OK;
    }
    // transform + split in one kernel -- unless the output overlaps the input: frames of samples
    // and frames of bins have different lengths, so workgroups would overwrite each other's
    // input.  That case takes the two-kernel route below, which reads everything first.
    if (fused_real(pl) && !ranges_overlap(in, batch * (size_t)n * sizeof(T), out, batch * (size_t)(n / 2 + 1) * sizeof(C<T>)))
        return pl->kern_half->fn_r2c(in, out, pl->tw_half, pl->tw_n, batch, 1, n / 2 + 1, 0, 0.f, pl->num_cu, s);
    if (pl->kern_half) {
        int rc = ensure_work(pl, 1, batch * (size_t)(n / 2) * sizeof(C<T>));
        if (rc != DSPMI_OK) return rc;
        // the real frame [n] reinterpreted as n/2 complex points is the packed input
        rc = run_pow2(pl->kern_half, in, pl->work[1], pl->tw_half, batch, 0, pl->esz, pl->num_cu, s);
        if (rc != DSPMI_OK) return rc;
        hipLaunchKernelGGL(r2c_split_kernel<T>, dim3(ew_grid(batch * (n / 2 + 1), pl->num_cu)), dim3(256), 0, s,
                           static_cast<const C<T> *>(pl->work[1]), static_cast<C<T> *>(out),
                           static_cast<const C<T> *>(pl->tw_n), n, batch);
        DSPMI_HIP(hipGetLastError());
        return DSPMI_OK;
    }
    if (pl->half_plan) {
        int rc = ensure_work(pl, 1, batch * (size_t)(n / 2) * sizeof(C<T>));
        if (rc != DSPMI_OK) return rc;
        rc = c2c_typed<T>(pl->half_plan, in, pl->work[1], batch, 0, s);
        if (rc != DSPMI_OK) return rc;
        hipLaunchKernelGGL(r2c_split_kernel<T>, dim3(ew_grid(batch * (n / 2 + 1), pl->num_cu)), dim3(256), 0, s,
                           static_cast<const C<T> *>(pl->work[1]), static_cast<C<T> *>(out),
                           static_cast<const C<T> *>(pl->tw_n), n, batch);
        DSPMI_HIP(hipGetLastError());
        return DSPMI_OK;
    }
    // odd lengths: complexify, full transform, keep bins 0..n/2
    int rc = ensure_work(pl, 1, 2 * batch * (size_t)n * sizeof(C<T>));
    if (rc != DSPMI_OK) return rc;
    C<T> *za = static_cast<C<T> *>(pl->work[1]);
    C<T> *zb = za + batch * (size_t)n;
    hipLaunchKernelGGL(real_to_complex_kernel<T>, dim3(ew_grid(batch * n, pl->num_cu)), dim3(256), 0, s,
                       static_cast<const T *>(in), za, batch * (size_t)n);
    DSPMI_HIP(hipGetLastError());
    rc = c2c_typed<T>(pl, za, zb, batch, 0, s);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(take_bins_kernel<T>, dim3(ew_grid(batch * (n / 2 + 1), pl->num_cu)), dim3(256), 0, s, zb,
                       static_cast<C<T> *>(out), n, batch);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <typename T>
int c2r_typed(dspmi_fft_plan *pl, const void *in, void *out, size_t batch, hipStream_t s)
{
    const unsigned n = pl->n;
    if (n == 1) {
        hipLaunchKernelGGL(take_real_kernel<T>, dim3(ew_grid(batch, pl->num_cu)), dim3(256), 0, s,
                           static_cast<const C<T> *>(in), static_cast<T *>(out), batch);
        DSPMI_HIP(hipGetLastError());
        return DSPMI_OK;
    }
    if (fused_real(pl)) {
        // merge + transform in one kernel.  Frames of bins and frames of samples have
        // different lengths, so workgroups would trample each other's input in an overlapping
        // buffer: that case takes the two-kernel route below.
        if (!ranges_overlap(in, batch * (size_t)(n / 2 + 1) * sizeof(C<T>), out, batch * (size_t)n * sizeof(T)))
            return pl->kern_half->fn_c2r(in, out, pl->tw_half, pl->tw_n, batch, pl->num_cu, s);
    }
    if (pl->kern_half) {
        int rc = ensure_work(pl, 1, batch * (size_t)(n / 2) * sizeof(C<T>));
        if (rc != DSPMI_OK) return rc;
        hipLaunchKernelGGL(c2r_merge_kernel<T>, dim3(ew_grid(batch * (n / 2), pl->num_cu)), dim3(256), 0, s,
                           static_cast<const C<T> *>(in), static_cast<C<T> *>(pl->work[1]),
                           static_cast<const C<T> *>(pl->tw_n), n, batch);
        DSPMI_HIP(hipGetLastError());
        return run_pow2(pl->kern_half, pl->work[1], out, pl->tw_half, batch, 1, pl->esz, pl->num_cu, s);
    }
    if (pl->half_plan) {
        int rc = ensure_work(pl, 1, batch * (size_t)(n / 2) * sizeof(C<T>));
        if (rc != DSPMI_OK) return rc;
        hipLaunchKernelGGL(c2r_merge_kernel<T>, dim3(ew_grid(batch * (n / 2), pl->num_cu)), dim3(256), 0, s,
                           static_cast<const C<T> *>(in), static_cast<C<T> *>(pl->work[1]),
                           static_cast<const C<T> *>(pl->tw_n), n, batch);
        DSPMI_HIP(hipGetLastError());
        return c2c_typed<T>(pl->half_plan, pl->work[1], out, batch, 1, s);
    }
    int rc = ensure_work(pl, 1, 2 * batch * (size_t)n * sizeof(C<T>));
    if (rc != DSPMI_OK) return rc;
    C<T> *za = static_cast<C<T> *>(pl->work[1]);
    C<T> *zb = za + batch * (size_t)n;
    hipLaunchKernelGGL(hermitian_extend_kernel<T>, dim3(ew_grid(batch * n, pl->num_cu)), dim3(256), 0, s,
                       static_cast<const C<T> *>(in), za, n, batch);
    DSPMI_HIP(hipGetLastError());
    rc = c2c_typed<T>(pl, za, zb, batch, 1, s);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(take_real_kernel<T>, dim3(ew_grid(batch * n, pl->num_cu)), dim3(256), 0, s, zb, static_cast<T *>(out),
                       batch * (size_t)n);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

bool is_pow2(unsigned n) { return n && (n & (n - 1)) == 0; }

}  // namespace

namespace {

// H[k] = g[min(k, n-k)] / n, k < n: multiplying bins 0..n/2 of a real frame by g and dividing
// the backward transform by n is a circular convolution with a real, even spectrum
__global__ __launch_bounds__(256) void filter_spectrum_kernel(C<float> *__restrict__ H, const float *__restrict__ g, unsigned n)
{
    GRID_STRIDE(k, (size_t)n) H[k] = {g[k <= n - k ? k : n - k] / (float)n, 0.f};
}

// f32, power-of-two n: the spectrum above is real and even, so two real frames ride through one
// n-point complex transform as re + i*im and come back separated -- the fused overlap-save
// kernel with one tap's worth of overlap (none).  One pass over HBM instead of five.
int filter_fused_f32(dspmi_fft_plan *pl, dspmi::ols_launch_fn fused, const float *in, float *out, const float *gains,
                     size_t batch, hipStream_t s)
{
    const unsigned n = pl->n;
    int rc = ensure_work(pl, 2, (size_t)n * sizeof(C<float>));
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(filter_spectrum_kernel, dim3(ew_grid(n, pl->num_cu)), dim3(256), 0, s, static_cast<C<float> *>(pl->work[2]),
                       gains, n);
    DSPMI_HIP(hipGetLastError());
    dspmi::OlsLaunch o;
    o.hist = nullptr;
    o.H = pl->work[2];
    o.hist_stride = 0;
    o.nh = 1;
    // every pair of frames is one row of 2n samples; an odd last frame is a row of its own
    for (size_t done = 0; done < batch;) {
        const size_t left = batch - done;
        const size_t pairs = left / 2 > 0x40000000u ? 0x40000000u : left / 2;
        o.x = in + done * n;
        o.y = out + done * n;
        o.nx = pairs ? 2 * n : n;
        o.x_stride = o.y_stride = o.nx;
        o.nchan = pairs ? (unsigned)pairs : 1;
        rc = fused(o, pl->tw_n, pl->num_cu, s);
        if (rc != DSPMI_OK) return rc;
        done += pairs ? 2 * pairs : 1;
    }
    return DSPMI_OK;
}

template <typename T>
int filter_typed(dspmi_fft_plan *pl, const void *in, void *out, const void *gains, size_t batch, hipStream_t s)
{
    const unsigned n = pl->n, nbin = n / 2 + 1;
    if constexpr (sizeof(T) == sizeof(float)) {
        const char *env = getenv("DSPMI_FILTER_COMPOSED");  // A/B knob: r2c, gain, c2r, scale as separate kernels
        dspmi::ols_launch_fn fused = (pl->kern && n >= 16 && !(env && atoi(env))) ? dspmi::ols_find(n) : nullptr;
        if (fused)
            return filter_fused_f32(pl, fused, static_cast<const float *>(in), static_cast<float *>(out),
                                    static_cast<const float *>(gains), batch, s);
    }
    int rc = ensure_work(pl, 2, batch * (size_t)nbin * sizeof(C<T>));
    if (rc != DSPMI_OK) return rc;
    rc = r2c_typed<T>(pl, in, pl->work[2], batch, s);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(bin_gain_kernel<T>, dim3(ew_grid(batch * nbin, pl->num_cu)), dim3(256), 0, s,
                       static_cast<C<T> *>(pl->work[2]), static_cast<const T *>(gains), nbin, batch);
    DSPMI_HIP(hipGetLastError());
    rc = c2r_typed<T>(pl, pl->work[2], out, batch, s);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(scale_kernel<T>, dim3(ew_grid(batch * (size_t)n, pl->num_cu)), dim3(256), 0, s, static_cast<T *>(out), (T)n,
                       batch * (size_t)n);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

}  // namespace

extern "C" {

namespace {

// c2c_only: a plan that another plan keeps for its complex transforms (the convolution of an
// any-length transform, the half-length transform of a real one).  It is never asked for a real
// transform, so it needs neither a half-length plan of its own nor, when it is split or any-length
// itself, the n-entry table (which only the single-kernel transform and the real <-> half-complex
// steps read).
int create_plan(dspmi_fft_t *plan, unsigned n, int precision, int device, bool c2c_only)
{
    if (!plan || n == 0) return dspmi::fail(DSPMI_EINVAL, "fft_create: null plan pointer or n == 0");
    if (precision != DSPMI_F32 && precision != DSPMI_F64) return dspmi::fail(DSPMI_EINVAL, "fft_create: bad precision");
    // the twiddle table of a plan has n entries: 2^24 points (256 MiB in double) is where this stops
    constexpr unsigned kMaxN = 1u << 24;
    if (n > kMaxN) return dspmi::fail(DSPMI_ENOSYS, "fft_create: n = %u exceeds the longest supported transform (%u)", n, kMaxN);
    const dspmi::FftKernel *k = nullptr, *kh = nullptr, *kM = nullptr, *k1 = nullptr, *k2 = nullptr;
    unsigned M = 0, n1 = 0, n2 = 0;
    bool sub_M = false, sub_half = false;
    if (n > 1) {
        if (is_pow2(n)) {
            k = dspmi::fft_find(n, precision);
            if (!k) {
                // n = n1 * n2, n1 <= n2, each with a kernel of its own
                unsigned lg = 0;
                while ((1u << lg) < n) lg++;
                n1 = 1u << (lg / 2);
                n2 = n / n1;
                k1 = dspmi::fft_find(n1, precision);
                k2 = dspmi::fft_find(n2, precision);
                if (!k1 || !k2 || n1 < 32)
                    return dspmi::fail(DSPMI_ENOSYS, "fft_create: no kernels to split n = %u as %u x %u", n, n1, n2);
            }
        } else {
            // Bluestein: a convolution of M >= 2n-1 points
            if (n > kMaxN / 4) return dspmi::fail(DSPMI_ENOSYS, "fft_create: n = %u is not a power of two and too long", n);
            M = 1;
            while (M < 2 * n - 1) M <<= 1;
            kM = dspmi::fft_find(M, precision);
            sub_M = !kM;
        }
        if (n % 2 == 0 && n >= 4 && !c2c_only) {
            if (is_pow2(n / 2)) kh = dspmi::fft_find(n / 2, precision);
            sub_half = !kh;  // the n/2-point complex transform is a plan of its own (any-length or split)
        }
    }
    if (device < 0) DSPMI_HIP(hipGetDevice(&device));
    int prev = 0;
    DSPMI_HIP(hipGetDevice(&prev));
    DSPMI_HIP(hipSetDevice(device));
    dspmi_fft_plan *pl = new dspmi_fft_plan();
    pl->device = device;
    pl->precision = precision;
    pl->n = n;
    pl->esz = precision == DSPMI_F64 ? sizeof(C<double>) : sizeof(C<float>);
    pl->kern = k;
    pl->kern_half = kh;
    pl->kern_M = kM;
    pl->kern_n1 = k1;
    pl->kern_n2 = k2;
    pl->n1 = n1;
    pl->n2 = n2;
    if (kM) pl->blu = dspmi::bluestein_find(M, precision);
    pl->M = M;
    if (k) snprintf(pl->name, sizeof pl->name, "%s", k->name);
    else if (k1) snprintf(pl->name, sizeof pl->name, "fft_%s_n%u_as_%ux%u", precision == DSPMI_F64 ? "f64" : "f32", n, n1, n2);
    else if (kM && pl->blu) snprintf(pl->name, sizeof pl->name, "bluestein_n%u_m%u_one_kernel", n, M);
    else if (kM) snprintf(pl->name, sizeof pl->name, "bluestein_n%u_via_%s", n, kM->name);
    else if (M) snprintf(pl->name, sizeof pl->name, "bluestein_n%u_m%u_split", n, M);
    else snprintf(pl->name, sizeof pl->name, "fft_n1");
    hipDeviceProp_t prop;
    int rc = DSPMI_OK;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) rc = dspmi::fail(DSPMI_EIO, "fft_create: hipGetDeviceProperties failed");
    else pl->num_cu = prop.multiProcessorCount;
    if (rc == DSPMI_OK) {
        void *unused = nullptr;
        rc = noalloc_scratch(device, &unused);  // set-up time: the no-allocation entry point then only reads the table
    }
    if (rc == DSPMI_OK && (k || !c2c_only)) rc = upload_twiddles(&pl->tw_n, n, precision);
    if (rc == DSPMI_OK && kh) rc = upload_twiddles(&pl->tw_half, n / 2, precision);
    if (rc == DSPMI_OK && k1) {
        // W_n^e = hi[e >> shift] * lo[e & mask]
        unsigned lg = 0;
        while ((1u << lg) < n) lg++;
        pl->tw_shift = (lg + 1) / 2;
        rc = upload_twiddles(&pl->tw_n1, n1, precision);
        if (rc == DSPMI_OK) rc = upload_twiddles(&pl->tw_n2, n2, precision);
        if (lg > 20) {
            // three factors of 128 .. 256 points, the longest first in memory order (t3)
            const unsigned l1 = lg / 3, l2 = (lg - l1) / 2, l3 = lg - l1 - l2;
            pl->t1 = 1u << l1;
            pl->t2 = 1u << l2;
            pl->t3 = 1u << l3;
            if (rc == DSPMI_OK) rc = upload_twiddles(&pl->tw_t1, pl->t1, precision);
            if (rc == DSPMI_OK) rc = upload_twiddles(&pl->tw_t2, pl->t2, precision);
            if (rc == DSPMI_OK) rc = upload_twiddles(&pl->tw_t3, pl->t3, precision);
            snprintf(pl->name, sizeof pl->name, "fft_%s_n%u_as_%ux%ux%u", precision == DSPMI_F64 ? "f64" : "f32", n, pl->t1, pl->t2, pl->t3);
        }
        if (rc == DSPMI_OK)
            rc = precision == DSPMI_F64 ? upload<double>(&pl->tw_hi, unit_circle<double>(n >> pl->tw_shift, 1))
                                        : upload<float>(&pl->tw_hi, unit_circle<float>(n >> pl->tw_shift, 1));
        if (rc == DSPMI_OK)
            rc = precision == DSPMI_F64 ? upload<double>(&pl->tw_lo, unit_circle<double>(1u << pl->tw_shift, n >> pl->tw_shift))
                                        : upload<float>(&pl->tw_lo, unit_circle<float>(1u << pl->tw_shift, n >> pl->tw_shift));
    }
    if (rc == DSPMI_OK && sub_M) rc = create_plan(&pl->plan_M, M, precision, device, true);
    if (rc == DSPMI_OK && M) rc = precision == DSPMI_F64 ? setup_bluestein<double>(pl) : setup_bluestein<float>(pl);
    if (rc == DSPMI_OK && sub_half) rc = create_plan(&pl->half_plan, n / 2, precision, device, true);
    (void)hipSetDevice(prev);
    if (rc != DSPMI_OK) {
        dspmi_fft_destroy(pl);
        return rc;
    }
    *plan = pl;
    return DSPMI_OK;
}

}  // namespace

int dspmi_fft_create(dspmi_fft_t *plan, unsigned n, int precision, int device)
{
    return create_plan(plan, n, precision, device, false);
}

int dspmi_fft_destroy(dspmi_fft_t plan)
{
    if (!plan) return DSPMI_OK;
    if (plan->half_plan) (void)dspmi_fft_destroy(plan->half_plan);
    if (plan->plan_M) (void)dspmi_fft_destroy(plan->plan_M);
    void *bufs[] = {plan->tw_n, plan->tw_half, plan->tw_M, plan->chirp, plan->filt, plan->tw_n1, plan->tw_n2, plan->tw_t1, plan->tw_t2, plan->tw_t3, plan->tw_hi, plan->tw_lo,
                    plan->work[0], plan->work[1], plan->work[2], plan->work[3], plan->work[4]};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    delete plan;
    return DSPMI_OK;
}

int dspmi_fft_exec_c2c(dspmi_fft_t plan, const void *in, void *out, size_t batch, int sign, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c: null plan");
    if (int rc = dspmi::check_device(plan->device, "fft_exec_c2c")) return rc;
    if (sign != DSPMI_FORWARD && sign != DSPMI_BACKWARD) return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c: sign must be -1 or +1");
    if (batch == 0) return DSPMI_OK;
    if (!in || !out) return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c: null buffer");
    // frames are transformed whole (loaded, then stored), so in place is fine; a shifted overlap
    // would let one workgroup overwrite another's input
    if (in != out && ranges_overlap(in, batch * (size_t)plan->n * plan->esz, out, batch * (size_t)plan->n * plan->esz))
        return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c: in and out overlap without being the same buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (plan->n == 1) {
        if (in != out) DSPMI_HIP(hipMemcpyAsync(out, in, batch * plan->esz, hipMemcpyDeviceToDevice, s));
        return DSPMI_OK;
    }
    const int backward = sign == DSPMI_BACKWARD;
    return plan->precision == DSPMI_F64 ? c2c_typed<double>(plan, in, out, batch, backward, s)
                                        : c2c_typed<float>(plan, in, out, batch, backward, s);
}

int dspmi_fft_exec_c2c_noalloc(dspmi_fft_t plan, const void *in, void *out, size_t batch, int sign, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c_noalloc: null plan");
    if (int rc = dspmi::check_device(plan->device, "fft_exec_c2c_noalloc")) return rc;
    if (sign != DSPMI_FORWARD && sign != DSPMI_BACKWARD)
        return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c_noalloc: sign must be -1 or +1");
    if (batch == 0) return DSPMI_OK;
    if (!in || !out) return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c_noalloc: null buffer");
    if (in != out && ranges_overlap(in, batch * (size_t)plan->n * plan->esz, out, batch * (size_t)plan->n * plan->esz))
        return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2c_noalloc: in and out overlap without being the same buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (plan->n == 1) {
        if (in != out) DSPMI_HIP(hipMemcpyAsync(out, in, batch * plan->esz, hipMemcpyDeviceToDevice, s));
        return DSPMI_OK;
    }
    void *work = nullptr;
    if (int rc = noalloc_scratch(plan->device, &work)) return rc;
    const int backward = sign == DSPMI_BACKWARD;
    return plan->precision == DSPMI_F64 ? c2c_typed<double>(plan, in, out, batch, backward, s, work, kNoAllocScratchBytes)
                                        : c2c_typed<float>(plan, in, out, batch, backward, s, work, kNoAllocScratchBytes);
}

int dspmi_fft_exec_r2c(dspmi_fft_t plan, const void *in, void *out, size_t batch, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c: null plan");
    if (int rc = dspmi::check_device(plan->device, "fft_exec_r2c")) return rc;
    if (batch == 0) return DSPMI_OK;
    if (!in || !out) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c: null buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return plan->precision == DSPMI_F64 ? r2c_typed<double>(plan, in, out, batch, s) : r2c_typed<float>(plan, in, out, batch, s);
}

int dspmi_fft_exec_c2r(dspmi_fft_t plan, const void *in, void *out, size_t batch, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2r: null plan");
    if (int rc = dspmi::check_device(plan->device, "fft_exec_c2r")) return rc;
    if (batch == 0) return DSPMI_OK;
    if (!in || !out) return dspmi::fail(DSPMI_EINVAL, "fft_exec_c2r: null buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return plan->precision == DSPMI_F64 ? c2r_typed<double>(plan, in, out, batch, s) : c2r_typed<float>(plan, in, out, batch, s);
}

int dspmi_fft_exec_r2c_spectrum(dspmi_fft_t plan, const void *in, float *ps, size_t batch, unsigned nbin, int mode,
                                float scale, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum: null plan");
    if (int rc = dspmi::check_device(plan->device, "fft_exec_r2c_spectrum")) return rc;
    if (plan->precision != DSPMI_F32) return dspmi::fail(DSPMI_ENOSYS, "fft_exec_r2c_spectrum: f32 plans only");
    if (batch == 0) return DSPMI_OK;
    if (!in || !ps) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum: null buffer");
    if (mode < 0 || mode > 2) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum: mode must be 0, 1 or 2");
    const unsigned n = plan->n;
    if (nbin == 0 || nbin > n / 2 + 1) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum: nbin must be 1..n/2+1");
    if (ranges_overlap(in, batch * (size_t)n * sizeof(float), ps, batch * (size_t)nbin * sizeof(float)))
        return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum: the spectrum buffer overlaps the input");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (fused_real(plan) && nbin >= n / 2)
        return plan->kern_half->fn_r2c(in, ps, plan->tw_half, plan->tw_n, batch, 2, nbin, mode, scale, plan->num_cu, s);
    // any other length: bins to scratch, then the stand-alone spectrum kernel
    int rc = ensure_work(plan, 2, batch * (size_t)(n / 2 + 1) * sizeof(C<float>));
    if (rc != DSPMI_OK) return rc;
    rc = r2c_typed<float>(plan, in, plan->work[2], batch, s);
    if (rc != DSPMI_OK) return rc;
    return dspmi_power_spectrum_f32(ps, plan->work[2], nbin, n / 2 + 1, batch, mode, scale, stream);
}

int dspmi_fir_spectrum_f32(dspmi_fir_t fir, dspmi_fft_t fft, float *ps, unsigned nbin, int mode, float scale, const float *x,
                           size_t x_stride, unsigned nx, unsigned nchan, const float *hist, size_t hist_stride, void *stream)
{
    if (!fir || !fft) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: null plan");
    if (int rc = dspmi::check_device(fft->device, "fir_spectrum")) return rc;
    dspmi::FirSpectrumLaunch a{};
    int fir_device = -1;
    if (int rc = dspmi::fir_half_images(fir, &a.tap_images, &a.tap_unscale, &a.slices, &a.nh, &fir_device, &a.taps)) return rc;
    if (fir_device != fft->device) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: the two plans belong to different devices");
    if (fft->precision != DSPMI_F32 || fft->n != 4096 || !fused_real(fft))
        return dspmi::fail(DSPMI_ENOSYS, "fir_spectrum: needs an f32 transform plan of 4096 points");
    if (nchan == 0 || nx == 0) return DSPMI_OK;
    if (nx % 4096) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: rows must hold whole frames of 4096 samples");
    if (!x || !ps) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: null buffer");
    if (x_stride < nx) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: stride smaller than the row length");
    if (hist && hist_stride + 1 < a.nh) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: hist_stride < nh-1");
    if (mode < 0 || mode > 2) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: mode must be 0, 1 or 2");
    if (nbin != 2048 && nbin != 2049) return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: nbin must be 2048 or 2049");
    const size_t frames = (size_t)nchan * (nx / 4096);
    if (ranges_overlap(x, ((size_t)(nchan - 1) * x_stride + nx) * sizeof(float), ps, frames * nbin * sizeof(float)))
        return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: the spectrum buffer overlaps the input");
    a.x = x;
    a.hist = hist;
    a.x_stride = x_stride;
    a.hist_stride = hist_stride;
    a.nx = nx;
    a.nchan = nchan;
    a.tw_half = fft->tw_half;
    a.tw_full = fft->tw_n;
    a.ps = ps;
    a.nbin = nbin;
    a.mode = mode;
    a.scale = scale;
    return dspmi::fir_spectrum_launch(a, static_cast<hipStream_t>(stream));
}

size_t dspmi_tone_fir_spectrum_work_bytes(unsigned nx, unsigned nchan, unsigned ntone)
{
    if (ntone == 0 || ntone > 4 || nx % 4096) return 0;
    return dspmi::tone_chain_work_bytes(nx, nchan, ntone <= 2 ? 2 : 4);
}

int dspmi_tone_fir_spectrum_f32(dspmi_fir_t fir, dspmi_fft_t fft, float *ps, unsigned nbin, int mode, float scale, unsigned nx,
                                unsigned nchan, unsigned ntone, double *w, const double *dw, const double *amp, const double *phi,
                                int use_sin, const float *hist, size_t hist_stride, float *hist_out, size_t hist_out_stride,
                                void *work, size_t work_bytes, void *stream)
{
    if (!fir || !fft) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: null plan");
    if (int rc = dspmi::check_device(fft->device, "tone_fir_spectrum")) return rc;
    dspmi::FirSpectrumLaunch a{};
    int fir_device = -1;
    if (int rc = dspmi::fir_half_images(fir, &a.tap_images, &a.tap_unscale, &a.slices, &a.nh, &fir_device, &a.taps)) return rc;
    if (fir_device != fft->device) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: the two plans belong to different devices");
    if (fft->precision != DSPMI_F32 || fft->n != 4096 || !fused_real(fft))
        return dspmi::fail(DSPMI_ENOSYS, "tone_fir_spectrum: needs an f32 transform plan of 4096 points");
    if (ntone == 0 || ntone > 4)
        return dspmi::fail(DSPMI_ENOSYS, "tone_fir_spectrum: 1..4 tones per channel (got %u); generate, then dspmi_fir_spectrum_f32", ntone);
    if (nchan == 0 || nx == 0) return DSPMI_OK;
    if (nx % 4096) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: rows must hold whole frames of 4096 samples");
    if (!ps || !w || !dw || !work || (!use_sin && (!amp || !phi))) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: null buffer");
    if (hist && hist_stride + 1 < a.nh) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: hist_stride < nh-1");
    if (hist_out && hist_out_stride + 1 < a.nh) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: hist_out_stride < nh-1");
    if (mode < 0 || mode > 2) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: mode must be 0, 1 or 2");
    if (nbin != 2048 && nbin != 2049) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: nbin must be 2048 or 2049");
    if (reinterpret_cast<uintptr_t>(work) & 255) return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: work space must be 256-byte aligned");
    const int slots = ntone <= 2 ? 2 : 4;
    const size_t need = dspmi::tone_chain_work_bytes(nx, nchan, slots);
    if (work_bytes < need)
        return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: work space of %zu bytes, %zu needed", work_bytes, need);
    const size_t frames = (size_t)nchan * (nx / 4096);
    if (ranges_overlap(work, need, ps, frames * nbin * sizeof(float)))
        return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: the spectrum buffer overlaps the work space");
    if (hist && hist_out && a.nh > 1 &&
        ranges_overlap(hist, ((size_t)(nchan - 1) * hist_stride + a.nh - 1) * sizeof(float), hist_out,
                       ((size_t)(nchan - 1) * hist_out_stride + a.nh - 1) * sizeof(float)))
        return dspmi::fail(DSPMI_EINVAL, "tone_fir_spectrum: hist_out overlaps hist (the rows' first and last tiles run concurrently)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dspmi::ToneChainTables tab = dspmi::tone_chain_carve(work, nx, nchan, slots);
    const unsigned lead = 32u * (unsigned)a.slices - 16u;   // window positions ahead of a tile's first output
    if (int rc = dspmi::tone_chain_prepare(tab, nx, nchan, ntone, slots, lead, w, dw, amp, phi, use_sin, s)) return rc;
    a.x = tab.spill;
    a.x_stride = nx;
    a.hist = hist;
    a.hist_stride = hist_stride;
    a.nx = nx;
    a.nchan = nchan;
    a.tw_half = fft->tw_half;
    a.tw_full = fft->tw_n;
    a.ps = ps;
    a.nbin = nbin;
    a.mode = mode;
    a.scale = scale;
    a.tone_slots = slots;
    a.lane_tab = tab.lane_tab;
    a.tile_tab = tab.tile_tab;
    a.hard = tab.hard;
    a.hist_out = hist_out;
    a.hist_out_stride = hist_out_stride;
    a.w = w;
    a.dw = dw;
    a.nseq = nchan * ntone;
    return dspmi::fir_spectrum_launch(a, s);
}

int dspmi_fft_exec_r2c_spectrum_s16x2(dspmi_fft_t plan, const int16_t *in, float *ps, size_t batch, unsigned nbin, int mode,
                                      float scale, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum_s16x2: null plan");
    if (int rc = dspmi::check_device(plan->device, "fft_exec_r2c_spectrum_s16x2")) return rc;
    if (plan->precision != DSPMI_F32) return dspmi::fail(DSPMI_ENOSYS, "fft_exec_r2c_spectrum_s16x2: f32 plans only");
    if (batch == 0) return DSPMI_OK;
    if (!in || !ps) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum_s16x2: null buffer");
    if (reinterpret_cast<uintptr_t>(in) & 7)
        return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum_s16x2: input must be 8-byte aligned");
    if (mode < 0 || mode > 2) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum_s16x2: mode must be 0, 1 or 2");
    const unsigned n = plan->n;
    if (nbin == 0 || nbin > n / 2 + 1) return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum_s16x2: nbin must be 1..n/2+1");
    if (ranges_overlap(in, batch * (size_t)n * 2 * sizeof(int16_t), ps, batch * (size_t)nbin * sizeof(float)))
        return dspmi::fail(DSPMI_EINVAL, "fft_exec_r2c_spectrum_s16x2: the spectrum buffer overlaps the input");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (fused_real(plan) && nbin >= n / 2)
        return plan->kern_half->fn_r2c(in, ps, plan->tw_half, plan->tw_n, batch, 2 | dspmi::kR2cSourceS16x2, nbin, mode, scale,
                                       plan->num_cu, s);
    // any other length: convert to mono floats first
    const size_t total = batch * (size_t)n;
    int rc = ensure_work(plan, 3, total * sizeof(float));
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(s16x2_to_mono_kernel, dim3(ew_grid(total, plan->num_cu)), dim3(256), 0, s,
                       static_cast<float *>(plan->work[3]), reinterpret_cast<const int *>(in), total);
    DSPMI_HIP(hipGetLastError());
    return dspmi_fft_exec_r2c_spectrum(plan, plan->work[3], ps, batch, nbin, mode, scale, stream);
}

int dspmi_fft_exec_filter(dspmi_fft_t plan, const void *in, void *out, const void *gains, size_t batch, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fft_exec_filter: null plan");
    if (int rc = dspmi::check_device(plan->device, "fft_exec_filter")) return rc;
    if (batch == 0) return DSPMI_OK;
    if (!in || !out || !gains) return dspmi::fail(DSPMI_EINVAL, "fft_exec_filter: null buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return plan->precision == DSPMI_F64 ? filter_typed<double>(plan, in, out, gains, batch, s)
                                        : filter_typed<float>(plan, in, out, gains, batch, s);
}

const char *dspmi_fft_variant(dspmi_fft_t plan)
{
    return plan ? plan->name : "";
}

}  // extern "C"

// Element-wise helpers used by the host wrappers.
namespace dspmi {

int ps_abs_f64(const void *bins, double *out, unsigned nout, double scale, size_t batch, unsigned in_stride, hipStream_t s)
{
    hipLaunchKernelGGL(ps_abs_kernel<double>, dim3(ew_grid(batch * nout, 256)), dim3(256), 0, s,
                       static_cast<const C<double> *>(bins), out, nout, scale, batch, in_stride);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

}  // namespace dspmi
