// Overlap-save fast convolution for long filters (f32).
//
// The direct kernel costs 2*nh flops per output; beyond a few hundred taps it is cheaper
// to filter in the frequency domain, the scheme the reference's notes name
// (doc/notes:81-82) and util/filter/filter.c:73-105 applies to a single block:
// forward FFT, multiply by the filter's spectrum, backward FFT, divide by n.
//
// The signal is cut into blocks of N samples that overlap by nh-1; block b yields the
// L = N-(nh-1) outputs that do not wrap around.  Because the taps are real, two real
// blocks ride through one complex transform as re + i*im and come back separated, so no
// real<->complex split pass is needed.  H = FFT_N(h)/N is computed once per plan.
//
// The measured path is the fused kernel of fft_stockham.hip (ols_fused_kernel: the block is
// loaded, transformed, multiplied, transformed back and stored without leaving the CU); the
// five-kernel composition below (gather -> batched FFT -> multiply -> batched FFT -> scatter)
// is the same arithmetic through HBM, kept as the A/B reference (DSPMI_OLS_COMPOSED=1).
#include "common.h"
#include "fft_internal.h"

#include <cmath>
#include <complex>
#include <vector>

namespace {

struct alignas(8) C32 {
    float re, im;
};

#define GRID_STRIDE(i, total) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (size_t)gridDim.x * blockDim.x)

struct OlsShape {
    unsigned N, L, nh;       // transform length, valid outputs per block, taps
    unsigned nblk;           // blocks per channel = ceil(nx / L)
    unsigned npair;          // complex transforms per channel = ceil(nblk / 2)
    unsigned nx, nchan;
    size_t x_stride, y_stride, hist_stride;
};

__device__ __forceinline__ float ols_sample(const float *__restrict__ xrow, const float *__restrict__ hrow, long s, long nx,
                                            long nh)
{
    if (s >= 0) return s < nx ? xrow[s] : 0.f;
    return (hrow && s >= 1 - nh) ? hrow[nh - 1 + s] : 0.f;
}

// work[c][p][m] = x[c][(2p)L + m - (nh-1)] + i * x[c][(2p+1)L + m - (nh-1)]
__global__ __launch_bounds__(256) void ols_gather_kernel(C32 *__restrict__ work, const float *__restrict__ x,
                                                         const float *__restrict__ hist, const OlsShape sh)
{
    const size_t per_chan = (size_t)sh.npair * sh.N;
    GRID_STRIDE(i, per_chan * sh.nchan) {
        const size_t c = i / per_chan;
        const size_t r = i - c * per_chan;
        const unsigned p = (unsigned)(r / sh.N), m = (unsigned)(r - (size_t)p * sh.N);
        const float *xrow = x + c * sh.x_stride;
        const float *hrow = hist ? hist + c * sh.hist_stride : nullptr;
        const long s0 = (long)(2 * p) * sh.L + m - (long)(sh.nh - 1);
        C32 v;
        v.re = ols_sample(xrow, hrow, s0, sh.nx, sh.nh);
        v.im = (2 * p + 1 < sh.nblk) ? ols_sample(xrow, hrow, s0 + sh.L, sh.nx, sh.nh) : 0.f;
        work[i] = v;
    }
}

__global__ __launch_bounds__(256) void ols_multiply_kernel(C32 *__restrict__ work, const C32 *__restrict__ H, unsigned N,
                                                           size_t total)
{
    GRID_STRIDE(i, total) {
        const C32 a = work[i], b = H[i & (N - 1)];
        work[i] = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    }
}

// y[c][bL + j] = part(work[c][b/2][nh-1+j]), j < L
__global__ __launch_bounds__(256) void ols_scatter_kernel(float *__restrict__ y, const C32 *__restrict__ work, const OlsShape sh)
{
    const size_t per_chan = (size_t)sh.nblk * sh.L;
    GRID_STRIDE(i, per_chan * sh.nchan) {
        const size_t c = i / per_chan;
        const size_t o = i - c * per_chan;
        if (o >= sh.nx) continue;
        const unsigned b = (unsigned)(o / sh.L), j = (unsigned)(o - (size_t)b * sh.L);
        const C32 v = work[(c * sh.npair + b / 2) * sh.N + (sh.nh - 1) + j];
        y[c * sh.y_stride + o] = (b & 1) ? v.im : v.re;
    }
}

// In-place radix-2 transform on the host, in double (forward, unnormalised; n a power of two):
// the taps' spectrum is computed once per plan, here, and rounded to f32 once.
void host_fft(std::vector<std::complex<double>> &z)
{
    const size_t n = z.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(z[i], z[j]);
    }
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (size_t len = 2; len <= n; len <<= 1) {
        std::vector<std::complex<double>> w(len / 2);
        for (size_t k = 0; k < len / 2; k++) {
            const long double a = -two_pi * (long double)k / (long double)len;
            w[k] = {(double)cosl(a), (double)sinl(a)};
        }
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                const std::complex<double> u = z[i + k], t = z[i + k + len / 2] * w[k];
                z[i + k] = u + t;
                z[i + k + len / 2] = u - t;
            }
    }
}

// DSPMI_OLS_COMPOSED=1 (A/B knob, read at plan creation): the five-kernel composition
bool composed_now()
{
    const char *env = getenv("DSPMI_OLS_COMPOSED");
    return env && atoi(env);
}

inline unsigned ew_grid(size_t total, int num_cu)
{
    size_t g = (total + 255) / 256;
    const size_t cap = (size_t)num_cu * 16;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

}  // namespace

struct dspmi_ols_plan {
    int device = 0, num_cu = 256;
    unsigned nh = 0, N = 0, L = 0;
    const dspmi::FftKernel *kern = nullptr;
    dspmi::ols_launch_fn fused = nullptr;  // one-kernel form (nullptr: composed path below)
    bool real_blocks = false;              // fused = the kernel that runs one real block per N/2-point transform
    void *tw = nullptr;    // exp(-2 pi i k / N)
    void *H = nullptr;     // FFT_N(taps) / N
    void *work = nullptr;  // [nchan][npair][N] complex
    size_t work_bytes = 0;
};

extern "C" {

int dspmi_ols_create_f32(dspmi_ols_t *plan, const float *taps, unsigned nh, unsigned fft_n, int device)
{
    if (!plan || !taps || nh == 0) return dspmi::fail(DSPMI_EINVAL, "ols_create: null argument or nh == 0");
    unsigned N = fft_n;
    if (N == 0) {
        // default: the smallest transform that keeps >= 3/4 of each block; beyond 2048 taps the
        // 32768-point kernel (it costs the same per point as the 16384-point one and keeps more)
        N = 16;
        while (N < 4 * nh && N < 16384) N <<= 1;
        if (nh > 2048 && !composed_now()) N = 32768;
    }
    const bool composed = composed_now();
    if ((N & (N - 1)) || N < 2 * nh || !(dspmi::fft_find(N, DSPMI_F32) || (!composed && dspmi::ols_find(N))))
        return dspmi::fail(DSPMI_ENOSYS, "ols_create: transform length %u unusable for %u taps (power of two >= 2*nh, <= 32768)",
                           N, nh);
    if (device < 0) DSPMI_HIP(hipGetDevice(&device));
    int prev = 0;
    DSPMI_HIP(hipGetDevice(&prev));
    DSPMI_HIP(hipSetDevice(device));
    dspmi_ols_plan *pl = new dspmi_ols_plan();
    pl->device = device;
    pl->nh = nh;
    pl->N = N;
    pl->L = N - (nh - 1);
    pl->kern = dspmi::fft_find(N, DSPMI_F32);  // the composed path's transform; none beyond 16384 points
    if (!composed) pl->fused = dspmi::ols_find(N);
    {
        // 32768-sample blocks: one real block per 16384-point transform, two workgroups per CU
        // (DSPMI_OLS_REAL=0 keeps the kernel that runs two blocks through a 32768-point transform)
        const char *env = getenv("DSPMI_OLS_REAL");
        if (!composed && !(env && !atoi(env)) && dspmi::ols_find_real(N)) {
            pl->fused = dspmi::ols_find_real(N);
            pl->real_blocks = true;
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) pl->num_cu = prop.multiProcessorCount;

    // Blocks beyond 16384 points run the kernel that splits a block into N/1024 rows of 1024
    // points (fft_stockham.hip, ols_wave_kernel).  Its tables: exp(-2 pi i k / 1024), k < 1024,
    // followed by the stage factors exp(-2 pi i k1 n2 / N) as [k1][n2]; H is stored row by row,
    // H[k1 + rows * k2] at [k1][k2].
    const unsigned rows = (N > 16384 && !composed) ? N / 1024 : 0;  // the five-kernel composition keeps the plain layout
    const long double two_pi = 6.283185307179586476925286766559005768L;
    std::vector<C32> tw(rows ? 1024 + (size_t)N : N), h(N, C32{0.f, 0.f});
    if (pl->real_blocks) {
        // ols_real_kernel (fft_stockham.hip): N2 = N/2 complex points as 32 rows of 512
        const unsigned N2 = N / 2, NS = 512, K1 = N2 / NS;
        tw.assign((size_t)NS + N2 + K1 * 64, C32{0.f, 0.f});
        auto unit = [&](unsigned long long e, unsigned n) {
            const long double a = -two_pi * (long double)(e % n) / (long double)n;
            return C32{(float)cosl(a), (float)sinl(a)};
        };
        for (unsigned k = 0; k < NS; k++) tw[k] = unit(k, NS);
        for (unsigned k1 = 0; k1 < K1; k1++)
            for (unsigned n2 = 0; n2 < NS; n2++) tw[NS + (size_t)k1 * NS + n2] = unit((unsigned long long)k1 * n2, N2);
        for (unsigned k1 = 0; k1 < K1; k1++)
            for (unsigned l = 0; l < 64; l++) tw[(size_t)NS + N2 + k1 * 64 + l] = unit(k1 + K1 * l, N2);
        // {Hev / N2, (i/2) Hod / N2} per bin: the transforms of the even and the odd taps, row by row
        std::vector<std::complex<double>> he(N2, {0.0, 0.0}), ho(N2, {0.0, 0.0});
        for (unsigned j = 0; j < nh; j++) (j & 1 ? ho : he)[j / 2] = {(double)taps[j], 0.0};
        host_fft(he);
        host_fft(ho);
        for (unsigned k = 0; k < N2; k++) {
            const size_t at = (size_t)(k % K1) * NS + k / K1;
            h[2 * at] = {(float)(he[k].real() / N2), (float)(he[k].imag() / N2)};
            h[2 * at + 1] = {(float)(-0.5 * ho[k].imag() / N2), (float)(0.5 * ho[k].real() / N2)};
        }
    } else if (rows) {
        for (unsigned k = 0; k < 1024; k++) {
            const long double a = -two_pi * (long double)k / 1024.0L;
            tw[k] = {(float)cosl(a), (float)sinl(a)};
        }
        for (unsigned k1 = 0; k1 < rows; k1++)
            for (unsigned n2 = 0; n2 < 1024; n2++) {
                const long double a = -two_pi * (long double)(((unsigned long long)k1 * n2) % N) / (long double)N;
                tw[1024 + (size_t)k1 * 1024 + n2] = {(float)cosl(a), (float)sinl(a)};
            }
    } else {
        for (unsigned k = 0; k < N; k++) {
            const long double a = -two_pi * (long double)k / (long double)N;
            tw[k] = {(float)cosl(a), (float)sinl(a)};
        }
    }
    if (!pl->real_blocks) {
        // H = FFT_N(taps) / N
        std::vector<std::complex<double>> hz(N, {0.0, 0.0});
        for (unsigned j = 0; j < nh; j++) hz[j] = {(double)taps[j], 0.0};
        host_fft(hz);
        for (unsigned k = 0; k < N; k++) {
            const size_t at = rows ? (size_t)(k % rows) * 1024 + k / rows : k;
            h[at] = {(float)(hz[k].real() / N), (float)(hz[k].imag() / N)};
        }
    }
    hipError_t e = hipMalloc(&pl->tw, tw.size() * sizeof(C32));
    if (e == hipSuccess) e = hipMalloc(&pl->H, N * sizeof(C32));
    if (e == hipSuccess) e = hipMemcpy(pl->tw, tw.data(), tw.size() * sizeof(C32), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->H, h.data(), N * sizeof(C32), hipMemcpyHostToDevice);
    int rc = DSPMI_OK;
    if (e != hipSuccess) rc = dspmi::fail(DSPMI_EIO, "ols_create: %s", hipGetErrorString(e));
    (void)hipSetDevice(prev);
    if (rc != DSPMI_OK) {
        dspmi_ols_destroy(pl);
        return rc;
    }
    *plan = pl;
    return DSPMI_OK;
}

int dspmi_ols_destroy(dspmi_ols_t plan)
{
    if (!plan) return DSPMI_OK;
    void *bufs[] = {plan->tw, plan->H, plan->work};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    delete plan;
    return DSPMI_OK;
}

unsigned dspmi_ols_fft_length(dspmi_ols_t plan) { return plan ? plan->N : 0; }

int dspmi_ols_exec_f32(dspmi_ols_t plan, float *y, size_t y_stride, const float *x, size_t x_stride, unsigned nx,
                       unsigned nchan, const float *hist, size_t hist_stride, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "ols_exec: null plan");
    if (int rc = dspmi::check_device(plan->device, "ols_exec")) return rc;
    if (nx == 0 || nchan == 0) return DSPMI_OK;
    if (!x || !y) return dspmi::fail(DSPMI_EINVAL, "ols_exec: null buffer");
    if (x_stride < nx || y_stride < nx) return dspmi::fail(DSPMI_EINVAL, "ols_exec: stride smaller than the row length");
    if (hist && hist_stride + 1 < plan->nh) return dspmi::fail(DSPMI_EINVAL, "ols_exec: hist_stride < nh-1");
    if (plan->nh > 1 && dspmi::rows_overlap(x, x_stride, nx, y, y_stride, nx, nchan, sizeof(float)))
        return dspmi::fail(DSPMI_EINVAL, "ols_exec: input and output rows overlap (filtering in place is not supported)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (plan->fused) {
        dspmi::OlsLaunch o;
        o.x = x; o.hist = hist; o.y = y; o.H = plan->H;
        o.x_stride = x_stride; o.y_stride = y_stride; o.hist_stride = hist_stride;
        o.nx = nx; o.nh = plan->nh; o.nchan = nchan;
        return plan->fused(o, plan->tw, plan->num_cu, s);
    }
    OlsShape sh;
    sh.N = plan->N; sh.L = plan->L; sh.nh = plan->nh;
    sh.nblk = (unsigned)(((unsigned long long)nx + sh.L - 1) / sh.L);
    sh.npair = (sh.nblk + 1) / 2;
    sh.nx = nx; sh.nchan = nchan;
    sh.x_stride = x_stride; sh.y_stride = y_stride; sh.hist_stride = hist_stride;
    const size_t frames = (size_t)nchan * sh.npair;
    const size_t bytes = frames * sh.N * sizeof(C32);
    if (bytes > plan->work_bytes) {
        if (plan->work) DSPMI_HIP(hipFree(plan->work));
        plan->work = nullptr;
        plan->work_bytes = 0;
        DSPMI_HIP(hipMalloc(&plan->work, bytes));
        plan->work_bytes = bytes;
    }
    C32 *w = static_cast<C32 *>(plan->work);
    const size_t total = frames * sh.N;
    hipLaunchKernelGGL(ols_gather_kernel, dim3(ew_grid(total, plan->num_cu)), dim3(256), 0, s, w, x, hist, sh);
    DSPMI_HIP(hipGetLastError());
    const bool big = bytes >= ((size_t)256 << 20);
    int rc = (big ? plan->kern->fn_stream : plan->kern->fn)(w, w, plan->tw, frames, 0, plan->num_cu, s);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(ols_multiply_kernel, dim3(ew_grid(total, plan->num_cu)), dim3(256), 0, s, w,
                       static_cast<const C32 *>(plan->H), sh.N, total);
    DSPMI_HIP(hipGetLastError());
    rc = (big ? plan->kern->fn_stream : plan->kern->fn)(w, w, plan->tw, frames, 1, plan->num_cu, s);
    if (rc != DSPMI_OK) return rc;
    hipLaunchKernelGGL(ols_scatter_kernel, dim3(ew_grid((size_t)nchan * sh.nblk * sh.L, plan->num_cu)), dim3(256), 0, s, y, w,
                       sh);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

}  // extern "C"
