// Host drop-in wrappers: the reference's C signatures over the device kernels.
//
// Each call stages its host double buffers to the device, runs the double
// precision kernels and blocks until the result is back in host memory -- the
// synchronous, caller-owned-buffer behaviour of the reference's static
// routines.  Any device failure prints the failing call and abort()s: these
// routines return void in the reference, and there is no CPU fallback.
#include "common.h"

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

int dspmi_fir_exec_f64_desc(dspmi_fir_t plan, double *y, size_t y_stride, unsigned nxx, const double *x,
                            size_t x_stride, unsigned nx, unsigned nchan, unsigned flags, void *stream);

namespace dspmi {
int ps_abs_f64(const void *bins, double *out, unsigned nout, double scale, size_t batch, unsigned in_stride,
               hipStream_t s);
int power_spectrum_f64(double *ps, const double *bins, unsigned nbin, unsigned in_stride, size_t batch, int mode, double scale,
                       hipStream_t s);
int phase_f64(double *phi, const double *bins, unsigned nbin, unsigned in_stride, size_t batch, hipStream_t s);
}  // namespace dspmi

namespace {

// RAII device buffer for the wrapper path.
struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t bytes) { DSPMI_HIP_OR_DIE(hipMalloc(&p, bytes ? bytes : 8)); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <typename T> T *as() { return static_cast<T *>(p); }
};

enum Order { ASCENDING, DESCENDING };

void run_fir(double *y, unsigned nxx, const double *x, unsigned nx, const double *h, unsigned nh, unsigned flags,
             Order order)
{
    if (nxx == 0) return;
    if (nh == 0) {  // an empty tap set sums nothing
        std::memset(y, 0, (size_t)nxx * sizeof(double));
        return;
    }
    dspmi_fir_t plan = nullptr;
    DSPMI_OR_DIE(dspmi_fir_create_f64(&plan, h, nh, -1));
    DevBuf dx((size_t)nx * sizeof(double)), dy((size_t)nxx * sizeof(double));
    if (nx) DSPMI_HIP_OR_DIE(hipMemcpy(dx.p, x, (size_t)nx * sizeof(double), hipMemcpyHostToDevice));
    if (order == DESCENDING)
        DSPMI_OR_DIE(dspmi_fir_exec_f64_desc(plan, dy.as<double>(), nxx, nxx, dx.as<double>(), nx, nx, 1, flags, nullptr));
    else
        DSPMI_OR_DIE(dspmi_fir_exec_f64(plan, dy.as<double>(), nxx, nxx, dx.as<double>(), nx, nx, 1, nullptr, 0, flags,
                                        nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(y, dy.p, (size_t)nxx * sizeof(double), hipMemcpyDeviceToHost));
    DSPMI_OR_DIE(dspmi_fir_destroy(plan));
}

std::mutex g_fir_mutex;
std::vector<double> g_fir_taps = {  // alsa/main.c:389-407
    -0.131625603115805, 0.103820148451119, 0.096541374482813, 0.102280926834003,
    0.113689768997426,  0.125850624234278, 0.135740574367069, 0.141133806318268,
    0.141133806318268,  0.135740574367069, 0.125850624234278, 0.113689768997426,
    0.102280926834003,  0.096541374482813, 0.103820148451119, -0.131625603115805};

void run_c2c(double (*in)[2], unsigned nx, double (*out)[2], int sign)
{
    if (nx == 0) return;
    dspmi_fft_t plan = nullptr;
    DSPMI_OR_DIE(dspmi_fft_create(&plan, nx, DSPMI_F64, -1));
    const size_t bytes = (size_t)nx * 2 * sizeof(double);
    DevBuf din(bytes), dout(bytes);
    DSPMI_HIP_OR_DIE(hipMemcpy(din.p, in, bytes, hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_c2c(plan, din.p, dout.p, 1, sign, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    DSPMI_OR_DIE(dspmi_fft_destroy(plan));
}

}  // namespace

extern "C" {

void dsp_conv0(const double *x, unsigned int nx, const double *h, unsigned int nh, double *y)
{
    // conv0 adds x[i]*h[k-i] into y[k] for ascending i, i.e. descending tap index
    run_fir(y, nx, x, nx, h, nh, 0, DESCENDING);
}

void dsp_conv1(const double *x, unsigned int nx, const double *h, unsigned int nh, double *y)
{
    // min(i, nh) taps: the j == i term, the only one that reads x[0], never enters
    run_fir(y, nx, x, nx, h, nh, DSPMI_FIR_DROP_FIRST, ASCENDING);
}

void dsp_convolve(double *xx, unsigned int nxx, const double *x, unsigned int nx, const double *h, unsigned int nh)
{
    run_fir(xx, nxx, x, nx, h, nh, 0, ASCENDING);
}

void dsp_convolve_alloc(double **xx, unsigned int *nxx, const double *x, unsigned int nx, const double *h,
                        unsigned int nh)
{
    *nxx = nx + nh;
    *xx = static_cast<double *>(malloc((size_t)*nxx * sizeof(double)));  // the caller free()s it
    if (!*xx) {
        fprintf(stderr, "libdspmi: dsp_convolve_alloc: out of host memory\n");
        abort();
    }
    run_fir(*xx, *nxx, x, nx, h, nh, 0, ASCENDING);
}

void dsp_fir_set_taps(const double *h, unsigned int nh)
{
    std::lock_guard<std::mutex> lock(g_fir_mutex);
    g_fir_taps.assign(h, h + nh);
}

void dsp_fir(double **xx, unsigned int *nxx, const double *x, unsigned int nx)
{
    std::vector<double> taps;
    {
        std::lock_guard<std::mutex> lock(g_fir_mutex);
        taps = g_fir_taps;
    }
    dsp_convolve_alloc(xx, nxx, x, nx, taps.data(), (unsigned)taps.size());
}

void dsp_dft(double (*in)[2], unsigned int nx, double (*out)[2]) { run_c2c(in, nx, out, DSPMI_FORWARD); }

void dsp_idft(double (*in)[2], unsigned int nx, double (*out)[2]) { run_c2c(in, nx, out, DSPMI_BACKWARD); }

void dsp_fft_r2c(double *xx, const double *x, unsigned int n)
{
    if (n == 0) return;
    dspmi_fft_t plan = nullptr;
    DSPMI_OR_DIE(dspmi_fft_create(&plan, n, DSPMI_F64, -1));
    const unsigned nbin = n / 2 + 1;
    DevBuf din((size_t)n * sizeof(double)), dout((size_t)nbin * 2 * sizeof(double));
    DSPMI_HIP_OR_DIE(hipMemcpy(din.p, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_r2c(plan, din.p, dout.p, 1, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(xx, dout.p, (size_t)nbin * 2 * sizeof(double), hipMemcpyDeviceToHost));
    DSPMI_OR_DIE(dspmi_fft_destroy(plan));
}

void dsp_fft_mag(double (*x)[2], unsigned int nx, double *xx)
{
    if (nx < 2) return;
    dspmi_fft_t plan = nullptr;
    DSPMI_OR_DIE(dspmi_fft_create(&plan, nx, DSPMI_F64, -1));
    const size_t bytes = (size_t)nx * 2 * sizeof(double);
    DevBuf din(bytes), dout(bytes), dmag((size_t)(nx / 2) * sizeof(double));
    DSPMI_HIP_OR_DIE(hipMemcpy(din.p, x, bytes, hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_c2c(plan, din.p, dout.p, 1, DSPMI_FORWARD, nullptr));
    DSPMI_OR_DIE(dspmi::ps_abs_f64(dout.p, dmag.as<double>(), nx / 2, (double)nx, 1, nx, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(xx, dmag.p, (size_t)(nx / 2) * sizeof(double), hipMemcpyDeviceToHost));
    DSPMI_OR_DIE(dspmi_fft_destroy(plan));
}

// nx/2 values read out of a transform of nx points: `mode` of the spectrum kernel, or the phase
static void run_readout(double (*x)[2], unsigned nx, double *out, int mode)
{
    const unsigned nbin = nx / 2;
    if (nbin == 0) return;
    DevBuf din((size_t)nbin * 2 * sizeof(double)), dout((size_t)nbin * sizeof(double));
    DSPMI_HIP_OR_DIE(hipMemcpy(din.p, x, (size_t)nbin * 2 * sizeof(double), hipMemcpyHostToDevice));
    if (mode < 0) DSPMI_OR_DIE(dspmi::phase_f64(dout.as<double>(), din.as<double>(), nbin, nbin, 1, nullptr));
    else DSPMI_OR_DIE(dspmi::power_spectrum_f64(dout.as<double>(), din.as<double>(), nbin, nbin, 1, mode, (double)nx, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(out, dout.p, (size_t)nbin * sizeof(double), hipMemcpyDeviceToHost));
}

void dsp_ps_percent(double (*x)[2], unsigned int nx, double *xx) { run_readout(x, nx, xx, DSPMI_PS_PERCENT); }

void dsp_ps_abs(double (*x)[2], unsigned int nx, double *xx) { run_readout(x, nx, xx, DSPMI_PS_ABS); }

void dsp_compute_phi(double (*x)[2], unsigned int nx, double *phi) { run_readout(x, nx, phi, -1); }

void dsp_make_blackman_window(double *x, unsigned int nx)
{
    // the denominator is nx, not nx - 1 (fir/src/main.c:143-144)
    for (unsigned i = 0; i < nx; i++)
        x[i] = 0.42 - 0.5 * cos((2.0 * M_PI * (double)i) / (double)nx) + 0.08 * cos((4 * M_PI * (double)i) / (double)nx);
}

void dsp_make_filter_kernel(const double *fresp, unsigned int nx, double (*kernel)[2])
{
    if (nx == 0) return;
    const unsigned n2 = 2 * nx;
    // spectrum: the response halved, then its mirror image (fir/src/main.c:166-179)
    std::vector<double> spec((size_t)n2 * 2, 0.0), imp((size_t)n2 * 2);
    for (unsigned i = 0; i < nx; i++) spec[2 * i] = fresp[i] / 2;
    for (unsigned i = 0; i < nx; i++) spec[2 * (nx + i)] = spec[2 * (nx - 1 - i)];
    run_c2c(reinterpret_cast<double(*)[2]>(spec.data()), n2, reinterpret_cast<double(*)[2]>(imp.data()), DSPMI_BACKWARD);
    // the first nk/2 impulse-response samples, mirrored about the centre of an nk-tap kernel
    const unsigned nk = n2 < 32 ? n2 : 32, half = nk / 2;
    for (unsigned i = 0; i < half; i++)
        for (int c = 0; c < 2; c++) kernel[half - 1 - i][c] = kernel[half + i][c] = imp[2 * i + c];
    for (unsigned i = nk; i < n2; i++) kernel[i][0] = kernel[i][1] = 0.0;
    std::vector<double> w(nk);
    dsp_make_blackman_window(w.data(), nk);
    for (unsigned i = 0; i < nk; i++) {
        kernel[i][0] *= w[i];
        kernel[i][1] *= w[i];
    }
}

void dsp_filter(double *xx, const double *x, unsigned int n, const double *coeffs)
{
    if (n == 0) return;
    dspmi_fft_t plan = nullptr;
    DSPMI_OR_DIE(dspmi_fft_create(&plan, n, DSPMI_F64, -1));
    const unsigned nbin = n / 2 + 1;
    DevBuf dx((size_t)n * sizeof(double)), dg((size_t)nbin * sizeof(double));
    DSPMI_HIP_OR_DIE(hipMemcpy(dx.p, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    DSPMI_HIP_OR_DIE(hipMemcpy(dg.p, coeffs, (size_t)nbin * sizeof(double), hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_filter(plan, dx.p, dx.p, dg.p, 1, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(xx, dx.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    DSPMI_OR_DIE(dspmi_fft_destroy(plan));
}

}  // extern "C"
