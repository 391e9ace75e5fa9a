// Host drop-in wrappers: the reference's C signatures over the device kernels.
//
// Each call stages its host double buffers to the device, runs the double
// precision kernels and blocks until the result is back in host memory -- the
// synchronous, caller-owned-buffer behaviour of the reference's static
// routines.  Any device failure prints the failing call and abort()s: these
// routines return void in the reference, and there is no CPU fallback.
//
// The reference's callers run these routines once per period with the same
// taps and window length (alsa/main.c:494-516 inside the capture loop,
// fft/src/main.c:343-349), so plans and staging buffers are kept between
// calls: the last few transform lengths and tap sets per device, and staging
// buffers that only grow (anything over 256 MiB is given back at the end of the
// call).  One mutex serialises the wrappers; dspmi_dropin_release() frees
// everything that is kept.
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

// What the wrappers keep between calls.  Never destroyed at exit (the HIP runtime may be gone by
// then); dspmi_dropin_release() empties it.
struct Kept {
    std::mutex m;
    unsigned long clock = 0;
    struct Fft {
        int dev;
        unsigned n;
        dspmi_fft_t plan;
        unsigned long used;
    };
    struct Fir {
        int dev;
        std::vector<double> taps;
        dspmi_fir_t plan;
        unsigned long used;
    };
    struct Buf {
        int dev, slot;
        void *p;
        size_t bytes;
    };
    std::vector<Fft> fft;
    std::vector<Fir> fir;
    std::vector<Buf> bufs;
};
constexpr size_t kKeepFft = 6, kKeepFir = 4;
constexpr size_t kKeepBufBytes = (size_t)256 << 20;
constexpr unsigned kKeepFftPoints = 1u << 20;  // longer plans own hundreds of MiB of work frames: made and destroyed per call

Kept &kept()
{
    static Kept *k = new Kept;
    return *k;
}

int current_device()
{
    int dev = 0;
    DSPMI_HIP_OR_DIE(hipGetDevice(&dev));
    return dev;
}

// staging buffer `slot` of the current device, at least `bytes` long (contents undefined)
void *staging(int slot, size_t bytes)
{
    Kept &k = kept();
    const int dev = current_device();
    if (bytes == 0) bytes = 8;
    for (Kept::Buf &b : k.bufs)
        if (b.dev == dev && b.slot == slot) {
            if (b.bytes < bytes) {
                DSPMI_HIP_OR_DIE(hipFree(b.p));
                b.p = nullptr;
                b.bytes = 0;
                DSPMI_HIP_OR_DIE(hipMalloc(&b.p, bytes));
                b.bytes = bytes;
            }
            return b.p;
        }
    Kept::Buf b{dev, slot, nullptr, bytes};
    DSPMI_HIP_OR_DIE(hipMalloc(&b.p, bytes));
    k.bufs.push_back(b);
    return b.p;
}

// end of a call: large staging buffers go back to the device
void trim_staging()
{
    Kept &k = kept();
    for (size_t i = 0; i < k.bufs.size();)
        if (k.bufs[i].bytes > kKeepBufBytes) {
            DSPMI_HIP_OR_DIE(hipFree(k.bufs[i].p));
            k.bufs.erase(k.bufs.begin() + (long)i);
        } else {
            i++;
        }
}

// A transform plan of n points in double precision on the current device; `owned` tells the
// caller to destroy it after use (too long to keep).
dspmi_fft_t fft_plan(unsigned n, bool *owned)
{
    Kept &k = kept();
    const int dev = current_device();
    *owned = false;
    for (Kept::Fft &e : k.fft)
        if (e.dev == dev && e.n == n) {
            e.used = ++k.clock;
            return e.plan;
        }
    dspmi_fft_t plan = nullptr;
    DSPMI_OR_DIE(dspmi_fft_create(&plan, n, DSPMI_F64, dev));
    if (n > kKeepFftPoints) {
        *owned = true;
        return plan;
    }
    if (k.fft.size() >= kKeepFft) {
        size_t old = 0;
        for (size_t i = 1; i < k.fft.size(); i++)
            if (k.fft[i].used < k.fft[old].used) old = i;
        DSPMI_OR_DIE(dspmi_fft_destroy(k.fft[old].plan));
        k.fft.erase(k.fft.begin() + (long)old);
    }
    k.fft.push_back({dev, n, plan, ++k.clock});
    return plan;
}

void fft_done(dspmi_fft_t plan, bool owned)
{
    if (owned) DSPMI_OR_DIE(dspmi_fft_destroy(plan));
    trim_staging();
}

// A double-precision FIR plan for these taps on the current device (compared value by value).
dspmi_fir_t fir_plan(const double *h, unsigned nh)
{
    Kept &k = kept();
    const int dev = current_device();
    for (Kept::Fir &e : k.fir)
        if (e.dev == dev && e.taps.size() == nh && std::memcmp(e.taps.data(), h, (size_t)nh * sizeof(double)) == 0) {
            e.used = ++k.clock;
            return e.plan;
        }
    dspmi_fir_t plan = nullptr;
    DSPMI_OR_DIE(dspmi_fir_create_f64(&plan, h, nh, dev));
    if (k.fir.size() >= kKeepFir) {
        size_t old = 0;
        for (size_t i = 1; i < k.fir.size(); i++)
            if (k.fir[i].used < k.fir[old].used) old = i;
        DSPMI_OR_DIE(dspmi_fir_destroy(k.fir[old].plan));
        k.fir.erase(k.fir.begin() + (long)old);
    }
    k.fir.push_back({dev, std::vector<double>(h, h + nh), plan, ++k.clock});
    return plan;
}

enum Order { ASCENDING, DESCENDING };

void run_fir(double *y, unsigned nxx, const double *x, unsigned nx, const double *h, unsigned nh, unsigned flags,
             Order order)
{
    if (nxx == 0) return;
    if (nh == 0) {  // an empty tap set sums nothing
        std::memset(y, 0, (size_t)nxx * sizeof(double));
        return;
    }
    std::lock_guard<std::mutex> lock(kept().m);
    dspmi_fir_t plan = fir_plan(h, nh);
    double *dx = static_cast<double *>(staging(0, (size_t)nx * sizeof(double)));
    double *dy = static_cast<double *>(staging(1, (size_t)nxx * sizeof(double)));
    if (nx) DSPMI_HIP_OR_DIE(hipMemcpy(dx, x, (size_t)nx * sizeof(double), hipMemcpyHostToDevice));
    if (order == DESCENDING) DSPMI_OR_DIE(dspmi_fir_exec_f64_desc(plan, dy, nxx, nxx, dx, nx, nx, 1, flags, nullptr));
    else DSPMI_OR_DIE(dspmi_fir_exec_f64(plan, dy, nxx, nxx, dx, nx, nx, 1, nullptr, 0, flags, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(y, dy, (size_t)nxx * sizeof(double), hipMemcpyDeviceToHost));
    trim_staging();
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
    std::lock_guard<std::mutex> lock(kept().m);
    bool owned = false;
    dspmi_fft_t plan = fft_plan(nx, &owned);
    const size_t bytes = (size_t)nx * 2 * sizeof(double);
    void *din = staging(0, bytes), *dout = staging(1, bytes);
    DSPMI_HIP_OR_DIE(hipMemcpy(din, in, bytes, hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_c2c(plan, din, dout, 1, sign, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    fft_done(plan, owned);
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
    std::lock_guard<std::mutex> lock(kept().m);
    bool owned = false;
    dspmi_fft_t plan = fft_plan(n, &owned);
    const unsigned nbin = n / 2 + 1;
    void *din = staging(0, (size_t)n * sizeof(double)), *dout = staging(1, (size_t)nbin * 2 * sizeof(double));
    DSPMI_HIP_OR_DIE(hipMemcpy(din, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_r2c(plan, din, dout, 1, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(xx, dout, (size_t)nbin * 2 * sizeof(double), hipMemcpyDeviceToHost));
    fft_done(plan, owned);
}

void dsp_fft_mag(double (*x)[2], unsigned int nx, double *xx)
{
    if (nx < 2) return;
    std::lock_guard<std::mutex> lock(kept().m);
    bool owned = false;
    dspmi_fft_t plan = fft_plan(nx, &owned);
    const size_t bytes = (size_t)nx * 2 * sizeof(double);
    void *din = staging(0, bytes), *dout = staging(1, bytes);
    double *dmag = static_cast<double *>(staging(2, (size_t)(nx / 2) * sizeof(double)));
    DSPMI_HIP_OR_DIE(hipMemcpy(din, x, bytes, hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_c2c(plan, din, dout, 1, DSPMI_FORWARD, nullptr));
    DSPMI_OR_DIE(dspmi::ps_abs_f64(dout, dmag, nx / 2, (double)nx, 1, nx, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(xx, dmag, (size_t)(nx / 2) * sizeof(double), hipMemcpyDeviceToHost));
    fft_done(plan, owned);
}

// nx/2 values read out of a transform of nx points: `mode` of the spectrum kernel, or the phase
static void run_readout(double (*x)[2], unsigned nx, double *out, int mode)
{
    const unsigned nbin = nx / 2;
    if (nbin == 0) return;
    std::lock_guard<std::mutex> lock(kept().m);
    double *din = static_cast<double *>(staging(0, (size_t)nbin * 2 * sizeof(double)));
    double *dout = static_cast<double *>(staging(1, (size_t)nbin * sizeof(double)));
    DSPMI_HIP_OR_DIE(hipMemcpy(din, x, (size_t)nbin * 2 * sizeof(double), hipMemcpyHostToDevice));
    if (mode < 0) DSPMI_OR_DIE(dspmi::phase_f64(dout, din, nbin, nbin, 1, nullptr));
    else DSPMI_OR_DIE(dspmi::power_spectrum_f64(dout, din, nbin, nbin, 1, mode, (double)nx, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(out, dout, (size_t)nbin * sizeof(double), hipMemcpyDeviceToHost));
    trim_staging();
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
    std::lock_guard<std::mutex> lock(kept().m);
    bool owned = false;
    dspmi_fft_t plan = fft_plan(n, &owned);
    const unsigned nbin = n / 2 + 1;
    void *dx = staging(0, (size_t)n * sizeof(double)), *dg = staging(1, (size_t)nbin * sizeof(double));
    DSPMI_HIP_OR_DIE(hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    DSPMI_HIP_OR_DIE(hipMemcpy(dg, coeffs, (size_t)nbin * sizeof(double), hipMemcpyHostToDevice));
    DSPMI_OR_DIE(dspmi_fft_exec_filter(plan, dx, dx, dg, 1, nullptr));
    DSPMI_HIP_OR_DIE(hipMemcpy(xx, dx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    fft_done(plan, owned);
}

void dspmi_dropin_release(void)
{
    Kept &k = kept();
    std::lock_guard<std::mutex> lock(k.m);
    int prev = 0;
    DSPMI_HIP_OR_DIE(hipGetDevice(&prev));
    for (Kept::Fft &e : k.fft) DSPMI_OR_DIE(dspmi_fft_destroy(e.plan));
    for (Kept::Fir &e : k.fir) DSPMI_OR_DIE(dspmi_fir_destroy(e.plan));
    for (Kept::Buf &b : k.bufs) {
        DSPMI_HIP_OR_DIE(hipSetDevice(b.dev));
        DSPMI_HIP_OR_DIE(hipFree(b.p));
    }
    DSPMI_HIP_OR_DIE(hipSetDevice(prev));
    k.fft.clear();
    k.fir.clear();
    k.bufs.clear();
}

}  // extern "C"
