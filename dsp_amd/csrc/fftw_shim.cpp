// The FFTW3 subset the reference calls (include/dspmi_fftw3.h) over the device transforms.
// A plan mirrors fftw_plan: it remembers the host arrays it was created for
// (fftc/main.c:33: plan, execute, destroy per call; alsa/main.c:425-426,471: one plan
// executed every period) and owns its device buffers; the transform's own scratch is
// allocated by the first execute and kept, so later executes allocate nothing.
#include "common.h"

#include "../../include/dspmi_fftw3.h"

#include <cstdlib>

struct dspmi_fftw_plan_s {
    enum Kind { C2C, R2C, C2R } kind;
    unsigned n;
    int sign;
    void *host_in, *host_out;
    size_t in_bytes, out_bytes;
    void *dev_in, *dev_out;
    dspmi_fft_t fft;
};

namespace {

fftw_plan make_plan(dspmi_fftw_plan_s::Kind kind, int n, void *in, void *out, int sign)
{
    // FFTW returns NULL when it cannot plan; alsa/main.c:427 is the one caller that checks
    if (n <= 0 || !in || !out) return nullptr;
    dspmi_fftw_plan_s *p = new dspmi_fftw_plan_s();
    p->kind = kind;
    p->n = (unsigned)n;
    p->sign = sign;
    p->host_in = in;
    p->host_out = out;
    const size_t nbin = (size_t)n / 2 + 1;
    p->in_bytes = (kind == dspmi_fftw_plan_s::R2C ? (size_t)n : kind == dspmi_fftw_plan_s::C2R ? 2 * nbin : 2 * (size_t)n) * sizeof(double);
    p->out_bytes = (kind == dspmi_fftw_plan_s::C2R ? (size_t)n : kind == dspmi_fftw_plan_s::R2C ? 2 * nbin : 2 * (size_t)n) * sizeof(double);
    p->dev_in = p->dev_out = nullptr;
    p->fft = nullptr;
    if (dspmi_fft_create(&p->fft, p->n, DSPMI_F64, -1) != DSPMI_OK || hipMalloc(&p->dev_in, p->in_bytes) != hipSuccess ||
        hipMalloc(&p->dev_out, p->out_bytes) != hipSuccess) {
        fprintf(stderr, "libdspmi: fftw plan for n = %d failed: %s\n", n, dspmi_last_error());
        dspmi_fftw_destroy_plan(p);
        return nullptr;
    }
    return p;
}

}  // namespace

extern "C" {

void *dspmi_fftw_malloc(size_t n)
{
    void *p = nullptr;
    return posix_memalign(&p, 64, n ? n : 1) == 0 ? p : nullptr;
}

void dspmi_fftw_free(void *p) { free(p); }

fftw_plan dspmi_fftw_plan_dft_1d(int n, fftw_complex *in, fftw_complex *out, int sign, unsigned)
{
    if (sign != FFTW_FORWARD && sign != FFTW_BACKWARD) return nullptr;
    return make_plan(dspmi_fftw_plan_s::C2C, n, in, out, sign);
}

fftw_plan dspmi_fftw_plan_dft_r2c_1d(int n, double *in, fftw_complex *out, unsigned)
{
    return make_plan(dspmi_fftw_plan_s::R2C, n, in, out, FFTW_FORWARD);
}

fftw_plan dspmi_fftw_plan_dft_c2r_1d(int n, fftw_complex *in, double *out, unsigned)
{
    return make_plan(dspmi_fftw_plan_s::C2R, n, in, out, FFTW_BACKWARD);
}

void dspmi_fftw_execute(const fftw_plan p)
{
    if (!p) {
        // the plan could not be made (the reference's wrappers do not check): report why and
        // leave with the status the reference's tools use for their own errors
        fprintf(stderr, "libdspmi: fftw_execute(NULL): %s\n", dspmi_last_error());
        fflush(nullptr);
        _exit(255);
    }
    DSPMI_HIP_OR_DIE(hipMemcpy(p->dev_in, p->host_in, p->in_bytes, hipMemcpyHostToDevice));
    switch (p->kind) {
    case dspmi_fftw_plan_s::C2C: DSPMI_OR_DIE(dspmi_fft_exec_c2c(p->fft, p->dev_in, p->dev_out, 1, p->sign, nullptr)); break;
    case dspmi_fftw_plan_s::R2C: DSPMI_OR_DIE(dspmi_fft_exec_r2c(p->fft, p->dev_in, p->dev_out, 1, nullptr)); break;
    case dspmi_fftw_plan_s::C2R: DSPMI_OR_DIE(dspmi_fft_exec_c2r(p->fft, p->dev_in, p->dev_out, 1, nullptr)); break;
    }
    DSPMI_HIP_OR_DIE(hipMemcpy(p->host_out, p->dev_out, p->out_bytes, hipMemcpyDeviceToHost));
}

void dspmi_fftw_destroy_plan(fftw_plan p)
{
    if (!p) return;
    if (p->fft) dspmi_fft_destroy(p->fft);
    if (p->dev_in) (void)hipFree(p->dev_in);
    if (p->dev_out) (void)hipFree(p->dev_out);
    delete p;
}

}  // extern "C"
