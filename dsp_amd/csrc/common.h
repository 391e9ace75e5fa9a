// Shared host-side helpers for libdspmi: error reporting and HIP call checks.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include <unistd.h>

#include "../../include/dspmi.h"

namespace dspmi {

// Thread-local last-error text, read through dspmi_last_error().
char *errbuf();
int fail(int code, const char *fmt, ...);

// A plan's buffers live on the device it was created for; executing it while another device is
// current would launch on that other device with foreign pointers.  Returns DSPMI_OK or fails.
inline int check_device(int plan_device, const char *what)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(DSPMI_EIO, "%s: hipGetDevice failed", what);
    if (cur != plan_device)
        return fail(DSPMI_EINVAL, "%s: the plan belongs to device %d but device %d is current", what, plan_device, cur);
    return DSPMI_OK;
}

// Whether any input row [x + i*x_stride, +nx) shares an element with any output row
// [y + j*y_stride, +ny) -- sizes and strides in elements of `esz` bytes.  Workgroups read the
// samples before their own tile (the filter's reach), which another workgroup may already have
// overwritten, so filtering in place is a race.  Rows interleaved in one buffer with a common
// stride are told apart exactly; with different strides the whole spans are compared.
inline bool rows_overlap(const void *x, size_t x_stride, size_t nx, const void *y, size_t y_stride, size_t ny, size_t nrow,
                         size_t esz)
{
    if (!x || !y || nx == 0 || ny == 0 || nrow == 0) return false;
    const char *a = static_cast<const char *>(x), *b = static_cast<const char *>(y);
    const size_t span_a = ((nrow - 1) * x_stride + nx) * esz, span_b = ((nrow - 1) * y_stride + ny) * esz;
    if (a + span_a <= b || b + span_b <= a) return false;
    if (x_stride != y_stride || nrow == 1) return true;
    const size_t row = x_stride * esz;
    const size_t r = b >= a ? (size_t)(b - a) % row : (row - (size_t)(a - b) % row) % row;  // (y - x) mod row
    return r < nx * esz || r + ny * esz > row;
}

// Kernel attributes (the dynamic-LDS limit) are per device: a launcher keeps one of these per
// kernel and sets the attribute the first time it runs on each device.  Two threads may both
// find the bit clear and both set the attribute -- harmless; the bit is set only afterwards, so
// nobody launches before the attribute is in place.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    static unsigned long long bit()
    {
        int d = 0;
        (void)hipGetDevice(&d);
        return 1ull << (d & 63);
    }
    bool pending() const { return (done.load(std::memory_order_acquire) & bit()) == 0; }
    void mark() { done.fetch_or(bit(), std::memory_order_release); }
};

// The f16 matrix-pipe form of an f32 FIR plan's taps (fir_direct.hip): the register images, the
// factor that undoes the taps' scale, the slice count, and the taps themselves (device, f32).
// Fails when the plan holds no such images.
int fir_half_images(dspmi_fir_t plan, const void **images, float *unscale, int *slices, unsigned *nh, int *device,
                    const float **taps);

}  // namespace dspmi

// Device-API check: record the failing call and return DSPMI_EIO.
#define DSPMI_HIP(call)                                                                      \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return dspmi::fail(DSPMI_EIO, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,      \
                               hipGetErrorString(e__));                                      \
    } while (0)

// Wrapper-path check: the reference's routines return void, so a device error
// cannot be reported to the caller; fail loudly instead of computing on the CPU.
#define DSPMI_HIP_OR_DIE(call)                                                               \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) {                                                             \
            fprintf(stderr, "libdspmi: %s:%d: %s -> %s\n", __FILE__, __LINE__, #call,        \
                    hipGetErrorString(e__));                                                 \
            abort();                                                                         \
        }                                                                                    \
    } while (0)

// A size or argument the library refuses ends the program with status 255, the status the
// reference's tools return on their own errors (util/fft/fft.c, util/filter/filter.c: return -1
// from main); a device failure aborts.  The process ends with _exit after its output streams are
// flushed: exit() would run the host program's atexit handlers and the HIP runtime's static
// destructors from inside a library call, possibly with work still queued on the device.
#define DSPMI_OR_DIE(call)                                                                   \
    do {                                                                                     \
        int r__ = (call);                                                                    \
        if (r__ != DSPMI_OK) {                                                               \
            fprintf(stderr, "libdspmi: %s:%d: %s failed (%d): %s\n", __FILE__, __LINE__,     \
                    #call, r__, dspmi_last_error());                                         \
            if (r__ == DSPMI_EIO) abort();                                                   \
            fflush(nullptr);                                                                 \
            _exit(255);                                                                      \
        }                                                                                    \
    } while (0)
