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

struct FftKernel {
    unsigned n;
    fft_launch_fn fn;         // default cache policy
    fft_launch_fn fn_stream;  // nontemporal loads and stores: batches far larger than the caches
    const char *name;
};

// Kernel for a power-of-two length, or nullptr.
const FftKernel *fft_find(unsigned n, int precision);

}  // namespace dspmi
