// Shared host-side helpers for libdspmi: error reporting and HIP call checks.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/dspmi.h"

namespace dspmi {

// Thread-local last-error text, read through dspmi_last_error().
char *errbuf();
int fail(int code, const char *fmt, ...);

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

#define DSPMI_OR_DIE(call)                                                                   \
    do {                                                                                     \
        int r__ = (call);                                                                    \
        if (r__ != DSPMI_OK) {                                                               \
            fprintf(stderr, "libdspmi: %s:%d: %s failed (%d): %s\n", __FILE__, __LINE__,     \
                    #call, r__, dspmi_last_error());                                         \
            abort();                                                                         \
        }                                                                                    \
    } while (0)
