// Library-wide state of libdspmi: per-thread error text, version, device count.
#include "common.h"

#include <cstring>

namespace dspmi {

char *errbuf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(errbuf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace dspmi

extern "C" {

const char *dspmi_last_error(void) { return dspmi::errbuf(); }

int dspmi_abi_version(void) { return 1; }

int dspmi_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return dspmi::fail(DSPMI_EIO, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

}  // extern "C"
