"""dsp_amd -- Python harness over libdspmi.so (the C-ABI / HIP library).

The product is the shared library (dsp_amd/libdspmi.so, declared in
include/dspmi.h); this package only loads it through ctypes so tests and
bench.py can drive it with torch-owned device memory.  There is no Python or
CPU implementation of any routine here: if the library is missing or fails to
load, importing the bindings raises.
"""
from .lib import (  # noqa: F401
    DspmiError, FirPlan, FftPlan, OlsPlan, ToneGen, power_spectrum, phase, lib, lib_path,
    PS_NORM_GUARDED, PS_NORM, PS_ABS, PS_PERCENT,
    conv0, conv1, convolve, convolve_alloc, fir, fir_set_taps,
    dft, idft, fft_r2c, fft_mag, filter, make_blackman_window, make_filter_kernel,
    ps_percent, ps_abs, compute_phi,
    FORWARD, BACKWARD, F32, F64, FIR_DROP_FIRST,
)
