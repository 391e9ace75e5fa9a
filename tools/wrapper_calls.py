#!/usr/bin/env python3
"""Time per call of the host drop-in routines when called repeatedly with the same taps / length,
as the reference's period loops do: first call (plan + buffers set up) against the calls after it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsp_amd
import oracle

rng = np.random.default_rng(0)
h = np.asarray(oracle.TAPS16, dtype=np.float64)
for name, nx in (("dsp_conv0, 16 taps", 2048), ("dsp_conv0, 16 taps", 1 << 20)):
    x = rng.standard_normal(nx)
    t0 = time.perf_counter(); dsp_amd.conv0(x, h); first = time.perf_counter() - t0
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); dsp_amd.conv0(x, h); ts.append(time.perf_counter() - t0)
    print("%-22s nx %8d: first call %8.3f ms, then median %7.3f ms" % (name, nx, first * 1e3, np.median(ts) * 1e3), flush=True)
    dsp_amd.lib().dspmi_dropin_release()
for n in (4096, 48000):
    x = rng.standard_normal(n)
    t0 = time.perf_counter(); dsp_amd.fft_r2c(x); first = time.perf_counter() - t0
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); dsp_amd.fft_r2c(x); ts.append(time.perf_counter() - t0)
    print("%-22s n  %8d: first call %8.3f ms, then median %7.3f ms" % ("dsp_fft_r2c", n, first * 1e3, np.median(ts) * 1e3), flush=True)
