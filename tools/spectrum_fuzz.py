"""Randomised check of dspmi_fir_spectrum_f32 against the two calls it replaces (FIR, then real
transform + normalised magnitude): tap counts 33..305, channel and frame counts, history, modes,
input scales.  usage: python tools/spectrum_fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsp_amd
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fft = dsp_amd.FftPlan(4096, "f32")
t0, ncase, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    nh = int(rng.integers(33, 306))
    h = (oracle.windowed_sinc_taps(nh) if rng.integers(0, 2) else rng.uniform(-1, 1, nh)).astype(np.float32)
    fir = dsp_amd.FirPlan(h, "f32")
    nchan, frames = int(rng.choice([1, 2, 5, rng.integers(6, 300)])), int(rng.choice([1, 2, 3, rng.integers(4, 20)]))
    scale = float(10.0 ** rng.integers(-4, 5))
    x = (torch.rand((nchan, frames * 4096), device="cuda") * 2 - 1) * scale
    hist = (torch.rand((nchan, nh - 1), device="cuda") * 2 - 1) * scale if rng.integers(0, 2) else None
    nbin, mode = int(rng.choice([2048, 2049])), int(rng.integers(0, 3))
    got = fir.spectrum(fft, x, hist=hist, nbin=nbin, mode=mode, scale=3.0)
    two = fft.r2c_spectrum(fir.exec(x, hist=hist).view(-1, 4096), nbin=nbin, mode=mode, scale=3.0)
    err = ((got - two).double().norm(dim=1) / two.double().norm(dim=1).clamp_min(1e-300)).max().item()
    worst = max(worst, err)
    if not err <= 1e-5:
        print("FAIL", dict(nh=nh, nchan=nchan, frames=frames, hist=hist is not None, nbin=nbin, mode=mode, scale=scale, err=err), flush=True)
        sys.exit(1)
    ncase += 1
    if ncase % 100 == 0:
        print("%d cases, %.0f s, worst %.2e" % (ncase, time.time() - t0, worst), flush=True)
print("ok: %d cases in %.0f s, worst relative L2 difference of a frame's spectrum %.2e" % (ncase, time.time() - t0, worst))
