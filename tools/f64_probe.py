"""Double-precision device paths (what the dsp_* host wrappers run): batched FIR and c2c FFT."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

nchan, nx = 256, 1 << 20
x = torch.rand(nchan, nx, device="cuda", dtype=torch.float64) - 0.5
y = torch.empty_like(x)
for nh in (16, 127, 1024):
    h = oracle.windowed_sinc_taps(nh) if nh != 16 else np.asarray(oracle.TAPS16)
    plan = dsp_amd.FirPlan(h, "f64")
    ms = timed(lambda: plan.exec(x, y=y), 5)
    gs = nchan * nx / ms / 1e6
    print("f64 FIR %4d taps: %.3f ms  %.1f Gsamples/s  %.2f TB/s  %.1f TFLOP/s (2*nh flop/sample; mul and add are separate instructions)"
          % (nh, ms, gs, gs * 16 / 1e3, gs * 2 * nh / 1e3), flush=True)
for n in (1024, 4096):
    batch = (1 << 29) // n // 2
    z = torch.view_as_complex(torch.rand(batch, n, 2, device="cuda", dtype=torch.float64) - 0.5)
    out = torch.empty_like(z)
    plan = dsp_amd.FftPlan(n, "f64")
    ms = timed(lambda: plan.c2c(z, out=out), 5)
    print("f64 c2c n=%d x %d: %.3f ms  %.2f TB/s" % (n, batch, ms, 32 * n * batch / ms / 1e9), flush=True)
