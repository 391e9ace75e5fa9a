"""r2c + spectrum (n = 4096) per-frame time by working-set size: cache-resident against streamed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

n = 4096
plan = dsp_amd.FftPlan(n, "f32")
for batch in (4096, 8192, 16384, 65536, 262144):
    x = torch.rand(batch, n, device="cuda") - 0.5
    ps = torch.empty(batch, n // 2 + 1, device="cuda")
    bins = torch.empty(batch, n // 2 + 1, dtype=torch.complex64, device="cuda")
    a = timed(lambda: plan.r2c_spectrum(x, out=ps), 10)
    b = timed(lambda: plan.r2c(x, out=bins), 10)
    print("batch %7d (%5.0f MB in): spectrum %.4f ms = %.2f ns/frame | bins %.4f ms = %.2f ns/frame"
          % (batch, batch * n * 4 / 1e6, a, a * 1e6 / batch, b, b * 1e6 / batch), flush=True)
