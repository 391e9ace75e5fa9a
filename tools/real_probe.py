"""r2c / c2r (f32) by frame length: fused kernels against the two-kernel compositions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

total = 1 << 28
for n in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
    batch = total // n
    nb = n // 2 + 1
    x = torch.rand(batch, n, device="cuda") - 0.5
    bins = torch.empty(batch, nb, dtype=torch.complex64, device="cuda")
    plan = dsp_amd.FftPlan(n, "f32")
    row = []
    for composed in (False, True):
        if composed:
            os.environ["DSPMI_FFT_REAL_COMPOSED"] = "1"
        else:
            os.environ.pop("DSPMI_FFT_REAL_COMPOSED", None)
        row.append(timed(lambda: plan.r2c(x, out=bins)))
        row.append(timed(lambda: plan.c2r(bins, out=x)))
    print("n=%5d  r2c fused %.3f composed %.3f ms | c2r fused %.3f composed %.3f ms" % (n, row[0], row[2], row[1], row[3]), flush=True)
