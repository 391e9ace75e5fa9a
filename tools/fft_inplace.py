"""c2c N=1024 x 2^20 frames: separate output buffer against in place (same bytes moved)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

n, batch = 1024, 1 << 20
z = torch.view_as_complex(torch.rand(batch, n, 2, device="cuda") - 0.5)
out = torch.empty_like(z)
plan = dsp_amd.FftPlan(n, "f32")
for rep in range(3):
    a = timed(lambda: plan.c2c(z, dsp_amd.FORWARD, out=out), 10)
    b = timed(lambda: plan.c2c(out, dsp_amd.FORWARD, out=out), 10)
    print("out of place %.3f ms %.2f TB/s | in place %.3f ms %.2f TB/s" % (a, 16 * n * batch / a / 1e9, b, 16 * n * batch / b / 1e9), flush=True)
