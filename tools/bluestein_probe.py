"""c2c of lengths that are not powers of two (f32): the one-kernel Bluestein form against the five
separate kernels, on about 2^27 complex points."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

for n in (100, 480, 482, 1000, 3000, 6000):
    batch = (1 << 27) // n
    z = torch.view_as_complex(torch.rand(batch, n, 2, device="cuda") - 0.5)
    out = torch.empty_like(z)
    plan = dsp_amd.FftPlan(n, "f32")
    os.environ.pop("DSPMI_BLUESTEIN_COMPOSED", None)
    a = timed(lambda: plan.c2c(z, out=out), 5)
    os.environ["DSPMI_BLUESTEIN_COMPOSED"] = "1"
    b = timed(lambda: plan.c2c(z, out=out), 5)
    os.environ.pop("DSPMI_BLUESTEIN_COMPOSED", None)
    gb = 16.0 * n * batch / 1e9
    print("n=%5d x %7d (%s): one kernel %.3f ms = %.2f TB/s of 16 B/point | five kernels %.3f ms" % (
        n, batch, plan.variant, a, gb / a, b), flush=True)
