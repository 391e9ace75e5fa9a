"""c2c of lengths that are not powers of two: the one-kernel Bluestein form against the five
separate kernels, on about 1 GiB of frames:  tools/bluestein_probe.py [f32|f64]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
esz = 8 if dtype == "f32" else 16
for n in (100, 200, 480, 482, 1000, 2000, 3000, 6000, 8000, 12000, 16000):
    batch = (1 << 30) // (n * esz)
    z = torch.view_as_complex(torch.rand(batch, n, 2, device="cuda", dtype=torch.float32 if dtype == "f32" else torch.float64) - 0.5)
    out = torch.empty_like(z)
    plan = dsp_amd.FftPlan(n, dtype)
    os.environ.pop("DSPMI_BLUESTEIN_COMPOSED", None)
    a = timed(lambda: plan.c2c(z, out=out), 5)
    os.environ["DSPMI_BLUESTEIN_COMPOSED"] = "1"
    b = timed(lambda: plan.c2c(z, out=out), 5)
    os.environ.pop("DSPMI_BLUESTEIN_COMPOSED", None)
    gb = 2.0 * esz * n * batch / 1e9
    print("n=%5d x %7d (%s): one kernel %.3f ms = %.2f TB/s of one read + one write | five kernels %.3f ms" % (
        n, batch, plan.variant, a, gb / a, b), flush=True)
