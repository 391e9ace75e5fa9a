"""r2c, c2r and r2c + spectrum of 4096- and 2048-sample frames (f32), 2^28 samples: ms per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402


def timed(fn, n=20, warm=20):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n


total = 1 << 28
for n in (2048, 4096, 8192):
    batch = total // n
    x = torch.rand(batch, n, device="cuda") - 0.5
    bins = torch.empty(batch, n // 2 + 1, dtype=torch.complex64, device="cuda")
    ps = torch.empty(batch, n // 2 + 1, device="cuda")
    plan = dsp_amd.FftPlan(n, "f32")
    print("n=%5d  r2c %.4f  c2r %.4f  r2c+spectrum %.4f ms" % (n, timed(lambda: plan.r2c(x, out=bins)), timed(lambda: plan.c2r(bins, out=x)),
                                                            timed(lambda: plan.r2c_spectrum(x, out=ps))), flush=True)
