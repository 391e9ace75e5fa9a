"""Real frames -> normalised magnitude spectrum (one kernel), 2^28 samples, by frame length."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


total = 1 << 28
for n in [int(a) for a in sys.argv[1:]] or (1024, 2048, 4096, 8192):
    x = torch.rand(total // n, n, device="cuda") - 0.5
    ps = torch.empty(total // n, n // 2 + 1, device="cuda")
    bins = torch.empty(total // n, n // 2 + 1, dtype=torch.complex64, device="cuda")
    plan = dsp_amd.FftPlan(n, "f32")
    t1 = timed(lambda: plan.r2c_spectrum(x, out=ps))
    t2 = timed(lambda: plan.r2c(x, out=bins))
    t3 = timed(lambda: plan.c2r(bins, out=x))
    print("n=%5d  r2c+spectrum %.3f ms (%.0f Gsamples/s)   r2c %.3f ms   c2r %.3f ms" % (n, t1, total / t1 / 1e6, t2, t3), flush=True)
