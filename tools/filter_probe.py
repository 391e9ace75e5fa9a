"""Times dspmi_fft_exec_filter on a large f32 batch, one-kernel form against the composition."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402

for n, batch in ((1024, 1 << 19), (4096, 1 << 17), (256, 1 << 21)):
    x = torch.rand(batch, n, device="cuda") - 0.5
    out = torch.empty_like(x)
    g = torch.rand(n // 2 + 1, device="cuda")
    plan = dsp_amd.FftPlan(n, "f32")
    for composed in ("0", "1"):
        os.environ["DSPMI_FILTER_COMPOSED"] = composed
        for _ in range(2):
            plan.filter(x, g, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            plan.filter(x, g, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("n=%d batch=%d %s: %.3f ms, %.1f Gsamples/s, %.2f TB/s of 8 B/sample" % (
            n, batch, "composed" if composed == "1" else "one kernel", ms, batch * n / ms / 1e6, 8 * batch * n / ms / 1e9), flush=True)
