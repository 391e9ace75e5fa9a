"""Times dspmi_fft_exec_c2r (f32): the fused kernel against the merge + transform pair (which an
output buffer overlapping the input selects)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if __name__ == "__main__":
  for n, batch in ((2048, 1 << 19), (8192, 1 << 17), (512, 1 << 21), (64, 1 << 24)):
      nb = n // 2 + 1
      flat = torch.rand(batch * nb * 2, device="cuda") - 0.5
      bins = torch.view_as_complex(flat.view(batch, nb, 2))
      out = torch.empty(batch, n, device="cuda")
      plan = dsp_amd.FftPlan(n, "f32")
      ms = timed(lambda: plan.c2r(bins, out=out))
      ms2 = timed(lambda: plan.c2r(bins, out=flat[: batch * n].view(batch, n)))
      gb = batch * (8 * nb + 4 * n) / 1e9
      print("n=%d batch=%d: fused %.3f ms (%.2f TB/s), two kernels %.3f ms" % (n, batch, ms, gb / ms, ms2), flush=True)
      r = timed(lambda: plan.r2c(out, out=bins))
      print("   r2c for comparison: %.3f ms (%.2f TB/s)" % (r, gb / r), flush=True)
