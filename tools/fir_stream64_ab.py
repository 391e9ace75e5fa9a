"""The stream kernel in double against the double tile kernel (DSPMI_FIR_STREAM=0), by tap count and row
length; ms per launch.  (Equality bit for bit is what tests/test_fir_gpu.py checks.)"""
import os
import sys

import numpy as np
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


def main():
    rng = np.random.default_rng(5)
    shapes = [(256, 1 << 20, (4, 8, 12, 16))] + [((1 << 27) // r, r, (8, 16, 21, 24, 32)) for r in (256, 1024)]
    for nchan, nx, taps in shapes:
        x = torch.rand((nchan, nx), dtype=torch.float64, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        print("%d rows of %d" % (nchan, nx))
        for nh in taps:
            plan = dsp_amd.FirPlan(rng.uniform(-1, 1, nh), "f64")
            out = []
            for tile in (False, True):
                if tile:
                    os.environ["DSPMI_FIR_STREAM"] = "0"
                else:
                    os.environ.pop("DSPMI_FIR_STREAM", None)
                ms = timed(lambda: plan.exec(x, nx, y=y))
                out.append("%s %.3f ms (%.0f GB/s)" % ("tile" if tile else "plan", ms, 16.0 * nchan * nx / ms / 1e6))
            os.environ.pop("DSPMI_FIR_STREAM", None)
            print("%2d taps: %s" % (nh, "  ".join(out)), flush=True)


if __name__ == "__main__":
    main()
