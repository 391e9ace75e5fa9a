"""Long filters on rows of a few thousand samples: overlap-save with one or two long blocks per row
against the direct matrix kernel, and what an f32 FIR plan picks by itself.  512 channels; ms per launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def timed(fn, n=6):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n


def main():
    nchan = 512
    for nh in (900, 2049, 4096, 8191):
        h = oracle.windowed_sinc_taps(nh).astype(np.float32)
        os.environ["DSPMI_FIR_VARIANT"] = "mfma"
        direct = dsp_amd.FirPlan(h, "f32")
        del os.environ["DSPMI_FIR_VARIANT"]
        plans = {n: dsp_amd.OlsPlan(h, n) for n in (16384, 32768) if n >= 2 * nh}
        auto = dsp_amd.FirPlan(h, "f32")
        for nx in (512, 1024, 2048, 4096, 8192, 16384, 32768, 49152):
            x = torch.rand((nchan, nx), device="cuda") * 2 - 1
            y = torch.empty_like(x)
            row = ["direct %8.3f" % timed(lambda: direct.exec(x, nx, y=y))]
            for n, pl in plans.items():
                row.append("ols%d %8.3f" % (n, timed(lambda: pl.exec(x, y=y))))
            row.append("plan %8.3f" % timed(lambda: auto.exec(x, nx, y=y)))
            print("nh %5d nx %6d (nx*nh %.1e): %s" % (nh, nx, nx * nh, "   ".join(row)), flush=True)


if __name__ == "__main__":
    main()
