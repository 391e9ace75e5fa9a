"""Mid-length filters (306 .. 899 taps) on many short rows: the fused overlap-save kernels by transform
length against the direct matrix kernel.  2^26 samples as rows of 256 ... 8192; ms per launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def timed(fn, n=8):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n


def main():
    total = 1 << 26
    for nh in (306, 511, 800):
        h = oracle.windowed_sinc_taps(nh).astype(np.float32)
        os.environ["DSPMI_FIR_VARIANT"] = "mfma"
        direct = dsp_amd.FirPlan(h, "f32")
        del os.environ["DSPMI_FIR_VARIANT"]
        auto = dsp_amd.FirPlan(h, "f32")
        plans = {n: dsp_amd.OlsPlan(h, n) for n in (1024, 2048, 4096, 16384) if n >= 2 * nh}
        for nx in (256, 512, 1024, 2048, 4096, 8192):
            nchan = total // nx
            x = torch.rand((nchan, nx), device="cuda") * 2 - 1
            y = torch.empty_like(x)
            hist = torch.rand((nchan, nh - 1), device="cuda") * 2 - 1   # carried history: every output is steady state
            row = ["direct %7.3f" % timed(lambda: direct.exec(x, nx, y=y, hist=hist))]
            for n, pl in plans.items():
                row.append("ols%d %7.3f" % (n, timed(lambda: pl.exec(x, y=y, hist=hist))))
            row.append("plan(%s) %7.3f" % (auto.variant.rsplit("_", 1)[1], timed(lambda: auto.exec(x, nx, y=y, hist=hist))))
            print("nh %4d rows of %5d: %s" % (nh, nx, "  ".join(row)), flush=True)


if __name__ == "__main__":
    main()
