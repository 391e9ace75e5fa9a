"""The stream kernel (up to 32 taps) against the tile (matrix) kernel: 2^29 samples as 512 long rows and
2^27 samples as rows of 2048; ms per launch, alternating."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def timed(fn, n=30, warm=30):
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
    for nchan, nx in ((512, 1 << 20), (65536, 2048)):
        x = torch.rand((nchan, nx), device="cuda") * 2 - 1
        y = torch.empty_like(x)
        for nh in (4, 8, 16, 20, 21, 24, 28, 32):
            h = oracle.windowed_sinc_taps(max(nh, 9))[:nh].astype(np.float32)
            plan = dsp_amd.FirPlan(h, "f32")
            out = []
            for rep in range(2):
                for v in ("1", "0"):
                    os.environ["DSPMI_FIR_STREAM"] = v
                    out.append("%s %.4f" % ("stream" if v == "1" else "tile", timed(lambda: plan.exec(x, nx, y=y))))
            os.environ.pop("DSPMI_FIR_STREAM")
            print("%d x %d, %2d taps (%s): %s" % (nchan, nx, nh, plan.variant, "  ".join(out)), flush=True)


if __name__ == "__main__":
    main()
