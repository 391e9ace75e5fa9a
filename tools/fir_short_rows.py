"""f32 FIR plans on many short rows: 2^27 samples as rows of 256 ... 2^20 samples; Gsamples/s by tap count."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def main():
    total = 1 << 27
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for nh in (16, 21, 32, 127, 305, 511):
        h = (np.asarray(oracle.TAPS16) if nh == 16 else oracle.windowed_sinc_taps(nh)).astype(np.float32)
        plan = dsp_amd.FirPlan(h, "f32")
        row = []
        for nx in (256, 1024, 2048, 4096, 16384, 1 << 20):
            nchan = total // nx
            x = torch.rand((nchan, nx), device="cuda") * 2 - 1
            y = torch.empty_like(x)
            for _ in range(40):
                plan.exec(x, nx, y=y)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(10):
                plan.exec(x, nx, y=y)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 10
            row.append("%6.0f" % (total / ms / 1e6))
        print("nh %4d %-36s Gsamples/s at rows of 256 1024 2048 4096 16384 2^20: %s" % (nh, plan.variant, " ".join(row)), flush=True)


if __name__ == "__main__":
    main()
