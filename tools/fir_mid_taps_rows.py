#!/usr/bin/env python3
"""306 .. 529 taps by row length: what the plan runs (overlap-save by transform length) against the
f16 matrix kernel pinned (DSPMI_FIR_VARIANT=f16).  2^27 samples per launch, Gsamples/s."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402

total = 1 << 27
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
taps = [int(a) for a in sys.argv[1:]] or [306, 400, 511, 625]
ROWS = tuple(int(r) for r in os.environ["ROWS"].split(",")) if "ROWS" in os.environ else (1024, 4096, 16384, 32768, 65536, 131072, 262144, 1 << 20)
for nh in taps:
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    for pin in (None, "f16"):
        if pin:
            os.environ["DSPMI_FIR_VARIANT"] = pin
        else:
            os.environ.pop("DSPMI_FIR_VARIANT", None)
        plan = dsp_amd.FirPlan(h, "f32")
        row = []
        for nx in ROWS:
            nchan = total // nx
            x = torch.rand((nchan, nx), device="cuda") * 2 - 1
            y = torch.empty_like(x)
            for _ in range(5):
                plan.exec(x, nx, y=y)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(10):
                plan.exec(x, nx, y=y)
            ev1.record()
            torch.cuda.synchronize()
            row.append("%6.0f" % (total / (ev0.elapsed_time(ev1) / 10) / 1e6))
        print("nh %4d %-36s rows of %s: %s" % (nh, plan.variant, " ".join(str(r) for r in ROWS), " ".join(row)), flush=True)
os.environ.pop("DSPMI_FIR_VARIANT", None)
