"""Time of the plan's own kernel by tap count on the bench's FIR shape (512 channels x 2^20
samples), for A/B runs of two library builds (DSPMI_LIB selects the other one):
  python tools/fir_taps_time.py [taps...]      prints one line per tap count: variant, ms, Gsamples/s"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import dsp_amd

taps_list = [int(a) for a in sys.argv[1:]] or [33, 63, 97, 113, 127, 160, 191, 255, 305]
nchan, nx = 512, 1 << 20
both = torch.empty((2, nchan, nx), dtype=torch.float32, device="cuda")
both[0].uniform_(-1.0, 1.0, generator=torch.Generator(device="cuda").manual_seed(3))
x, y = both[0], both[1]
tag = os.path.basename(os.environ.get("DSPMI_LIB", "libdspmi.so"))
for nh in taps_list:
    plan = dsp_amd.FirPlan(bench.TAPS16 if nh == 16 else bench.sinc_taps(nh), "f32")
    for _ in range(30):
        plan.exec(x, nx, y=y)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.exec(x, nx, y=y)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20)
    ms = sorted(best)[1]
    print("%s taps %4d %-40s %.4f ms %7.1f Gsamples/s" % (tag, nh, plan.variant, ms, nchan * nx / ms / 1e6))
