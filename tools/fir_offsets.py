"""FIR 512 ch x 2^20 (16 and 127 taps) with the output buffer at different offsets from the input:
the same question as tools/fft_offsets.py for the FIR kernels."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

nchan, nx = 512, 1 << 20
nfl = nchan * nx
buf = torch.empty(2 * nfl + (80 << 20) // 4, dtype=torch.float32, device="cuda")
buf.uniform_(-1, 1)
x = buf[:nfl].view(nchan, nx)
for nh in (16, 127):
    taps = (np.asarray(oracle.TAPS16) if nh == 16 else oracle.windowed_sinc_taps(nh)).astype(np.float32)
    plan = dsp_amd.FirPlan(taps)
    row = []
    for off in range(0, 65 << 20, 4 << 20):
        y = buf[nfl + off // 4: 2 * nfl + off // 4].view(nchan, nx)
        ms = timed(lambda: plan.exec(x, y=y), 10)
        row.append("%d:%.3f" % (off >> 20, ms))
    print("%3d taps (%s), ms by offset in MiB: %s" % (nh, plan.variant, " ".join(row)), flush=True)
