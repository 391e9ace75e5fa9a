"""Direct MFMA FIR by tap count (512 ch x 2^20): time per launch; DSPMI_FIR_TPB / DSPMI_FIR_BPC apply."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DSPMI_FIR_VARIANT"] = "mfma"
import dsp_amd  # noqa: E402
import oracle  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

x = torch.rand(512, 1 << 20, device="cuda") - 0.5
y = torch.empty_like(x)
for nh in (8, 16, 32, 48, 63, 64, 127):
    plan = dsp_amd.FirPlan(oracle.windowed_sinc_taps(nh).astype(np.float32))
    ms = timed(lambda: plan.exec(x, y=y), 10)
    print("taps %3d: %.4f ms  %.1f Gsamples/s  %.2f TB/s  %.1f TFLOP/s" % (nh, ms, 512 * 1048576 / ms / 1e6, 512 * 1048576 * 8 / ms / 1e9, 512 * 1048576 * 2 * nh / ms / 1e9), flush=True)
