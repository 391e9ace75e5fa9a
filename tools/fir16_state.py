"""Why does the 16-tap FIR run 0.77 ms after the FFT workload in one process and 0.73 ms on its own?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

plan = dsp_amd.FirPlan(np.asarray(oracle.TAPS16, dtype=np.float32))


def fir(tag):
    both = torch.empty((2, 512, 1 << 20), dtype=torch.float32, device="cuda")
    both[0].uniform_(-1, 1)
    ms = timed(lambda: plan.exec(both[0], y=both[1]), 10)
    print("%-46s %#x  %.4f ms" % (tag, both.data_ptr(), ms), flush=True)
    del both
    torch.cuda.empty_cache()


fir("fresh process")
fir("again")
big = torch.empty(16 << 30, dtype=torch.uint8, device="cuda")
fir("while 16 GiB more are allocated")
del big
torch.cuda.empty_cache()
fir("after freeing them")
fplan = dsp_amd.FftPlan(1024, "f32")
z = torch.view_as_complex(torch.empty((2, 1 << 20, 1024, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1))
for _ in range(20):
    fplan.c2c(z[0], out=z[1])
torch.cuda.synchronize()
fir("while the FFT buffers are allocated, after 20 FFTs")
del z
torch.cuda.empty_cache()
fir("after freeing the FFT buffers")
