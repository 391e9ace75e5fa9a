"""c2c N=1024 x 2^20 frames with the output buffer at different offsets from the input buffer:
does the rate depend on how the two streams fall onto the memory channels?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

n, batch = 1024, 1 << 20
nbytes = batch * n * 8
buf = torch.empty((2 * nbytes + (160 << 20)) // 4, dtype=torch.float32, device="cuda")
buf.uniform_(-1, 1)
plan = dsp_amd.FftPlan(n, "f32")
zin = torch.view_as_complex(buf[: nbytes // 4].view(batch, n, 2))
print("base address %#x" % buf.data_ptr(), flush=True)
offs = [int(a, 0) for a in sys.argv[1:]] or (list(range(0, 65 << 20, 1 << 20)) +
                                             [(16 << 20) + d for d in (-(1 << 19), -65536, -4096, 4096, 65536, 1 << 19)])
base = (int(os.environ.get("FFT_BASE_SHIFT", "0"), 0) // 4)
zin = torch.view_as_complex(buf[base: base + nbytes // 4].view(batch, n, 2))
for off in offs:
    o = buf[base + (nbytes + off) // 4: base + (2 * nbytes + off) // 4]
    zout = torch.view_as_complex(o.view(batch, n, 2))
    ms = timed(lambda: plan.c2c(zin, out=zout), 10)
    print("offset %9d: %.3f ms  %.1f GB/s" % (off, ms, 2 * nbytes / ms / 1e6), flush=True)
