"""r2c / c2r (n = 2048, 2^28 samples) with the output buffer at different offsets behind the input."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tools.c2r_probe import timed  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = (1 << 28) // n
nb = n // 2 + 1
nx, nbins = batch * n, batch * nb * 2          # floats
buf = torch.empty(nx + nbins + (80 << 20) // 4, dtype=torch.float32, device="cuda")
buf.uniform_(-1, 1)
plan = dsp_amd.FftPlan(n, "f32")
x = buf[:nx].view(batch, n)
row_f, row_b = [], []
for off in range(0, 65 << 20, 4 << 20):
    start = (nx * 4 + off + 15) // 16 * 4
    bins = torch.view_as_complex(buf[start: start + nbins].view(batch, nb, 2))
    row_f.append("%d:%.3f" % (off >> 20, timed(lambda: plan.r2c(x, out=bins), 10)))
for off in range(0, 65 << 20, 4 << 20):
    start = (nx * 4 + off + 15) // 16 * 4
    bins = torch.view_as_complex(buf[start: start + nbins].view(batch, nb, 2))
    row_b.append("%d:%.3f" % (off >> 20, timed(lambda: plan.c2r(bins, out=x), 10)))
print("n=%d r2c ms by offset (MiB): %s" % (n, " ".join(row_f)))
print("n=%d c2r ms by offset (MiB): %s" % (n, " ".join(row_b)))
