"""Time one overlap-save configuration: tools/ols_one.py <taps> <fft length> [channels] [samples]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import dsp_amd, oracle
nh, n = int(sys.argv[1]), int(sys.argv[2])
nchan = int(sys.argv[3]) if len(sys.argv) > 3 else 64
nx = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 22
taps = oracle.windowed_sinc_taps(nh).astype(np.float32)
x = torch.rand((nchan, nx), device="cuda") * 2 - 1
y = torch.empty_like(x)
plan = dsp_amd.OlsPlan(taps, fft_n=n)
for _ in range(3): plan.exec(x, y=y)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): plan.exec(x, y=y)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
seg = x[0, :40000].cpu().numpy()
want = oracle.fir_batch_f32(seg[None, :], taps, 40000)[0]
got = y[0, :40000].cpu().numpy()
print("taps %5d N %5d: %7.3f ms %7.1f Gsamples/s  rel L2 %.2e" % (nh, n, ms, nchan * nx / ms / 1e6, np.linalg.norm(got - want) / np.linalg.norm(want)), flush=True)
