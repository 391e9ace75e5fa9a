"""Batched f32 FIR by kernel family and tap count: time and error of the bf16 split kernel, the
f32 matrix kernel and overlap-save on the bench's shape (channels x 2^20 samples).
usage: python tools/fir_variants_probe.py [channels] [taps...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsp_amd
import oracle

nchan = int(sys.argv[1]) if len(sys.argv) > 1 else 512
taps_list = [int(a) for a in sys.argv[2:]] or [16, 17, 31, 63, 64, 127, 191, 255, 305]
nx = 1 << 20
g = torch.Generator(device="cuda").manual_seed(7)
both = torch.empty((2, nchan, nx), dtype=torch.float32, device="cuda")
both[0].uniform_(-1.0, 1.0, generator=g)
x, y = both[0], both[1]


def block_err(yrow, xrow, h):
    """worst relative L2 error over the 4096-sample blocks of a 65536-sample head of one row"""
    n = 1 << 16
    want = oracle.fir_batch_f32(xrow[None, :n], h, n)[0]
    got = yrow[:n]
    worst = 0.0
    for b in range(0, n, 4096):
        worst = max(worst, float(np.linalg.norm(got[b:b + 4096] - want[b:b + 4096]) / np.linalg.norm(want[b:b + 4096])))
    return worst


for nh in taps_list:
    h = (np.asarray(oracle.TAPS16) if nh == 16 else oracle.windowed_sinc_taps(nh)).astype(np.float32)
    line = "taps %4d:" % nh
    for variant in ("f16", "bf16", "mfma", "ols"):
        os.environ["DSPMI_FIR_VARIANT"] = variant
        plan = dsp_amd.FirPlan(h, "f32")
        del os.environ["DSPMI_FIR_VARIANT"]
        name = plan.variant
        if variant in ("bf16", "f16") and variant not in name or variant == "ols" and "overlap" not in name:
            continue
        y.zero_()
        for _ in range(3):
            plan.exec(x, nx, y=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            plan.exec(x, nx, y=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        err = block_err(y[nchan // 2].cpu().numpy(), x[nchan // 2].cpu().numpy(), h)
        line += "  %s %.3f ms %6.1f Gs/s err %.2e |" % (variant, ms, nchan * nx / ms / 1e6, err)
    print(line, flush=True)
