"""Randomised bit-for-bit sweep of the double-precision FIR plans: the kernel without tiles against
the tile kernel (DSPMI_FIR_STREAM=0) on the same buffers, and rows that start from zero state against
the oracle's convolve.  Tap counts 1..40, row lengths 1..6000, channel counts, carried history, row
strides, start alignments and output lengths drawn at random.
usage: python tools/fir_f64_fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsp_amd
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
ncase = 0
names = {}
while time.time() - t0 < budget:
    nh = int(rng.integers(1, 41))
    nx = int(rng.choice([rng.integers(1, 200), rng.integers(200, 2100), rng.integers(2100, 6000)]))
    nchan = int(rng.choice([1, 2, 3, rng.integers(4, 600)]))
    kind = rng.integers(0, 4)
    nxx = [nx, nx, nx + nh, max(1, nx - int(rng.integers(0, 50)))][kind]
    use_hist = bool(rng.integers(0, 2)) and nh > 1
    # mostly the layout the kernel without tiles takes (even strides and starts), sometimes not
    odd = rng.integers(0, 4) == 0
    off_x, off_y = (int(rng.integers(0, 4)), int(rng.integers(0, 4))) if odd else (2 * int(rng.integers(0, 3)), 2 * int(rng.integers(0, 3)))
    pad_x, pad_y = (int(rng.integers(0, 4)), int(rng.integers(0, 4))) if odd else (0, 0)
    sx = off_x + nx + pad_x
    sy = off_y + nxx + pad_y
    if not odd:
        sx, sy = (sx + 1) & ~1, (sy + 1) & ~1
    scale = float(10.0 ** rng.integers(-3, 4))
    h = rng.uniform(-1, 1, nh)
    x = rng.uniform(-1, 1, (nchan, nx)) * scale
    hist = rng.uniform(-1, 1, (nchan, max(nh - 1, 1)))[:, :nh - 1] * scale
    plan = dsp_amd.FirPlan(h, "f64")
    bigx = torch.zeros((nchan, sx), dtype=torch.float64, device="cuda")
    bigx[:, off_x:off_x + nx] = torch.from_numpy(x).cuda()
    xv = bigx[:, off_x:off_x + nx]
    hd = torch.from_numpy(hist).cuda() if use_hist else None
    outs = []
    for stream in (True, False):
        if not stream:
            os.environ["DSPMI_FIR_STREAM"] = "0"
        bigy = torch.full((nchan, sy), 7.0, dtype=torch.float64, device="cuda")
        yv = bigy[:, off_y:off_y + nxx]
        plan.exec(xv, nxx, y=yv, hist=hd)
        torch.cuda.synchronize()
        os.environ.pop("DSPMI_FIR_STREAM", None)
        names[plan.variant] = names.get(plan.variant, 0) + 1
        outs.append(yv.cpu().numpy().copy())
        outside = torch.cat([bigy[:, :off_y].flatten(), bigy[:, off_y + nxx:].flatten()])
        if outside.numel() and not bool((outside == 7.0).all()):
            print("FAIL: wrote outside the rows", dict(variant=plan.variant, nh=nh, nx=nx, nxx=nxx, off_y=off_y), flush=True)
            sys.exit(1)
    case = dict(nh=nh, nx=nx, nxx=nxx, nchan=nchan, hist=use_hist, off_x=off_x, off_y=off_y, sx=sx, sy=sy)
    if not np.array_equal(outs[0], outs[1]):
        print("FAIL: the two kernels differ", case, flush=True)
        sys.exit(1)
    if not use_hist:
        c = int(rng.integers(0, nchan))
        if not np.array_equal(outs[0][c], oracle.convolve(x[c], h, nxx)):
            print("FAIL: differs from the oracle", case, flush=True)
            sys.exit(1)
    ncase += 1
    if ncase % 200 == 0:
        print("%d cases, %.0f s" % (ncase, time.time() - t0), flush=True)
print("ok: %d cases in %.0f s; plan names:" % (ncase, time.time() - t0), names)
