"""Randomised check of the two overlap-save kernels for 32768-sample blocks: tap counts 2..16384, row
lengths, channel counts, history, few or many workgroups (DSPMI_OLS_GRID).  Half the cases run the kernel
with one real block per transform (the default) and are held to the five-kernel composition
(DSPMI_OLS_COMPOSED=1) within the FIR bound; the others run the block-pair kernel (DSPMI_OLS_REAL=0), start-time
spread on and off, with the next pair's input requested early and without (DSPMI_OLS_PREFETCH=0): those
outputs must be equal bit for bit, and every fourth is also held to the composition.
usage: python tools/ols_fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsp_amd
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
ncase = 0
worst = 0.0
rel = {}
while time.time() - t0 < budget:
    real = bool(rng.integers(0, 2))
    nh = int(rng.choice([2049, 4096, 8191, 16384, rng.integers(2049, 16385), rng.integers(2, 2049)]))
    L = 32768 - (nh - 1)
    npairs = int(rng.integers(1, 14))
    nx = int(max(1, 2 * npairs * L - rng.integers(0, 2 * L)))
    nchan = int(rng.choice([1, 2, 3, 5]))
    grid = int(rng.choice([1, 2, 3, 5, 8, 1000]))
    use_hist = bool(rng.integers(0, 2))
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = torch.from_numpy(rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)).cuda()
    hist = torch.from_numpy(rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)).cuda() if use_hist else None
    os.environ["DSPMI_OLS_REAL"] = "1" if real else "0"
    plan = dsp_amd.OlsPlan(h, 32768)
    del os.environ["DSPMI_OLS_REAL"]
    os.environ["DSPMI_OLS_GRID"] = str(grid)
    os.environ["DSPMI_OLS_STAGGER"] = str(int(rng.choice([0, 60000])))
    os.environ.pop("DSPMI_OLS_PREFETCH", None)
    early = plan.exec(x, hist=hist).cpu().numpy()
    os.environ["DSPMI_OLS_PREFETCH"] = "0"
    plain = plan.exec(x, hist=hist).cpu().numpy()
    os.environ.pop("DSPMI_OLS_PREFETCH")
    if not np.array_equal(early, plain):
        bad = np.argwhere(early != plain)
        print("FAIL: early requests change the output", dict(nh=nh, nx=nx, nchan=nchan, grid=grid, hist=use_hist, first=bad[0].tolist(), count=len(bad)), flush=True)
        sys.exit(1)
    if real or ncase % 4 == 0:
        os.environ["DSPMI_OLS_COMPOSED"] = "1"
        ref = dsp_amd.OlsPlan(h, 32768)
        del os.environ["DSPMI_OLS_COMPOSED"]
        want = ref.exec(x, hist=hist).cpu().numpy().astype(np.float64)
        for c in range(nchan):
            for a in range(0, nx, 4096):
                w = want[c, a:a + 4096]
                err = np.linalg.norm(early[c, a:a + 4096] - w)
                # a full block of outputs that each see nh real samples: 1e-6 of the block's own norm, nothing
                # added; blocks holding start-up outputs (no history, first nh-1 of the row) or fewer than 4096
                # outputs: 1e-6 of the larger of their norm and the norm such a block has on average
                steady = len(w) == 4096 and (use_hist or a >= nh - 1)
                lim = 1e-6 * (np.linalg.norm(w) if steady else max(np.linalg.norm(w), np.linalg.norm(h) * np.sqrt(len(w) / 3.0)))
                if np.linalg.norm(w) > 0:
                    k = "steady" if steady else "startup/short"
                    rel[k] = max(rel.get(k, 0.0), err / np.linalg.norm(w))
                if err > lim:
                    print("FAIL: differs from the five-kernel composition", dict(real=real, nh=nh, nx=nx, nchan=nchan, grid=grid, hist=use_hist, chan=c, block=a, err=err, lim=lim), flush=True)
                    sys.exit(1)
                worst = max(worst, err / lim)
    ncase += 1
    if ncase % 25 == 0:
        print("%d cases, %.0f s" % (ncase, time.time() - t0), flush=True)
print("ok: %d cases in %.0f s; worst block difference from the composition %.2f of the bound; relative to the block's own norm:" % (ncase, time.time() - t0, worst),
      {k: "%.2e" % v for k, v in rel.items()})
