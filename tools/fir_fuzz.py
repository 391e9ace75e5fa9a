"""Randomised parity sweep of the f32 FIR paths against the oracle: tap counts, row lengths,
output lengths, channel counts, history, row strides and start alignments drawn at random, every
kernel family pinned in turn.  usage: python tools/fir_fuzz.py [seconds] [seed]"""
import os
import re
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
TOL = 1e-6
t0 = time.time()
ncase = 0
worst = {}
nblocks = {}
while time.time() - t0 < budget:
    variant = rng.choice(["f16", "bf16", "mfma", "ols", "auto"])
    nh = int(rng.choice([rng.integers(1, 33), rng.integers(33, 306), rng.integers(306, 1200)], p=[0.2, 0.6, 0.2]))
    if variant in ("f16", "bf16") and nh > 305:
        nh = int(rng.integers(1, 306))
    nx = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 70000)]))
    nchan = int(rng.choice([1, 2, 3, rng.integers(4, 40)]))
    kind = rng.integers(0, 4)
    nxx = [nx, nx + nh, nx + nh - 1, max(1, nx - int(rng.integers(0, 50)))][kind]
    use_hist = bool(rng.integers(0, 2))
    pad_x, pad_y = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    off_x, off_y = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    h = (oracle.windowed_sinc_taps(nh) if nh >= 9 and rng.integers(0, 2) else rng.uniform(-1, 1, nh)).astype(np.float32)
    scale = float(10.0 ** rng.integers(-3, 4))
    x = (rng.uniform(-1, 1, (nchan, nx)) * scale).astype(np.float32)
    hist = (rng.uniform(-1, 1, (nchan, max(nh - 1, 1))) * scale).astype(np.float32)[:, :nh - 1]
    if variant == "auto":
        os.environ.pop("DSPMI_FIR_VARIANT", None)
    else:
        os.environ["DSPMI_FIR_VARIANT"] = variant
    plan = dsp_amd.FirPlan(h, "f32")
    os.environ.pop("DSPMI_FIR_VARIANT", None)
    bigx = torch.zeros((nchan, off_x + nx + pad_x), device="cuda")
    bigx[:, off_x:off_x + nx] = torch.from_numpy(x).cuda()
    bigy = torch.full((nchan, off_y + nxx + pad_y), 7.0, device="cuda")
    xv, yv = bigx[:, off_x:off_x + nx], bigy[:, off_y:off_y + nxx]
    hd = torch.from_numpy(hist).cuda() if use_hist and nh > 1 else None
    plan.exec(xv, nxx, y=yv, hist=hd)
    torch.cuda.synchronize()
    got = yv.cpu().numpy()
    full = np.concatenate([hist if hd is not None else np.zeros((nchan, nh - 1), np.float32), x], axis=1)
    want = oracle.fir_batch_f32(full, h, nh - 1 + nxx)[:, nh - 1:]
    floor = 2.0 ** -37 * np.sqrt(nh) * np.abs(h).max() * max(np.abs(full).max(), 1e-300)
    tag = re.sub(r"(_s\d+|_n\d+|\d+)$", "", plan.variant)   # kernel family: without slice count / transform length
    f16 = "f16x2" in plan.variant
    for c in range(nchan):
        edges = list(range(0, nxx, 4096)) + [nxx]
        if len(edges) > 2 and edges[-1] - edges[-2] < 1024:
            del edges[-2]
        for a, b in zip(edges, edges[1:]):
            err = np.linalg.norm(got[c, a:b].astype(np.float64) - want[c, a:b])
            ref = np.linalg.norm(want[c, a:b])
            # Three classes of block, each with its own bound:
            #  steady  a full 4096-output block of outputs that each see nh real samples (carried history, or
            #          past the first nh-1 outputs; not reaching past the input's end): 1e-6 of the block's OWN
            #          norm, nothing added -- except the f16 kernel's absolute floor, which is its contract;
            #  startup a block holding some of the first nh-1 outputs of a row without history (sums over the
            #          filter's outer taps only: a transform's rounding error is relative to what goes through
            #          it, not to these);
            #  short   fewer than 4096 outputs, or outputs past the input's end (the nx + nh tail).
            # The last two are held to 1e-6 of the larger of the block's norm and the norm a block of these
            # taps and samples has on average, plus the floor.
            startup = hd is None and a < nh - 1
            steady = (b - a == 4096) and not startup and b <= nx
            kind_name = "steady" if steady else "startup" if startup else "short"
            typical = np.linalg.norm(h) * np.sqrt(np.mean(full[c].astype(np.float64) ** 2) * (b - a))
            if steady:
                lim = TOL * ref + (floor * np.sqrt(b - a) if f16 else 0.0)
            else:
                lim = TOL * max(ref, typical) + floor * np.sqrt(b - a)
                # 1e-6 is a figure for the root mean square over a block: a block of a few outputs has no mean
                # to speak of (one output of an 850-term f32 sum lands at 1.3e-6 about once in 10^4 cases)
                if b - a < 64:
                    lim *= 4.0
            case = dict(variant=plan.variant, nh=nh, nx=nx, nxx=nxx, nchan=nchan, hist=hd is not None, off_x=off_x, off_y=off_y,
                        chan=c, block=a, len=b - a, err=float("%.3g" % err), ref=float("%.3g" % ref), scale=scale)
            if err > lim:
                print("FAIL (%s block)" % kind_name, case, flush=True)
                sys.exit(1)
            if ref > 0:
                key = (tag, kind_name)
                if err / ref > worst.get(key, (0.0, None))[0]:
                    worst[key] = (err / ref, case)
                nblocks[key] = nblocks.get(key, 0) + 1
    outside = torch.cat([bigy[:, :off_y].flatten(), bigy[:, off_y + nxx:].flatten()])
    if outside.numel() and not bool((outside == 7.0).all()):
        print("FAIL: wrote outside the rows", dict(variant=plan.variant, nh=nh, nx=nx, nxx=nxx, off_y=off_y), flush=True)
        sys.exit(1)
    ncase += 1
    if ncase % 200 == 0:
        print("%d cases, %.0f s" % (ncase, time.time() - t0), flush=True)
print("ok: %d cases in %.0f s.  Worst error of a block relative to the block's own norm, by kernel and class of block" % (ncase, time.time() - t0))
print("(steady: asserted <= 1e-6 with nothing added, the f16 kernel's floor of 2^-37 sqrt(nh) max|h| max|x| aside; startup / short: see the source)")
for (tag, kind_name), (e, case) in sorted(worst.items()):
    print("  %-32s %-8s %8d blocks  worst %.2e  in %s" % (tag, kind_name, nblocks[(tag, kind_name)], e, case))
