"""Randomised cross-check of the transform plans against an independent implementation: torch.fft (rocFFT) in
double precision on the same device.  Lengths: powers of two up to 2^22, and arbitrary lengths (primes and
composites, the chirp-z path) up to 200000; f32 and f64 plans; c2c both ways, r2c, c2r; random batch counts.
The bound is a relative L2 error per call of 1e-5 (f32) / 1e-12 (f64) -- the tests' bounds.
usage: python tools/fft_fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsp_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
t0 = time.time()
ncase = 0
worst = {"f32": 0.0, "f64": 0.0}
plans = {}
while time.time() - t0 < budget:
    kind = rng.integers(0, 3)
    if kind == 0:
        n = 1 << int(rng.integers(1, 23))
    elif kind == 1:
        n = int(rng.integers(2, 5000))
    else:
        n = int(rng.integers(5000, 200000))
    dtype = "f64" if rng.integers(0, 3) == 0 else "f32"
    batch = int(max(1, min(rng.integers(1, 400), (1 << 22) // n)))
    ct, rt = (torch.complex128, torch.float64) if dtype == "f64" else (torch.complex64, torch.float32)
    key = (n, dtype)
    if key not in plans:
        if len(plans) > 40:
            plans.clear()
        plans[key] = dsp_amd.FftPlan(n, dtype)
    plan = plans[key]
    op = ["fwd", "bwd", "r2c", "c2r"][int(rng.integers(0, 4))]
    if op in ("fwd", "bwd"):
        x = torch.view_as_complex(torch.rand((batch, n, 2), generator=g, device="cuda", dtype=rt) * 2 - 1)
        got = plan.c2c(x, dsp_amd.FORWARD if op == "fwd" else dsp_amd.BACKWARD)
        x128 = x.to(torch.complex128)
        want = torch.fft.fft(x128, dim=-1) if op == "fwd" else torch.fft.ifft(x128, dim=-1, norm="forward")
    elif op == "r2c":
        x = torch.rand((batch, n), generator=g, device="cuda", dtype=rt) * 2 - 1
        got = plan.r2c(x)
        want = torch.fft.rfft(x.to(torch.float64), dim=-1)
    else:
        x = torch.view_as_complex(torch.rand((batch, n // 2 + 1, 2), generator=g, device="cuda", dtype=rt) * 2 - 1)
        x[:, 0] = x[:, 0].real.to(ct)                  # a real signal's spectrum: DC (and Nyquist) real
        if n % 2 == 0:
            x[:, -1] = x[:, -1].real.to(ct)
        want = torch.fft.irfft(x.to(torch.complex128), n=n, dim=-1, norm="forward")
        got = plan.c2r(x.clone())
    err = float(torch.linalg.vector_norm(got.to(want.dtype) - want) / torch.linalg.vector_norm(want))
    lim = 1e-12 if dtype == "f64" else 1e-5
    if not err <= lim:
        print("FAIL", dict(n=n, dtype=dtype, op=op, batch=batch, err=err, variant=plan.variant), flush=True)
        sys.exit(1)
    worst[dtype] = max(worst[dtype], err)
    ncase += 1
    if ncase % 100 == 0:
        print("%d cases, %.0f s, worst %s" % (ncase, time.time() - t0, {k: "%.1e" % v for k, v in worst.items()}), flush=True)
print("ok: %d cases in %.0f s; worst relative L2 error %s" % (ncase, time.time() - t0, {k: "%.2e" % v for k, v in worst.items()}))
