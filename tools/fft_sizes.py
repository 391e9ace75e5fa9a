#!/usr/bin/env python3
"""GB/s of the batched c2c FFT at each supported power-of-two length (1 GiB of frames in, 1 GiB out):
   tools/fft_sizes.py [f64] [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dsp_amd

args = sys.argv[1:]
dtype = "f32"
if args and args[0] in ("f32", "f64"):
    dtype = args.pop(0)
esz = 8 if dtype == "f32" else 16
for n in [int(a) for a in args] or [64, 256, 1024, 2048, 4096, 8192, 16384]:
    batch = (1 << 30) // (n * esz)
    z = torch.view_as_complex(torch.rand((batch, n, 2), device="cuda", dtype=torch.float32 if dtype == "f32" else torch.float64) * 2 - 1)
    out = torch.empty_like(z)
    plan = dsp_amd.FftPlan(n, dtype)
    for _ in range(2):
        plan.c2c(z, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.c2c(z, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("n=%6d  %-34s %7.3f ms  %7.1f GB/s" % (n, plan.variant, ms, 2 * batch * n * esz / ms / 1e6), flush=True)
