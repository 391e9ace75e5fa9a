"""c2c FFT throughput when the batch stays in the last-level cache: an estimate of the kernel's
own (LDS / ALU / latency) limit, apart from HBM."""
import sys, time
import torch
sys.path.insert(0, ".")
import dsp_amd
for n in (1024, 4096):
    plan = dsp_amd.FftPlan(n, "f32")
    for frames in (2048, 4096, 8192, 16384, 65536):
        frames = frames * 1024 // n
        x = torch.randn((frames, n, 2), device="cuda").view(torch.float32)
        x = torch.view_as_complex(x.view(frames, n, 2))
        y = torch.empty_like(x)
        for inplace in (False, True):
            out = x if inplace else y
            for _ in range(3): plan.c2c(x, out=out)
            torch.cuda.synchronize()
            reps = 50
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps): plan.c2c(x, out=out)
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / reps * 1e3
            print("n %5d frames %6d %s: %8.1f us  %6.2f ns/frame  %7.1f GB/s-equivalent" % (n, frames, "in place " if inplace else "out of pl", us, us * 1e3 / frames, 16.0 * n * frames / us / 1e3), flush=True)
