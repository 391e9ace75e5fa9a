"""Tone generator timing: tools/tone_probe.py  (1024 channels x 262144 samples per call, two sine tones)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import dsp_amd
nchan, n = 1024, 1 << 18
for label, freqs in (("bench set", np.array([[100.0 * (c + 1), 12000.0] for c in range(nchan)])),
                     ("low", np.array([[100.0, 440.0]] * nchan)),
                     ("one tone", np.array([[1000.0]] * nchan))):
    gen = dsp_amd.ToneGen(48000.0, freqs, use_sin=True)
    x = torch.empty((nchan, n), dtype=torch.float32, device="cuda")
    for call in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gen.read(n, out=x)
        torch.cuda.synchronize()
        print("%-9s call %d: %7.3f ms" % (label, call, (time.perf_counter() - t0) * 1e3), flush=True)
