"""Fused generator + FIR + spectrum against the two calls: sample agreement through hist_out,
spectra, and timing.  python tools/tone_chain_probe.py [nh]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import dsp_amd
sys.path.insert(0, ".")
import bench

nh = int(sys.argv[1]) if len(sys.argv) > 1 else 127
nchan, nfft, chunk = 1024, 4096, 1 << 18
taps = bench.TAPS16 if nh == 16 else bench.sinc_taps(nh)
freqs = np.array([[100.0 * (c + 1), 12000.0] for c in range(nchan)])
fir, fft = dsp_amd.FirPlan(taps, "f32"), dsp_amd.FftPlan(nfft, "f32")
g1, g2 = dsp_amd.ToneGen(48000.0, freqs, use_sin=True), dsp_amd.ToneGen(48000.0, freqs, use_sin=True)
ps1 = torch.empty((nchan * chunk // nfft, nfft // 2 + 1), dtype=torch.float32, device="cuda")
ps2 = torch.empty_like(ps1)
x = torch.empty((nchan, chunk), dtype=torch.float32, device="cuda")
ha = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")
hb = torch.empty_like(ha)
h2 = ha.clone()
work = torch.empty((dsp_amd.lib().dspmi_tone_fir_spectrum_work_bytes(chunk, nchan, 2),), dtype=torch.uint8, device="cuda")
for step in range(6):
    fir.tone_spectrum(fft, g1, chunk, hist=ha, hist_out=hb, out=ps1, work=work)
    g2.read(chunk, out=x)
    fir.spectrum(fft, x, hist=h2, out=ps2)
    h2 = x[:, chunk - (nh - 1):].contiguous()
    d = (hb - h2).abs()
    e = ((ps1 - ps2).double().norm(dim=1) / ps2.double().norm(dim=1)).max().item()
    hard = work.view(torch.int32)  # flags live inside; count via the library layout is not exposed: skip
    print("chunk %d: w equal %s, hist max diff %.3g, differing %.4f, spectra worst rel %.3g" %
          (step, torch.equal(g1.w, g2.w), d.max().item(), (d != 0).float().mean().item(), e))
    ha, hb = hb, ha

def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def fused():
    global ha, hb
    fir.tone_spectrum(fft, g1, chunk, hist=ha, hist_out=hb, out=ps1, work=work)
    ha, hb = hb, ha

def two():
    g2.read(chunk, out=x)
    fir.spectrum(fft, x, hist=h2, out=ps2)
    h2.copy_(x[:, chunk - (nh - 1):])

print("nh %d: fused %.4f ms per chunk, two calls %.4f ms, filter+spectrum alone %.4f ms" %
      (nh, timeit(fused), timeit(two), timeit(lambda: fir.spectrum(fft, x, hist=h2, out=ps2))))
