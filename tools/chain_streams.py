"""The chain (tones -> 127-tap FIR + 4096-point spectrum) with the generator of chunk k+1 on a
second stream beside the filter + spectrum kernel of chunk k (two sample buffers), against the
same launches in one stream.  usage: python tools/chain_streams.py [priority of the generator's stream: 0 or -1] [chunks]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def main():
    prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nchan, nfft, nh = 1024, 4096, 127
    nchunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    chunk = (1 << 20) // nchunk
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    freqs = np.array([[100.0 * (c + 1), 12000.0] for c in range(nchan)])
    gen = dsp_amd.ToneGen(48000.0, freqs, use_sin=True)
    fir = dsp_amd.FirPlan(h, "f32")
    fft = dsp_amd.FftPlan(nfft, "f32")
    xs = [torch.empty((nchan, chunk), dtype=torch.float32, device="cuda") for _ in range(2)]
    ps = torch.empty((nchan * chunk // nfft, nfft // 2 + 1), dtype=torch.float32, device="cuda")
    hist = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")
    s_gen = torch.cuda.Stream(priority=prio)
    s_fir = torch.cuda.Stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one_stream(steps):
        for _ in range(steps * nchunk):
            gen.read(chunk, out=xs[0])
            fir.spectrum(fft, xs[0], hist=hist, out=ps)
            hist.copy_(xs[0][:, chunk - (nh - 1):])

    def two_streams(steps):
        made = [torch.cuda.Event() for _ in range(2)]
        used = [torch.cuda.Event() for _ in range(2)]
        for k in range(steps * nchunk):
            b = k & 1
            with torch.cuda.stream(s_gen):
                if k >= 2:
                    s_gen.wait_event(used[b])
                gen.read(chunk, out=xs[b])
                made[b].record(s_gen)
            with torch.cuda.stream(s_fir):
                s_fir.wait_event(made[b])
                fir.spectrum(fft, xs[b], hist=hist, out=ps)
                hist.copy_(xs[b][:, chunk - (nh - 1):])
                used[b].record(s_fir)

    for name, fn in (("one stream", one_stream), ("two streams", two_streams), ("one stream", one_stream), ("two streams", two_streams)):
        fn(6)
        torch.cuda.synchronize()
        ev0.record()
        fn(10)
        s_gen.synchronize()
        s_fir.synchronize()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 10
        print("%s (generator priority %d, %d chunks): %.3f ms per step, %.1f Gsamples/s" % (name, prio, nchunk, ms, nchan * (1 << 20) / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
