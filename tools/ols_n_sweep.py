"""Overlap-save by transform length for mid-range tap counts, 512 channels x 2^20 samples (two rotating buffer pairs)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def main():
    nchan, nx = 512, 1 << 20
    xs = [torch.rand((nchan, nx), device="cuda") * 2 - 1 for _ in range(2)]
    ys = [torch.empty_like(xs[0]) for _ in range(2)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for nh in (255, 306, 400, 511, 768, 1024, 1536, 2048):
        h = oracle.windowed_sinc_taps(nh).astype(np.float32)
        row = []
        for n in (1024, 2048, 4096, 8192, 16384, 32768):
            if n < 2 * nh:
                row.append("      -")
                continue
            pl = dsp_amd.OlsPlan(h, n)
            for i in range(6):
                pl.exec(xs[i % 2], y=ys[i % 2])
            torch.cuda.synchronize()
            ev0.record()
            for i in range(10):
                pl.exec(xs[i % 2], y=ys[i % 2])
            ev1.record()
            torch.cuda.synchronize()
            row.append("%7.3f" % (ev0.elapsed_time(ev1) / 10))
            pl.close()
        print("nh %5d  ms by N (1024 .. 32768): %s" % (nh, " ".join(row)), flush=True)


if __name__ == "__main__":
    main()
