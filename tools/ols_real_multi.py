"""Both 32768-sample-block overlap-save kernels on multi-channel launches (rotating buffers)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def main():
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for nh, nchan, nx in [(8191, 1, 1 << 26), (8191, 64, 1 << 20), (8191, 512, 1 << 20), (2049, 512, 1 << 20), (4096, 64, 1 << 22)]:
        h = oracle.windowed_sinc_taps(nh).astype(np.float32)
        nbuf = max(2, (3 << 30) // (nchan * nx * 8))
        nbuf = min(nbuf, 6)
        xs = [torch.rand((nchan, nx), device="cuda") * 2 - 1 for _ in range(nbuf)]
        ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
        for rep in range(2):
            for real in ("1", "0"):
                os.environ["DSPMI_OLS_REAL"] = real
                pl = dsp_amd.OlsPlan(h, 32768)
                n = 40 if nchan * nx <= (1 << 26) else 8
                for i in range(n):
                    pl.exec(xs[i % nbuf], y=ys[i % nbuf])
                torch.cuda.synchronize()
                ev0.record()
                for i in range(n):
                    pl.exec(xs[i % nbuf], y=ys[i % nbuf])
                ev1.record()
                torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1) / n
                print("nh %5d  %4d ch x %9d  %s: %.4f ms  %.1f Gsamples/s" % (nh, nchan, nx, "real blocks" if real == "1" else "block pairs",
                                                                            ms, nchan * nx / ms / 1e6), flush=True)
                pl.close()
        del xs, ys
    os.environ.pop("DSPMI_OLS_REAL")


if __name__ == "__main__":
    main()
