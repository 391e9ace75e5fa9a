"""Times the 32768-sample-block overlap-save kernels on the 2^26-sample, 8191-tap workload under the
environment given on the command line: tools/ols_real_sweep.py VAR=a,b,c [VAR2=...] (one run per combination)."""
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def main():
    axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
    nh, nx = 8191, 1 << 26
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    nbuf = int(os.environ.get("NBUF", "1"))   # rotate over this many input / output pairs (beyond the last-level cache)
    xs = [torch.rand((1, nx), device="cuda") * 2 - 1 for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        for combo in itertools.product(*[v for _, v in axes]):
            for (k, _), v in zip(axes, combo):
                os.environ[k] = v
            pl = dsp_amd.OlsPlan(h, 32768)
            for i in range(80):
                pl.exec(xs[i % nbuf], y=ys[i % nbuf])
            torch.cuda.synchronize()
            ev0.record()
            for i in range(60):
                pl.exec(xs[i % nbuf], y=ys[i % nbuf])
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 60
            print(" ".join("%s=%s" % (k, v) for (k, _), v in zip(axes, combo)), "%.4f ms  %.1f Gsamples/s" % (ms, nx / ms / 1e6), flush=True)
            pl.close()


if __name__ == "__main__":
    main()
