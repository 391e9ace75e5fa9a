"""The one-real-block-per-transform overlap-save kernel (DSPMI_OLS_REAL, the default for 32768-sample
blocks) against the kernel that runs block pairs (DSPMI_OLS_REAL=0) and against the oracle, then
both timed on the 2^26-sample, 8191-tap workload."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402


def plans(h):
    os.environ["DSPMI_OLS_REAL"] = "1"
    new = dsp_amd.OlsPlan(h, 32768)
    os.environ["DSPMI_OLS_REAL"] = "0"
    old = dsp_amd.OlsPlan(h, 32768)
    os.environ.pop("DSPMI_OLS_REAL")
    return new, old


def main():
    rng = np.random.default_rng(7)
    bad = 0
    for nh, nx, nchan, with_hist in [(8191, 150000, 2, False), (8191, 70001, 3, True), (2049, 100000, 2, True), (4096, 90000, 2, True),
                                     (16, 40000, 2, False), (16384, 120001, 1, True), (8191, 1, 1, False), (5000, 24577 * 3, 2, False)]:
        h = oracle.windowed_sinc_taps(nh).astype(np.float32)
        x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
        hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32) if with_hist else None
        new, old = plans(h)
        xd = torch.from_numpy(x).cuda()
        hd = torch.from_numpy(hist).cuda() if with_hist else None
        a = new.exec(xd, hist=hd).cpu().numpy()
        b = old.exec(xd, hist=hd).cpu().numpy()
        # (against the norm steady-state output has: a row shorter than the filter is start-up transient only)
        den = max(np.linalg.norm(b.astype(np.float64)), np.linalg.norm(h.astype(np.float64)) * np.linalg.norm(x.astype(np.float64)))
        diff = np.linalg.norm(a.astype(np.float64) - b) / den
        msg = "nh %5d nx %7d nchan %d hist %d: new vs pairs %.2e" % (nh, nx, nchan, with_hist, diff)
        if nh <= 8191 and nx <= 150000:
            full = np.concatenate([hist, x], axis=1) if with_hist else x
            want = oracle.fir_batch_f32(full, h, full.shape[1])[:, full.shape[1] - nx:]
            worst_new = worst_old = 0.0
            for c in range(nchan):
                for lo in range(nh - 1 if not with_hist else 0, nx, 4096):
                    w = want[c, lo:lo + 4096]
                    n = np.linalg.norm(w)
                    if n == 0:
                        continue
                    worst_new = max(worst_new, np.linalg.norm(a[c, lo:lo + 4096] - w) / n)
                    worst_old = max(worst_old, np.linalg.norm(b[c, lo:lo + 4096] - w) / n)
            msg += "  worst block vs oracle: new %.2e pairs %.2e" % (worst_new, worst_old)
            if worst_new > 1e-6:
                bad += 1
        if not diff < 1e-5:
            bad += 1
        print(msg, flush=True)
    if bad:
        print("FAILED", bad)
        return 1
    nh, nx = 8191, 1 << 26
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    new, old = plans(h)
    xd = torch.rand((1, nx), device="cuda") * 2 - 1
    y = torch.empty_like(xd)
    for rep in range(3):
        for name, pl in (("real blocks", new), ("block pairs", old)):
            for _ in range(3):
                pl.exec(xd, y=y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                pl.exec(xd, y=y)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / 20
            print("%s: %.4f ms  %.1f Gsamples/s" % (name, ms, nx / ms / 1e6), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
