"""Relative L2 error per 4096-sample output block of the long-filter f32 paths against the
double-accumulating checker: the direct MFMA kernel (pinned) and overlap-save, windowed-sinc
taps and uniform input (the bench inputs).  Prints one line per (path, nh)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd
import oracle


def block_errors(got, want, blk=4096):
    out = []
    for lo in range(0, len(want), blk):
        w = want[lo:lo + blk]
        out.append(float(np.linalg.norm(got[lo:lo + blk] - w) / np.linalg.norm(w)))
    return out


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (1, nx)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    rows = []
    for nh in (127, 511, 2049, 4096, 4097, 8191):
        h = oracle.windowed_sinc_taps(nh).astype(np.float32)
        want = oracle.fir_batch_f32(x, h, nx)[0]
        os.environ["DSPMI_FIR_VARIANT"] = "mfma"
        direct = dsp_amd.FirPlan(h, "f32")
        del os.environ["DSPMI_FIR_VARIANT"]
        paths = [("direct:" + direct.variant, direct.exec(xd).cpu().numpy()[0])]
        for n in sorted({0, 16384, 32768}):
            if n and n < 2 * nh:
                continue
            try:
                ols = dsp_amd.OlsPlan(h, n)
            except dsp_amd.DspmiError:
                continue
            paths.append(("ols_n%d" % ols.fft_length, ols.exec(xd).cpu().numpy()[0]))
        for name, got in paths:
            e = block_errors(got, want)
            first = (nh + 4095) // 4096      # blocks that contain start-up outputs
            inner = e[first:] or [float("nan")]
            row = dict(path=name, nh=nh, nx=nx, startup_blocks_max=max(e[:first]), interior_max=max(inner),
                       interior_median=float(np.median(inner)))
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    main()
