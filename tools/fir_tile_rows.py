"""The f16 FIR kernel's tile size by row length: 2^27 samples as rows of 256 ... 16384 samples, tiles of
4096 / 2048 / 1024 outputs pinned (DSPMI_FIR_TILE) and the plan's own choice; Gsamples/s.
usage: python tools/fir_tile_rows.py [taps...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
import oracle   # noqa: E402

total = 1 << 27
rows = (256, 512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192, 12288, 16384, 1 << 20)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("rows of            " + " ".join("%6d" % r for r in rows))
for nh in [int(a) for a in sys.argv[1:]] or [33, 127, 305]:
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    for tile in ("4096", "2048", "1024", "auto"):
        if tile == "auto":
            os.environ.pop("DSPMI_FIR_TILE", None)
        else:
            os.environ["DSPMI_FIR_TILE"] = tile
        out = []
        for nx in rows:
            nchan = total // nx
            x = torch.rand((nchan, nx), device="cuda") * 2 - 1
            y = torch.empty_like(x)
            for _ in range(20):
                plan.exec(x, nx, y=y)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(10):
                plan.exec(x, nx, y=y)
            ev1.record()
            torch.cuda.synchronize()
            out.append("%6.0f" % (nchan * nx / (ev0.elapsed_time(ev1) / 10) / 1e6))
        print("nh %4d tile %5s  %s" % (nh, tile, " ".join(out)), flush=True)
