import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd, oracle
h = np.asarray(oracle.TAPS16, dtype=np.float32)
plan = dsp_amd.FirPlan(h, "f32")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for total in (1 << 27, 1 << 29):
    for nx in (1024, 16384, 1 << 18, 1 << 20):
        nchan = total // nx
        x = torch.rand((nchan, nx), device="cuda") * 2 - 1
        y = torch.empty_like(x)
        out = []
        for v in ("1", "0", "1", "0"):
            os.environ["DSPMI_FIR_STREAM"] = v
            for _ in range(30): plan.exec(x, nx, y=y)
            torch.cuda.synchronize(); ev0.record()
            for _ in range(20): plan.exec(x, nx, y=y)
            ev1.record(); torch.cuda.synchronize()
            out.append("%s %.0f" % ("stream" if v == "1" else "tile", total / (ev0.elapsed_time(ev1) / 20) / 1e6))
        print("total 2^%d rows of %7d: %s" % (total.bit_length() - 1, nx, "  ".join(out)), flush=True)
        del x, y
