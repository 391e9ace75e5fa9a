"""Overlap-save against the direct kernel for short and mid-length filters: time and error."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import dsp_amd, oracle

nchan, nx = 512, 1 << 20
torch.manual_seed(1)
x = (torch.rand((nchan, nx), device="cuda") * 2 - 1)
y = torch.empty_like(x)
for nh in ([int(a) for a in sys.argv[1:]] or (127, 63, 255, 511)):
    taps = oracle.windowed_sinc_taps(nh).astype(np.float32)
    want = None
    xs = x[:2, :70000].cpu().numpy()
    want = np.stack([oracle.fir_batch_f32(xs[c:c + 1], taps, 70000)[0] for c in range(2)]) if hasattr(oracle, "fir_batch_f32") else None
    os.environ["DSPMI_FIR_VARIANT"] = "mfma"
    d = dsp_amd.FirPlan(taps)
    del os.environ["DSPMI_FIR_VARIANT"]
    for n in (0, 512, 1024, 2048):
        try:
            plan = d if n == 0 else dsp_amd.OlsPlan(taps, fft_n=n)
        except Exception as e:
            print(nh, n, "unavailable", e); continue
        run = (lambda: plan.exec(x, y=y)) if n else (lambda: plan.exec(x, nx, y=y))
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        got = y[:2, :70000].cpu().numpy().astype(np.float64)
        err = ""
        if want is not None:
            w = np.asarray(want, dtype=np.float64)
            e = [np.linalg.norm(got[c, b:b + 4096] - w[c, b:b + 4096]) / np.linalg.norm(w[c, b:b + 4096]) for c in range(2) for b in range(0, 65536, 4096)]
            err = "max rel L2 per 4096-block %.2e" % max(e)
        print("nh %4d  %-10s %7.3f ms  %7.1f Gsamples/s  %6.0f GB/s  %s" % (nh, "direct" if n == 0 else "ols N=%d" % n, ms, nchan * nx / ms / 1e6, 8 * nchan * nx / ms / 1e6, err), flush=True)
