"""Generates the golden vectors under tests/golden/.

Run in the build container (where /root/reference is mounted and oracle/_ref
has been built by `make -C oracle`):  python tests/golden/make_golden.py

Sources of truth:
  * conv_*  : outputs of the reference's OWN conv0/conv1 (convolution/main.c
              compiled in place into oracle/_ref/libref_conv.so) and the table
              printed by the reference program oracle/_ref/convolution_main.
  * fft_*   : the reference delegates to FFTW3, which is absent here, so these
              are "parity unpinned" vectors: a long-double O(n^2) DFT of seeded
              inputs, cross-checked against numpy's pocketfft when generated.
  * tone_*  : the reference's tone generators restated in the oracle
              (fft/src/main.c:265-291, tonegen/tonegen.c:10-59) and pushed
              through the oracle's `convolve` with the reference's tap tables.
Only data is stored: inputs and expected outputs.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402


def main():
    assert oracle.ref_lib() is not None, "build oracle/_ref first (make -C oracle)"

    # --- the reference program's own printed table ---------------------------
    exe = os.path.join(HERE, "..", "..", "oracle", "_ref", "convolution_main")
    rows = [ln.split() for ln in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()]
    kat = {
        "source": "stdout of /root/reference/convolution/main.c built as convolution/do_build.sh:2 does",
        "x": [float(r[1]) for r in rows],
        "h": oracle.KAT9_H.tolist(),
        "y0_printed": [float(r[2]) for r in rows],
        "y1_printed": [float(r[3]) for r in rows],
        "y0": oracle.ref_conv0(oracle.KAT9_X, oracle.KAT9_H).tolist(),
        "y1": oracle.ref_conv1(oracle.KAT9_X, oracle.KAT9_H).tolist(),
    }
    assert kat["x"] == oracle.KAT9_X.tolist()
    with open(os.path.join(HERE, "conv_kat9.json"), "w") as f:
        json.dump(kat, f, indent=1)

    # --- seeded cases through the reference's conv0 / conv1 ---------------------
    rng = np.random.default_rng(0xD5B0)
    cases = {}
    for i, (nx, nh) in enumerate([(1, 1), (5, 9), (9, 9), (64, 16), (100, 21), (257, 32), (1000, 127), (4096, 16), (3000, 300)]):
        x = rng.uniform(-1, 1, nx)
        h = rng.uniform(-1, 1, nh)
        cases["x%d" % i] = x
        cases["h%d" % i] = h
        cases["y0_%d" % i] = oracle.ref_conv0(x, h)
        cases["y1_%d" % i] = oracle.ref_conv1(x, h)
    np.savez(os.path.join(HERE, "conv_ref_cases.npz"), **cases)

    # --- tone -> FIR (fft/src/main.c:320-387 call shape) -------------------------------
    tone = oracle.sin_tonegen([100.0, 12000.0], 48000.0, 4096)
    np.savez(os.path.join(HERE, "tone_fir.npz"),
             tone=tone,
             taps16=oracle.TAPS16, taps21=oracle.TAPS21,
             y16=oracle.convolve(tone, oracle.TAPS16), y21=oracle.convolve(tone, oracle.TAPS21))

    # --- fftc/main.c:150-154 input (arguments as written there, i.e. swapped) ----
    g = oracle.Tonegen()
    g.add(2000, 48000, 100, 0)
    g.add(4000, 48000, 100, 0.666)
    g.add(6000, 48000, 100, 0)
    fftc_x = g.read_complex(480)

    # --- DFT vectors ("parity unpinned") ---------------------------------------------
    ff = {"fftc_x": fftc_x, "fftc_y": oracle.dft(fftc_x, -1, "direct")}
    ff["fftc_z"] = oracle.dft(ff["fftc_y"], +1, "direct")
    for n in (8, 64, 480, 482, 1024, 4096):
        z = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
        fwd = oracle.dft(z, -1, "direct")
        bwd = oracle.dft(z, +1, "direct")
        assert np.abs(fwd - np.fft.fft(z)).max() < 1e-11 * n
        assert np.abs(bwd - np.fft.ifft(z) * n).max() < 1e-11 * n
        ff["in_%d" % n] = z
        ff["fwd_%d" % n] = fwd
        ff["bwd_%d" % n] = bwd
    xr = rng.uniform(-1, 1, 4096)
    gains = np.zeros(2049)
    gains[:300] = 1.0
    gains[300:600] = 0.5
    ff["filter_x"] = xr
    ff["filter_gains"] = gains
    ff["filter_y"] = oracle.filter(xr, gains)
    ff["r2c_4096"] = oracle.fft_r2c(xr)
    assert np.abs(ff["r2c_4096"] - np.fft.rfft(xr)).max() < 1e-9
    np.savez(os.path.join(HERE, "fft_cases.npz"), **ff)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
