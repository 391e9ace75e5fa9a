"""Golden vectors for the one FFT path of the reference whose FFT library IS in this image.

fftpy/main.py:30-45 turns a window x into |numpy.fft.rfft(x, norm='ortho')| divided by its sum;
its windows are a 20 Hz sine of 1 + int(fsampl / tsampl) samples plus up to 0.01 of uniform
noise (fftpy/main.py:11-27).  The module cannot be imported (it opens a plot window and loops
forever at import, fftpy/main.py:71,83), and nothing of it is needed: the arithmetic is numpy's,
called here directly.  Only inputs and outputs are stored:

    python tests/golden/make_golden_fftpy.py   ->  tests/golden/fftpy_spectrum.npz

This script runs in the build container only (it is not shipped to the GPU box).
"""
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# (fsampl, tsampl, noise): the module's own 1000 Hz / 1 s (1001 samples), and other window lengths
WINDOWS = [(1000.0, 1.0, True), (1000.0, 1.0, False), (1000.0, 2.0, True), (4096.0, 1.0, True),
           (44100.0, 25.0, True), (999.0, 1.0, True), (8191.0, 1.0, True)]


def main():
    out = {"numpy_version": np.array(np.__version__)}
    for i, (fs, ts, noise) in enumerate(WINDOWS):
        n = 1 + int(fs / ts)
        x = np.sin(2.0 * np.pi * 20.0 * np.arange(n, dtype=np.float64) / fs)
        if noise:
            rnd = random.Random(1000 + i)
            x += np.array([rnd.randint(0, 10000) for _ in range(n)], dtype=np.float64) / 1000000.0
        y = np.abs(np.fft.rfft(x, norm='ortho'))
        out["x%d" % i] = x
        out["y%d" % i] = y / y.sum()
    np.savez(os.path.join(HERE, "fftpy_spectrum.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
