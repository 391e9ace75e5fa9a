"""Golden vectors for the one FFT path of the reference whose FFT library IS in this image.

fftpy/main.py computes  y = |numpy.fft.rfft(x, norm='ortho')| / sum(...)  (compute_power_spectrum,
fftpy/main.py:30-45) of a 20 Hz sine sampled by read_signal (fftpy/main.py:11-27).  The module itself
cannot be imported (it opens a matplotlib window and loops forever at import, fftpy/main.py:71,83),
so its two functions are restated below, statement for statement in what they compute, and run
against numpy's own rfft -- the reference's FFT for this path.  Only inputs and outputs are stored:

    python tests/golden/make_golden_fftpy.py   ->  tests/golden/fftpy_spectrum.npz

The noise of read_signal comes from an unseeded random.seed() in the reference; here the
generator is seeded so that the vectors are reproducible (same formula: randint(0, 10000) / 1e6).
"""
import os
import random

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))


def read_signal(fsampl, tsampl, noise, seed):
    nsampl = 1 + int(fsampl / tsampl)                     # fftpy/main.py:13
    fsin = 20.0
    x = numpy.empty(nsampl)
    for i in range(0, nsampl):
        x[i] = numpy.sin(2.0 * numpy.pi * fsin * float(i) / fsampl)
    if noise:
        random.seed(seed)
        for i in range(0, nsampl):
            r = random.randint(0, 10000)
            x[i] += float(r) / 1000000.0
    return x


def compute_power_spectrum(x):
    c = numpy.fft.rfft(x, norm='ortho')                   # fftpy/main.py:32
    y = numpy.empty(c.size)
    q = 0.0
    for i in range(0, c.size):
        y[i] = numpy.sqrt(numpy.real(c[i]) ** 2 + numpy.imag(c[i]) ** 2)
        q += y[i]
    for i in range(0, c.size):
        y[i] = y[i] / q
    return y


def main():
    out = {"numpy_version": numpy.array(numpy.__version__)}
    # (fsampl, tsampl): the module's own 1000 Hz / 1 s (1001 samples), and other window lengths
    for i, (fs, ts, noise) in enumerate([(1000.0, 1.0, True), (1000.0, 1.0, False), (1000.0, 2.0, True), (4096.0, 1.0, True),
                                         (44100.0, 25.0, True), (999.0, 1.0, True), (8191.0, 1.0, True)]):
        x = read_signal(fs, ts, noise, 1000 + i)
        out["x%d" % i] = x
        out["y%d" % i] = compute_power_spectrum(x)
    numpy.savez(os.path.join(HERE, "fftpy_spectrum.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
