"""The oracle against the reference's own outputs (CPU only).

FIR / convolution: pinned by the reference's conv0/conv1 (compiled in place,
oracle/_ref) through the committed vectors, and by the table the reference
program prints.  FFT: "parity unpinned" (FFTW3 absent) -- checked against the
committed long-double DFT vectors and numpy's pocketfft.
"""
import json
import os

import numpy as np
import pytest

import oracle


def test_kat9_matches_reference_printout(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "conv_kat9.json")))
    x, h = np.array(kat["x"]), np.array(kat["h"])
    y0, y1 = oracle.conv0(x, h), oracle.conv1(x, h)
    assert np.array_equal(y0, np.array(kat["y0"]))
    assert np.array_equal(y1, np.array(kat["y1"]))
    # the printed table has 6 decimals
    assert np.allclose(y0, kat["y0_printed"], atol=5e-7, rtol=0)
    assert np.allclose(y1, kat["y1_printed"], atol=5e-7, rtol=0)


def test_conv_restatement_bit_exact_against_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "conv_ref_cases.npz"))
    n = len([k for k in g.files if k.startswith("x")])
    assert n >= 9
    for i in range(n):
        x, h = g["x%d" % i], g["h%d" % i]
        assert np.array_equal(oracle.conv0(x, h), g["y0_%d" % i]), i
        assert np.array_equal(oracle.conv1(x, h), g["y1_%d" % i]), i


@pytest.mark.skipif(oracle.ref_lib() is None, reason="reference tree not mounted / oracle/_ref not built")
def test_conv_restatement_against_live_reference():
    rng = np.random.default_rng(7)
    for nx, nh in [(1, 1), (2, 7), (33, 5), (500, 64), (2048, 127)]:
        x, h = rng.standard_normal(nx), rng.standard_normal(nh)
        assert np.array_equal(oracle.conv0(x, h), oracle.ref_conv0(x, h))
        assert np.array_equal(oracle.conv1(x, h), oracle.ref_conv1(x, h))


def test_convolve_agrees_with_conv0_and_conv1_identities(golden_dir):
    """`convolve` (fir/src/main.c:258-280) has no vector of its own in the reference; it is
    the same sum as conv0 in another order, and conv1 is `convolve` with x[0] dropped."""
    g = np.load(os.path.join(golden_dir, "conv_ref_cases.npz"))
    for i in range(9):
        x, h = g["x%d" % i], g["h%d" % i]
        full = oracle.convolve(x, h)
        assert len(full) == len(x) + len(h) and full[-1] == 0.0
        np.testing.assert_allclose(full[:len(x)], g["y0_%d" % i], rtol=0, atol=1e-13)
        x1 = x.copy(); x1[0] = 0.0
        assert np.array_equal(oracle.convolve(x1, h, len(x)), g["y1_%d" % i])
        b = oracle.fir_batch_f32(x.astype(np.float32), h.astype(np.float32), len(x), drop_first=True)[0]
        ref = oracle.conv1(x.astype(np.float32).astype(np.float64), h.astype(np.float32).astype(np.float64))
        np.testing.assert_allclose(b, ref, rtol=0, atol=1e-12)


def test_tone_fir_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "tone_fir.npz"))
    tone = oracle.sin_tonegen([100.0, 12000.0], 48000.0, 4096)
    assert np.array_equal(tone, g["tone"])
    assert np.array_equal(oracle.convolve(tone, oracle.TAPS16), g["y16"])
    assert np.array_equal(oracle.convolve(tone, oracle.TAPS21), g["y21"])
    assert len(g["y16"]) == 4096 + 16 and len(g["y21"]) == 4096 + 21


def test_stream_history_equals_one_long_block():
    rng = np.random.default_rng(3)
    h = rng.standard_normal(31).astype(np.float32)
    x = rng.standard_normal(1000).astype(np.float32)
    whole = oracle.fir_batch_f32(x, h, 1000)[0]
    a = oracle.fir_stream_f32(x[:400], np.zeros(30, np.float32), h)
    b = oracle.fir_stream_f32(x[400:], x[370:400], h)
    np.testing.assert_allclose(np.concatenate([a, b]), whole, rtol=0, atol=1e-12)


# ---- DFT: parity unpinned (see oracle/dsp_oracle.c header) -------------------

@pytest.mark.parametrize("n", [8, 64, 480, 482, 1024, 4096])
def test_dft_vectors(golden_dir, n):
    g = np.load(os.path.join(golden_dir, "fft_cases.npz"))
    z = g["in_%d" % n]
    if n <= 1024:
        assert np.array_equal(oracle.dft(z, -1, "direct"), g["fwd_%d" % n])
    np.testing.assert_allclose(np.fft.fft(z), g["fwd_%d" % n], rtol=0, atol=1e-11 * n)
    np.testing.assert_allclose(np.fft.ifft(z) * n, g["bwd_%d" % n], rtol=0, atol=1e-11 * n)
    if n & (n - 1) == 0:
        np.testing.assert_allclose(oracle.dft(z, -1, "radix2"), g["fwd_%d" % n], rtol=0, atol=1e-11 * n)
        np.testing.assert_allclose(oracle.dft(z, +1, "radix2"), g["bwd_%d" % n], rtol=0, atol=1e-11 * n)


def test_fftc_round_trip_vector(golden_dir):
    g = np.load(os.path.join(golden_dir, "fft_cases.npz"))
    x = g["fftc_x"]
    # fftc/main.c:151-153 passes (2000, fsampl) where tonegen_add expects (fsampl, freq):
    # dw = 2*pi*48000/2000, every tone lands on a multiple of 2*pi -> a constant signal
    assert np.allclose(x.imag, 0)
    np.testing.assert_allclose(g["fftc_z"] / 480, x, atol=1e-9)


def test_r2c_c2r_filter(golden_dir):
    g = np.load(os.path.join(golden_dir, "fft_cases.npz"))
    x = g["filter_x"]
    np.testing.assert_allclose(oracle.fft_r2c(x), g["r2c_4096"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(oracle.fft_c2r(g["r2c_4096"], 4096), 4096 * x, rtol=0, atol=1e-8)
    y = oracle.filter(x, g["filter_gains"])
    np.testing.assert_allclose(y, g["filter_y"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(y, np.fft.irfft(np.fft.rfft(x) * g["filter_gains"], 4096), rtol=0, atol=1e-12)
    # all-pass gains give the input back (the /n at util/filter/filter.c:102)
    np.testing.assert_allclose(oracle.filter(x, np.ones(2049)), x, rtol=0, atol=1e-12)


def test_spectrum_post_ops():
    rng = np.random.default_rng(11)
    z = rng.standard_normal(64) + 1j * rng.standard_normal(64)
    mag = np.abs(z)
    np.testing.assert_allclose(oracle.power_spectrum(z), mag / mag.sum(), rtol=1e-14)
    np.testing.assert_allclose(oracle.power_spectrum(z * 1e-9), np.abs(z * 1e-9), rtol=1e-14)  # sum <= 1e-5: not normalised
    np.testing.assert_allclose(oracle.ps_abs(z), 2 * mag[:32] / 64, rtol=1e-14)
    np.testing.assert_allclose(oracle.alsa_power_spectrum(z), mag[:32] / mag[:32].sum(), rtol=1e-14)
    d = (z.real - z.imag)[:32]
    want = d / d.sum() if d.sum() >= 1e-4 else d
    np.testing.assert_allclose(oracle.ps_percent(z), want, rtol=1e-14)
    phi = np.where(np.abs(z.real[:32]) >= 1e-5, np.arctan(z.imag[:32] / z.real[:32]), 0)
    np.testing.assert_allclose(oracle.compute_phi(z), phi, rtol=1e-14)


def test_tonegen_recurrence():
    g = oracle.Tonegen()
    g.add(48000.0, 1000.0, 2.0, 0.25)
    g.add(48000.0, 3000.0, 0.5, 0.0)
    a = g.read(100)
    b = g.read(50)  # continues from the accumulated phase
    w1 = w2 = 0.0
    ref = []
    for _ in range(150):
        ref.append(2.0 * np.cos(w1 + 0.25) + 0.5 * np.cos(w2 + 0.0))
        w1 += (2.0 * np.pi * 1000.0) / 48000.0
        w2 += (2.0 * np.pi * 3000.0) / 48000.0
    np.testing.assert_allclose(np.concatenate([a, b]), ref, rtol=0, atol=1e-12)


def test_capture_block_stages_compose_from_their_parts():
    """alsa/main.c:453-540 restated as whole-block functions equal the step-by-step composition."""
    rng = np.random.default_rng(11)
    fr = rng.integers(-32768, 32768, size=(2048, 2)).astype(np.int16)
    mono = (fr[:, 0].astype(np.float64) + fr[:, 1].astype(np.float64)) / 2
    y = oracle.convolve(mono, oracle.TAPS16, 2048 + 16)[:2048]
    out = oracle.alsa_do_fir(fr, oracle.TAPS16)
    assert np.array_equal(out[:, 0], np.trunc(y).astype(np.int64).astype(np.int16))
    assert np.array_equal(out[:, 0], out[:, 1])
    ps = oracle.alsa_do_power_spectrum(fr)
    z = np.fft.fft(mono)[:1024]
    assert np.allclose(ps, np.abs(z) / np.abs(z).sum(), rtol=1e-9, atol=1e-15)
    # tonegen/main.c "10 2000,400,0" at 44.1 kHz: the stream starts 400,400,383,383,336,336
    t = oracle.Tonegen()
    t.add(44100.0, 2000.0, 400.0, 0.0)
    raw = oracle.double_to_s16x2(t.read(5))
    assert raw.reshape(-1)[:6].tolist() == [400, 400, 383, 383, 336, 336]


def test_fftpy_spectrum_against_numpy_rfft_vectors(golden_dir):
    """The one FFT path of the reference whose FFT library is in this image: fftpy/main.py:30-45 calls
    numpy.fft.rfft.  The vectors hold numpy's own output (tests/golden/make_golden_fftpy.py); the
    checker's transform (long-double DFT off powers of two, radix-2 on them) must reproduce it."""
    g = np.load(os.path.join(golden_dir, "fftpy_spectrum.npz"))
    for i in range(7):
        x, want = g["x%d" % i], g["y%d" % i]
        got = oracle.fftpy_power_spectrum(x)
        assert got.shape == want.shape == (len(x) // 2 + 1,)
        assert np.linalg.norm(got - want) <= 1e-12 * np.linalg.norm(want), i
        assert abs(got.sum() - 1.0) < 1e-12
