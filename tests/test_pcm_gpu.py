"""Capture-format edges of the path: interleaved 16-bit stereo blocks in and out, as the
reference's sound-card loop feeds its FIR and spectrum stages (alsa/main.c:453-540) and its
tone generator writes them (tonegen/main.c:14-24).  Integer results are compared bit for bit."""
import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu


def stereo_blocks(rng, nblock, nsampl, amp=32767):
    return rng.integers(-amp - 1, amp + 1, size=(nblock, nsampl, 2), dtype=np.int64).astype(np.int16)


@pytest.mark.parametrize("nblock,nsampl,taps", [
    (3, 2048, "taps16"),       # the sound-card loop's block and table (alsa/main.c:389-407,530)
    (2, 2048, "taps21"),       # tap magnitudes ~1e4: results far outside the 16-bit range wrap as the cast does
    (5, 1, "taps16"),
    (4, 15, "taps16"),         # shorter than the filter
    (3, 2049, "sinc127"),      # tile boundary + 1
    (2, 10000, "sinc300"),
])
def test_fir_s16x2_blocks_bit_exact(nblock, nsampl, taps):
    rng = np.random.default_rng(nblock * 1000 + nsampl)
    h = {"taps16": oracle.TAPS16, "taps21": oracle.TAPS21,
         "sinc127": oracle.windowed_sinc_taps(127, 6000.0, 48000.0),
         "sinc300": oracle.windowed_sinc_taps(300, 3000.0, 48000.0)}[taps]
    frames = stereo_blocks(rng, nblock, nsampl)
    frames[0, : min(nsampl, 40)] = 32767          # full-scale run: largest sums
    plan = dsp_amd.FirPlan(h, dtype="f64")
    got = plan.exec_s16x2(torch.from_numpy(frames).cuda()).cpu().numpy()
    for b in range(nblock):
        want = oracle.alsa_do_fir(frames[b], h)
        assert np.array_equal(got[b], want), (b, np.flatnonzero((got[b] != want).any(axis=1))[:5])
    assert np.array_equal(got[..., 0], got[..., 1])


def test_fir_s16x2_strided_blocks_and_argument_checks():
    rng = np.random.default_rng(5)
    h = oracle.TAPS16
    plan = dsp_amd.FirPlan(h, dtype="f64")
    big = torch.from_numpy(stereo_blocks(rng, 4, 3000)).cuda()
    view = big[:, :2048]                                  # blocks 3000 frames apart
    out = torch.zeros_like(big)
    plan.exec_s16x2(view, out=out[:, :2048])
    assert int(out[:, 2048:].abs().max()) == 0            # nothing written past a block
    for b in range(4):
        assert np.array_equal(out[b, :2048].cpu().numpy(), oracle.alsa_do_fir(view[b].cpu().numpy(), h))
    with pytest.raises(dsp_amd.DspmiError):
        plan.exec_s16x2(big, out=big)                     # in place: tiles would read overwritten frames
    with pytest.raises(dsp_amd.DspmiError):
        dsp_amd.FirPlan(h, dtype="f32").exec_s16x2(big)   # exact integer results need the f64 plan


@pytest.mark.parametrize("nsampl", [2048, 64, 4096, 480])
def test_spectrum_s16x2_blocks(nsampl):
    rng = np.random.default_rng(nsampl)
    nblock = 5
    frames = stereo_blocks(rng, nblock, nsampl, amp=20000)
    t = np.arange(nsampl)
    tone = (8000 * np.sin(2 * np.pi * 0.1 * t)).astype(np.int16)
    frames[1, :, 0] = tone
    frames[1, :, 1] = tone
    plan = dsp_amd.FftPlan(nsampl, dtype="f32")
    got = plan.r2c_spectrum_s16x2(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert got.shape == (nblock, nsampl // 2)
    for b in range(nblock):
        want = oracle.alsa_do_power_spectrum(frames[b])
        err = np.linalg.norm(got[b] - want) / np.linalg.norm(want)
        assert err <= 1e-5, (b, err)                      # f32 transform against the double oracle
    assert int(np.argmax(got[1])) == round(0.1 * nsampl)


def test_tonegen_s16x2_frames():
    fs = 44100.0
    freqs = np.array([[2000.0, 400.0 + 10 * c] for c in range(9)])
    amps = np.array([[400.0, 3000.0]] * 9)
    phis = np.array([[0.0, 0.5]] * 9)
    gen = dsp_amd.ToneGen(fs, freqs, amps, phis)
    gen_seq = dsp_amd.ToneGen(fs, freqs, amps, phis)
    a = gen.read_s16x2(5000)
    b = gen.read_s16x2(777)
    # rotation inside a chunk vs evaluation per sample: the sums agree to ~1e-10 at this amplitude,
    # so the truncated values can differ only on an integer boundary
    assert int((a != gen_seq.read_s16x2(5000, sequential=True)).sum()) <= 4
    got = torch.cat([a, b], dim=1).cpu().numpy()
    assert np.array_equal(got[..., 0], got[..., 1])
    for c in (0, 4, 8):
        t = oracle.Tonegen()
        for j in range(2):
            t.add(fs, freqs[c, j], amps[c, j], phis[c, j])
        buf = t.read(5777)
        want = oracle.double_to_s16x2(buf)
        # the cast truncates: a sample differs only where the double sum sits within rounding
        # error of an integer (device and host cosines may differ in the last bit)
        diff = np.flatnonzero(got[c, :, 0] != want[:, 0])
        assert all(abs(buf[i] - np.round(buf[i])) < 1e-9 for i in diff), diff[:5]
        assert len(diff) <= 2
    # the reference's command-line example "10 2000,400,0": 441 samples of 400*cos at 2 kHz
    g = dsp_amd.ToneGen(fs, [[2000.0]], [[400.0]], [[0.0]])
    first = g.read_s16x2(9).cpu().numpy()[0, :, 0]
    assert list(first[:2]) == [400, 383]
