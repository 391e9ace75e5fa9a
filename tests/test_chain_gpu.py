"""Stages around the FIR / FFT (spectrum post-ops, tone generator) and the whole chain
tone generator -> FIR low-pass -> N=4096 real FFT -> normalised magnitude spectrum, streamed in
chunks with carried history, against the oracle run end to end on the CPU."""
import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu


def rel_l2(got, want):
    den = np.linalg.norm(want)
    return np.linalg.norm(np.asarray(got) - want) / den if den > 0 else np.linalg.norm(got)


@pytest.mark.parametrize("nbin", [1, 63, 240, 1024, 2049, 5000])
def test_power_spectrum_modes(nbin):
    rng = np.random.default_rng(nbin)
    z = (rng.standard_normal((6, nbin + 3)) + 1j * rng.standard_normal((6, nbin + 3))).astype(np.complex64)
    z[1] *= 1e-9          # sum below the 1e-5 guard: left un-normalised by fft_to_power_spectrum
    zd = torch.from_numpy(z).cuda()
    got0 = dsp_amd.power_spectrum(zd, nbin, dsp_amd.PS_NORM_GUARDED).cpu().numpy()
    got1 = dsp_amd.power_spectrum(zd, nbin, dsp_amd.PS_NORM).cpu().numpy()
    got2 = dsp_amd.power_spectrum(zd, nbin, dsp_amd.PS_ABS, scale=float(2 * nbin)).cpu().numpy()
    for f in range(6):
        zz = z[f, :nbin].astype(np.complex128)
        assert rel_l2(got0[f], oracle.power_spectrum(zz)) <= 2e-6
        mag = np.abs(zz)
        assert rel_l2(got1[f], mag / mag.sum()) <= 2e-6
        zfull = np.concatenate([zz, np.zeros(nbin)])          # ps_abs looks at nx/2 bins of an nx-point spectrum
        assert rel_l2(got2[f], oracle.ps_abs(zfull)) <= 2e-6
    if nbin > 1:
        assert abs(got1[0].sum() - 1.0) < 1e-5


def same_samples(a, b, scale):
    """The time-parallel generator steps cos / sin by rotation inside a chunk where the one-lane
    walk evaluates them afresh: the double sums agree to ~1e-13, so the f32 samples are the same
    except where a sum sits on a rounding boundary (then one unit in the last place apart)."""
    d = (a.double() - b.double()).abs()
    return bool(d.max() <= 1.3e-7 * scale) and float((d == 0).double().mean()) >= 0.9999


def test_tonegen_phase_jumps_are_the_sequential_additions():
    """The time-parallel generator reaches every chunk's starting angle by collapsing the rounded
    additions of one binade into an integer step.  Bit for bit the one-lane walk, over increments
    from far below the angle's rounding unit to far above the angle, exact rounding ties
    (increments of 1.5, 2.5 ... units of the angle's last place), odd and even starting
    multiples, zero and tiny starting angles, and negative increments (no shortcut)."""
    rng = np.random.default_rng(7)
    nchan, ntone = 512, 8
    dw = 10.0 ** rng.uniform(-13, 3, (nchan, ntone))
    w0 = np.where(rng.random((nchan, ntone)) < 0.3, 0.0, 10.0 ** rng.uniform(-8, 7, (nchan, ntone)))
    # exact ties: angle in [2^20, 2^21) has unit 2^-32; increments (q + 1/2) units
    ulp = 2.0 ** -32
    for i in range(64):
        w0[i, 0] = 2.0 ** 20 + ulp * (i * 977 + (i & 1))          # odd and even multiples
        dw[i, 0] = ulp * (rng.integers(0, 5) + 0.5)
        w0[i, 1] = 2.0 ** 21 - ulp * rng.integers(1, 2000)        # about to cross a binade
        dw[i, 1] = ulp * (rng.integers(1, 1 << 20) + 0.5)
        dw[i, 2] = -abs(dw[i, 2])                                 # negative: walked, not jumped
        w0[i, 3] = 5e-324 * rng.integers(0, 50)                   # subnormal start
    gens = []
    for _ in range(2):
        g = dsp_amd.ToneGen(48000.0, np.ones((nchan, ntone)), np.ones((nchan, ntone)), np.zeros((nchan, ntone)))
        g.dw.copy_(torch.from_numpy(dw).cuda())
        g.w.copy_(torch.from_numpy(w0).cuda())
        gens.append(g)
    for n in (3000, 1, 255, 257, 40000):
        a = gens[0].read(n, sequential=True)
        b = gens[1].read(n)
        assert torch.equal(gens[0].w, gens[1].w), n
        assert same_samples(a, b, 8.0), n
    # and the walk itself is the host's: a few sequences replayed in numpy
    wend = gens[1].w.cpu().numpy()
    for c, j in ((0, 0), (1, 1), (5, 2), (100, 4), (511, 7)):
        acc = np.float64(w0[c, j])
        d = np.float64(dw[c, j])
        for _ in range(3000 + 1 + 255 + 257 + 40000):
            acc = acc + d
        assert acc == wend[c, j], (c, j)


def test_tonegen_matches_recurrence_and_continues():
    fs = 48000.0
    freqs = np.array([[100.0 * (c + 1), 12000.0] for c in range(70)])
    amps = np.array([[1.0, 0.5]] * 70)
    phis = np.array([[0.0, 0.3]] * 70)
    gen = dsp_amd.ToneGen(fs, freqs, amps, phis)
    a = gen.read(1000).cpu().numpy()
    b = gen.read(333).cpu().numpy()          # continues from the accumulated phases
    for c in (0, 1, 37, 69):
        t = oracle.Tonegen()
        for j in range(2):
            t.add(fs, freqs[c, j], amps[c, j], phis[c, j])
        want = t.read(1333)
        got = np.concatenate([a[c], b[c]])
        assert np.abs(got - want.astype(np.float32)).max() <= 2e-7 * 1.5   # f32 rounding of a double sum
    # the time-parallel form reaches the same angles (identical running phases); see same_samples
    g_seq, g_par = dsp_amd.ToneGen(fs, freqs, amps, phis), dsp_amd.ToneGen(fs, freqs, amps, phis)
    for n in (1000, 333, 70000):
        assert same_samples(g_seq.read(n, sequential=True), g_par.read(n), 1.5)
    assert torch.equal(g_seq.w, g_par.w)
    # the sine form of fft/src/main.c:265-291
    g2 = dsp_amd.ToneGen(fs, [[100.0, 12000.0]], use_sin=True)
    got = g2.read(4096).cpu().numpy()[0]
    want = oracle.sin_tonegen([100.0, 12000.0], fs, 4096)
    assert np.abs(got - want.astype(np.float32)).max() <= 3e-7


@pytest.mark.parametrize("ntone", [9, 17, 32])
def test_tonegen_up_to_the_reference_limit_of_32_tones(ntone):
    """tonegen_t holds up to 32 tones (tonegen/tonegen.h:14-19); beyond 8 the device generator
    takes them in groups, every sample still adding its tones in the reference's order."""
    fs, nchan = 44100.0, 6
    rng = np.random.default_rng(ntone)
    freqs = rng.uniform(20.0, 20000.0, (nchan, ntone))
    amps = rng.uniform(0.1, 100.0, (nchan, ntone))
    phis = rng.uniform(-3.0, 3.0, (nchan, ntone))
    phis[:, ::3] = 0.0
    g_par, g_seq = dsp_amd.ToneGen(fs, freqs, amps, phis), dsp_amd.ToneGen(fs, freqs, amps, phis)
    outs = []
    for n in (700, 5000, 1, 20000):
        a, b = g_par.read(n), g_seq.read(n, sequential=True)
        assert same_samples(b, a, float(amps.sum(axis=1).max())), n
        outs.append(a.cpu().numpy())
    assert torch.equal(g_par.w, g_seq.w)
    got = np.concatenate(outs, axis=1)
    for c in (0, nchan - 1):
        t = oracle.Tonegen()
        for j in range(ntone):
            t.add(fs, freqs[c, j], amps[c, j], phis[c, j])
        want = t.read(got.shape[1])
        scale = np.abs(want).max()
        assert np.abs(got[c] - want.astype(np.float32)).max() <= 2e-7 * scale
    with pytest.raises(dsp_amd.DspmiError):
        dsp_amd.ToneGen(fs, np.ones((1, 33))).read(10)


def test_chain_streaming_matches_oracle_end_to_end():
    """tonegen -> 16-tap FIR (alsa/main.c:389-407) -> N=4096 r2c -> |X|/sum, 3 chunks of 2 frames."""
    fs, nchan, nfft, frames_per_chunk, nchunks = 48000.0, 5, 4096, 2, 3
    freqs = np.array([[100.0 * (c + 1), 12000.0] for c in range(nchan)])
    taps = oracle.TAPS16.astype(np.float32)
    gen = dsp_amd.ToneGen(fs, freqs, use_sin=True)
    fir = dsp_amd.FirPlan(taps, "f32")
    fft = dsp_amd.FftPlan(nfft, "f32")
    hist = torch.zeros((nchan, len(taps) - 1), dtype=torch.float32, device="cuda")
    spectra, filtered = [], []
    n = nfft * frames_per_chunk
    for _ in range(nchunks):
        x = gen.read(n)
        y = fir.exec(x, n, hist=hist)
        hist = x[:, -(len(taps) - 1):].contiguous()
        bins = fft.r2c(y.view(nchan * frames_per_chunk, nfft))
        spectra.append(dsp_amd.power_spectrum(bins).view(nchan, frames_per_chunk, nfft // 2 + 1).cpu().numpy())
        filtered.append(y.cpu().numpy())
    got_ps = np.concatenate(spectra, axis=1)
    got_y = np.concatenate(filtered, axis=1)
    total = n * nchunks
    for c in range(nchan):
        tone = oracle.sin_tonegen(list(freqs[c]), fs, total).astype(np.float32)
        want_y = oracle.fir_batch_f32(tone, taps, total)[0]
        assert rel_l2(got_y[c], want_y) <= 1e-6
        for f in range(frames_per_chunk * nchunks):
            want = oracle.power_spectrum(oracle.fft_r2c(want_y[f * nfft:(f + 1) * nfft]))
            assert rel_l2(got_ps[c, f], want) <= 1e-5, (c, f)
    # the 12 kHz tone is in the stop band of the 16-tap table (ripple ~0.35): its bin ends up
    # below the pass-band tone's although both tones enter with unit amplitude
    k_lo, k_hi = round(100.0 / fs * nfft), round(12000.0 / fs * nfft)
    assert got_ps[0, 1, k_hi] < 0.5 * got_ps[0, 1, k_lo]


@pytest.mark.parametrize("n", [4, 8, 64, 1024, 4096, 8192, 16384, 1000])
def test_fused_r2c_spectrum(n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, (9, n)).astype(np.float32)
    x[3] *= 1e-10                                     # below the 1e-5 guard
    plan = dsp_amd.FftPlan(n, "f32")
    xd = torch.from_numpy(x).cuda()
    ps0 = plan.r2c_spectrum(xd).cpu().numpy()
    ps1 = plan.r2c_spectrum(xd, nbin=n // 2, mode=dsp_amd.PS_NORM).cpu().numpy()
    ps2 = plan.r2c_spectrum(xd, nbin=n // 2, mode=dsp_amd.PS_ABS, scale=float(n)).cpu().numpy()
    bins = plan.r2c(xd).cpu().numpy()
    for f in range(9):
        want_bins = oracle.fft_r2c(x[f].astype(np.float64))
        assert rel_l2(bins[f], want_bins) <= 1e-5
        assert rel_l2(ps0[f], oracle.power_spectrum(want_bins)) <= 1e-5
        mag = np.abs(want_bins[:n // 2])
        assert rel_l2(ps1[f], mag / mag.sum()) <= 1e-5
        assert rel_l2(ps2[f], 2 * mag / n) <= 1e-5


@pytest.mark.parametrize("nx", [2, 480, 1024, 5000])
def test_transform_readouts_of_the_fftc_demo(nx):
    """ps_percent, ps_abs, compute_phi (fftc/main.c:46-123): host routines in double, and the batched
    f32 device forms; a frame whose sum stays under ps_percent's 1e-4 guard is left unscaled and real
    parts under 1e-5 give a zero phase."""
    rng = np.random.default_rng(nx)
    z = rng.uniform(-1, 1, nx) + 1j * rng.uniform(-1, 1, nx)
    z[: nx // 2].real += 0.7                       # a sum clearly above the guard
    z[0] = 3e-6 + 0.5j                             # |re| below 1e-5: phase 0
    np.testing.assert_allclose(dsp_amd.ps_percent(z), oracle.ps_percent(z), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(dsp_amd.ps_abs(z), oracle.ps_abs(z), rtol=1e-15, atol=0)   # the device may fuse re*re + im*im
    np.testing.assert_allclose(dsp_amd.compute_phi(z), oracle.compute_phi(z), rtol=1e-14, atol=1e-16)
    assert dsp_amd.compute_phi(z)[0] == 0.0
    tiny = z * 1e-7
    np.testing.assert_allclose(dsp_amd.ps_percent(tiny), oracle.ps_percent(tiny), rtol=1e-12, atol=1e-20)
    # device forms, three frames: the one above, its negative (sum < 0: unscaled), a tiny one
    frames = np.stack([z, -z, tiny]).astype(np.complex64)
    zd = torch.from_numpy(frames).cuda()
    nb = nx // 2
    got = dsp_amd.power_spectrum(zd, nb, dsp_amd.PS_PERCENT).cpu().numpy()
    phi = dsp_amd.phase(zd, nb).cpu().numpy()
    for f in range(3):
        zz = frames[f].astype(np.complex128)
        np.testing.assert_allclose(got[f], oracle.ps_percent(zz), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(phi[f], oracle.compute_phi(zz), rtol=2e-6, atol=1e-6)


def test_tonegen_complex_output_feeds_the_c2c_transform():
    """tonegen_read_complex (tonegen/tonegen.c:42-59) on the device, then the fftc demo's transform
    of it (fftc/main.c:146-158: tones of amplitude 100 at 2, 4 and 6 kHz, 480 points at 48 kHz)."""
    fs, n = 48000.0, 480
    freqs = [[2000.0, 4000.0, 6000.0]] * 3
    amps = [[100.0, 100.0, 100.0]] * 3
    phis = [[0.0, 0.666, 0.0]] * 3
    gen = dsp_amd.ToneGen(fs, freqs, amps, phis)
    ref = dsp_amd.ToneGen(fs, freqs, amps, phis)
    z = gen.read_complex(n)
    x = ref.read(n)
    assert torch.equal(z.real, x) and int(z.imag.abs().max()) == 0
    z2 = gen.read_complex(100, sequential=True)           # the stream continues where it stopped
    assert torch.equal(z2.real, ref.read(100, sequential=True))
    t = oracle.Tonegen()
    for j in range(3):
        t.add(fs, freqs[0][j], amps[0][j], phis[0][j])
    want = t.read_complex(n)
    np.testing.assert_allclose(z[0].cpu().numpy(), want, rtol=0, atol=2e-5)
    spec = dsp_amd.FftPlan(n, "f32").c2c(z).cpu().numpy()
    peaks = np.argsort(-np.abs(spec[0][: n // 2]))[:3]
    assert sorted(peaks) == [20, 40, 60]                  # 2, 4, 6 kHz at 100 Hz per bin


@pytest.mark.parametrize("nh", [2, 16, 21, 33, 64, 127, 200, 305])
def test_fir_and_spectrum_in_one_kernel(nh):
    """dspmi_fir_spectrum_f32 against the oracle run stage by stage in double (filter from the
    carried history, then |rfft| / sum of every 4096 filtered samples) and against the two
    device calls it replaces; every normalisation mode, with and without the Nyquist bin, rows
    of several frames, a frame whose sum stays under the 1e-5 guard."""
    rng = np.random.default_rng(nh)
    nchan, nfft, frames = 7, 4096, 3
    nx = nfft * frames
    h = (oracle.windowed_sinc_taps(nh) if nh > 1 else np.array([0.8])).astype(np.float32)
    if nh == 16:
        h = np.asarray(oracle.TAPS16, dtype=np.float32)        # alsa/main.c:389-407, the chain's own short filter
    fir = dsp_amd.FirPlan(h, "f32")
    assert ("f16x2" in fir.variant) == (nh >= 33)                # shorter plans hold the f16 tap images for this entry point only
    fft = dsp_amd.FftPlan(nfft, "f32")
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    x[2] *= 1e-9                                   # spectra below the guard: left un-normalised in mode 0
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)
    hist[2] *= 1e-9
    xd, hd = torch.from_numpy(x).cuda(), torch.from_numpy(hist).cuda()
    want_y = oracle.fir_batch_f32(np.concatenate([hist, x], axis=1), h, nh - 1 + nx)[:, nh - 1:]
    y = fir.exec(xd, nx, hist=hd)
    for nbin, mode, scale in ((2049, dsp_amd.PS_NORM_GUARDED, 1.0), (2048, dsp_amd.PS_NORM, 1.0), (2048, dsp_amd.PS_ABS, float(nfft)),
                              (2049, dsp_amd.PS_ABS, 3.0)):
        got = fir.spectrum(fft, xd, hist=hd, nbin=nbin, mode=mode, scale=scale).cpu().numpy()
        two = fft.r2c_spectrum(y.view(-1, nfft), nbin=nbin, mode=mode, scale=scale).cpu().numpy()
        assert got.shape == (nchan * frames, nbin)
        for c in range(nchan):
            for f in range(frames):
                bins = oracle.fft_r2c(want_y[c, f * nfft:(f + 1) * nfft])
                mag = np.abs(bins[:nbin])
                if mode == dsp_amd.PS_ABS:
                    want = 2 * mag / scale
                elif mode == dsp_amd.PS_NORM or mag.sum() > 1e-5:
                    want = mag / mag.sum()
                else:
                    want = mag
                assert rel_l2(got[c * frames + f], want) <= 1e-5, (nh, nbin, mode, c, f)
                assert rel_l2(got[c * frames + f], two[c * frames + f]) <= 1e-5, (nh, nbin, mode, c, f)
    # without history every row starts from zero state
    got = fir.spectrum(fft, xd).cpu().numpy()
    two = fft.r2c_spectrum(fir.exec(xd, nx).view(-1, nfft)).cpu().numpy()
    for i in range(nchan * frames):
        assert rel_l2(got[i], two[i]) <= 1e-5, i


def test_fir_and_spectrum_walks_many_frames():
    """Enough frames that every workgroup walks several (the steady state of the tile pipeline,
    with the frame image reused tile after tile), rows of one frame and of many; the two device
    calls as reference, spot frames against the oracle."""
    nh, nfft = 127, 4096
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    fir, fft = dsp_amd.FirPlan(h, "f32"), dsp_amd.FftPlan(nfft, "f32")
    g = torch.Generator(device="cuda").manual_seed(3)
    for nchan, frames in ((3000, 1), (96, 40)):
        x = torch.rand((nchan, frames * nfft), device="cuda", generator=g) * 2 - 1
        hist = torch.rand((nchan, nh - 1), device="cuda", generator=g) * 2 - 1
        got = fir.spectrum(fft, x, hist=hist, nbin=2048, mode=dsp_amd.PS_NORM)
        two = fft.r2c_spectrum(fir.exec(x, hist=hist).view(-1, nfft), nbin=2048, mode=dsp_amd.PS_NORM)
        err = ((got - two).double().norm(dim=1) / two.double().norm(dim=1)).max().item()
        assert err <= 1e-5, (nchan, frames, err)
        assert abs(got.sum(dim=1) - 1).max().item() < 1e-4
        for c, f in ((0, 0), (nchan - 1, frames - 1), (nchan // 2, frames // 2)):
            lo = f * nfft
            seg = torch.cat([hist[c], x[c]])[lo:lo + nfft + nh - 1].cpu().numpy()
            yy = oracle.fir_batch_f32(seg[None, :], h, len(seg))[0][nh - 1:]
            mag = np.abs(oracle.fft_r2c(yy)[:2048])
            assert rel_l2(got[c * frames + f].cpu().numpy(), mag / mag.sum()) <= 1e-5, (c, f)


def test_fir_and_spectrum_refuses_what_it_cannot_run():
    fft = dsp_amd.FftPlan(4096, "f32")
    x = torch.zeros((2, 8192), device="cuda")
    with pytest.raises(dsp_amd.DspmiError, match="f16"):
        dsp_amd.FirPlan(oracle.windowed_sinc_taps(400).astype(np.float32), "f32").spectrum(fft, x)   # 400 taps: overlap-save
    fir = dsp_amd.FirPlan(oracle.windowed_sinc_taps(127).astype(np.float32), "f32")
    with pytest.raises(dsp_amd.DspmiError, match="4096"):
        fir.spectrum(dsp_amd.FftPlan(2048, "f32"), x)
    with pytest.raises(dsp_amd.DspmiError, match="whole frames"):
        fir.spectrum(fft, torch.zeros((2, 5000), device="cuda"), out=torch.zeros((2, 2049), device="cuda"))


def nearly_same_samples(a, b, scale):
    """Samples computed in the fused kernel against the generator's own: the double sums agree to
    ~1e-10 of the amplitudes, so the f32 values are the same except where a sum sits that close to
    a rounding boundary (one unit in the last place apart) or to zero."""
    d = (a.double() - b.double()).abs()
    return bool(d.max() <= 1.3e-7 * scale) and float((d > 1e-9 * scale).double().mean()) <= 2e-3


def _chain_two_calls(gen, fir, fft, n, hist, mode=0, nbin=None):
    x = gen.read(n)
    ps = fir.spectrum(fft, x, hist=hist, mode=mode, nbin=nbin)
    return ps, x[:, n - (fir.nh - 1):].contiguous(), x


@pytest.mark.parametrize("nh,ntone,use_sin", [(16, 2, True), (127, 2, True), (127, 1, False), (33, 3, False),
                                              (305, 4, False), (2, 2, True), (200, 2, False)])
def test_generator_filter_and_spectrum_in_one_kernel(nh, ntone, use_sin):
    """dspmi_tone_fir_spectrum_f32 against the calls it replaces (dspmi_tonegen_f32, then
    dspmi_fir_spectrum_f32), streamed in chunks with the history handed from call to call: same
    running angles bit for bit, the carried samples the generator's own (a sample whose double sum
    sits on a rounding boundary may differ by one unit in the last place), spectra within 1e-6, and
    every spectrum within 1e-5 of the oracle's chain run in double.  Rows start from angle zero, so
    the first tiles cross the powers of two above 2^13 (tiles the kernel reads from the stand-alone
    generator's output) and the later ones run on the tables alone."""
    fs, nchan, nfft = 48000.0, 6, 4096
    rng = np.random.default_rng(nh * 10 + ntone)
    freqs = np.array([[100.0 * (c + 1), 12000.0, 5321.7, 17.3][:ntone] for c in range(nchan)])
    amps = rng.uniform(0.2, 2.0, (nchan, ntone))
    phis = rng.uniform(-3.0, 3.0, (nchan, ntone))
    phis[:, 0] = 0.0
    taps = (oracle.TAPS16 if nh == 16 else oracle.windowed_sinc_taps(nh)).astype(np.float32)
    mk = lambda: dsp_amd.ToneGen(fs, freqs, None if use_sin else amps, None if use_sin else phis, use_sin=use_sin)
    g_one, g_two = mk(), mk()
    fir, fft = dsp_amd.FirPlan(taps, "f32"), dsp_amd.FftPlan(nfft, "f32")
    hist_a = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")
    hist_b = torch.empty_like(hist_a)
    hist_two = hist_a.clone()
    scale = float(ntone if use_sin else amps.sum(axis=1).max())
    spectra, signal = [], []
    for i, frames in enumerate((8, 1, 3)):
        n = frames * nfft
        first = i == 0
        got = fir.tone_spectrum(fft, g_one, n, hist=None if first else hist_a, hist_out=hist_b)
        want, hist_two, x = _chain_two_calls(g_two, fir, fft, n, None if first else hist_two)
        assert torch.equal(g_one.w, g_two.w)
        if nh > 1:
            assert nearly_same_samples(hist_b, hist_two, scale), (i, (hist_b - hist_two).abs().max())
        g, w_ = got.cpu().numpy(), want.cpu().numpy()
        for f in range(g.shape[0]):
            assert rel_l2(g[f], w_[f]) <= 1e-6, (i, f, rel_l2(g[f], w_[f]))
        spectra.append(g.reshape(nchan, frames, -1))
        signal.append(x.cpu().numpy())
        hist_a, hist_b = hist_b, hist_a
    # end to end against the oracle: its generator (double), its FIR on the f32 samples, its transform
    got_ps = np.concatenate(spectra, axis=1)
    total = got_ps.shape[1] * nfft
    for c in (0, nchan - 1):
        if use_sin:
            tone = oracle.sin_tonegen(list(freqs[c]), fs, total)
        else:
            t = oracle.Tonegen()
            for j in range(ntone):
                t.add(fs, freqs[c, j], amps[c, j], phis[c, j])
            tone = t.read(total)
        want_y = oracle.fir_batch_f32(tone.astype(np.float32), taps, total)[0]
        for f in range(got_ps.shape[1]):
            want = oracle.power_spectrum(oracle.fft_r2c(want_y[f * nfft:(f + 1) * nfft]))
            assert rel_l2(got_ps[c, f], want) <= 1e-5, (c, f)


def test_generator_in_the_kernel_at_large_angles_and_odd_increments():
    """Angles far along a stream (1e5 ... 3e8 rad: the exact increment of a tile differs from dw by
    up to half a unit of the angle, which the kernel applies as a drift), an angle about to cross a
    power of two, a tone that stands still (dw = 0), decreasing angles and negative starts (no
    tables: those rows come from the stand-alone generator), and the spectrum modes."""
    nchan, ntone, nfft, nh = 8, 2, 4096, 63
    rng = np.random.default_rng(5)
    dw = rng.uniform(0.01, 3.1, (nchan, ntone))
    dw[:, 0] = rng.uniform(0.01, 0.6, nchan)   # one tone in the pass band: the errors are then relative to a signal, not to a residue
    w0 = 10.0 ** rng.uniform(5, 8.5, (nchan, ntone))
    w0[1, 0] = 2.0 ** 24 - 1000.0       # crosses 2^24 in the first tile
    dw[2, 1] = 0.0
    w0[2, 1] = 0.7
    dw[3, 0] = -0.3                      # decreasing
    w0[4, 1] = -5.0                      # negative start
    taps = oracle.windowed_sinc_taps(nh).astype(np.float32)
    fir, fft = dsp_amd.FirPlan(taps, "f32"), dsp_amd.FftPlan(nfft, "f32")
    gens = []
    for _ in range(2):
        g = dsp_amd.ToneGen(48000.0, np.ones((nchan, ntone)), use_sin=True)
        g.dw.copy_(torch.from_numpy(dw).cuda())
        g.w.copy_(torch.from_numpy(w0).cuda())
        gens.append(g)
    hist = torch.from_numpy(rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)).cuda()
    hout = torch.empty_like(hist)
    for mode, nbin, frames in ((0, 2049, 5), (1, 2048, 2), (2, 2049, 1)):
        n = frames * nfft
        got = fir.tone_spectrum(fft, gens[0], n, hist=hist, hist_out=hout, mode=mode, nbin=nbin, scale=4096.0)
        x = gens[1].read(n)
        want = fir.spectrum(fft, x, hist=hist, mode=mode, nbin=nbin, scale=4096.0)
        assert torch.equal(gens[0].w, gens[1].w)
        assert nearly_same_samples(hout, x[:, n - (nh - 1):], 2.0)
        g, w_ = got.cpu().numpy(), want.cpu().numpy()
        for f in range(g.shape[0]):
            assert rel_l2(g[f], w_[f]) <= 1e-6, (mode, f, rel_l2(g[f], w_[f]))
        hist = hout.clone()


def test_generator_in_the_kernel_refuses_what_it_cannot_run():
    fir, fft = dsp_amd.FirPlan(oracle.TAPS16.astype(np.float32), "f32"), dsp_amd.FftPlan(4096, "f32")
    with pytest.raises(dsp_amd.DspmiError):
        fir.tone_spectrum(fft, dsp_amd.ToneGen(48000.0, np.ones((2, 5))), 4096)                       # five tones
    gen = dsp_amd.ToneGen(48000.0, np.ones((2, 2)))
    with pytest.raises(dsp_amd.DspmiError):
        fir.tone_spectrum(fft, gen, 4096, work=torch.empty((1024,), dtype=torch.uint8, device="cuda"))  # work space too small
    h = torch.zeros((2, 15), dtype=torch.float32, device="cuda")
    with pytest.raises(dsp_amd.DspmiError):
        fir.tone_spectrum(fft, gen, 4096, hist=h, hist_out=h)                                         # history in place
    with pytest.raises(dsp_amd.DspmiError):
        fir.tone_spectrum(dsp_amd.FftPlan(2048, "f32"), gen, 4096)
