"""Overlap-save fast convolution against the oracle's direct sum and against the direct kernel."""
import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu

# The bound of the FIR path, 1e-6 relative L2 per 4096-sample output block, holds for every block
# of steady-state output.  The blocks that contain the first nh - 1 outputs are a start-up
# transient: with windowed-sinc taps they are sums over the filter's tiny outer taps only, far
# smaller than the signal, while the rounding error of a transform is relative to what goes
# through it (the whole input block times the whole filter).  Those blocks are held to the same
# ABSOLUTE error as a steady-state block: 1e-6 of |taps| * |input block|, the norm a block of
# steady-state output of white input has.
TOL_FIR = 1e-6
BLK = 4096


def rel_l2(got, want):
    den = np.linalg.norm(want)
    return np.linalg.norm(np.asarray(got, np.float64) - want) / den if den > 0 else np.linalg.norm(got)


def check_blocks(got, want, x, h, tag):
    """got/want/x: one row.  Returns the worst steady-state block error."""
    hn = np.linalg.norm(h.astype(np.float64))
    worst = 0.0
    for lo in range(0, len(want), BLK):
        g, w = got[lo:lo + BLK].astype(np.float64), want[lo:lo + BLK]
        err = np.linalg.norm(g - w)
        if lo >= len(h) - 1:
            assert err <= TOL_FIR * np.linalg.norm(w), (tag, lo, err / np.linalg.norm(w))
            worst = max(worst, err / np.linalg.norm(w))
        else:
            scale = hn * np.linalg.norm(x[lo:lo + BLK].astype(np.float64))
            assert err <= TOL_FIR * max(np.linalg.norm(w), scale), (tag, lo, "start-up", err / scale)
    return worst


@pytest.mark.parametrize("nh,fft_n", [(16, 64), (127, 1024), (300, 2048), (1000, 4096), (8191, 16384), (8191, 0), (33, 0),
                                      (16, 32768), (2049, 32768), (16384, 32768)])
@pytest.mark.parametrize("nx,nchan", [(1, 1), (5000, 3), (40000, 2), (70001, 2)])
def test_ols_matches_oracle(nh, fft_n, nx, nchan):
    rng = np.random.default_rng(nh + nx)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    plan = dsp_amd.OlsPlan(h, fft_n)
    assert plan.fft_length >= 2 * nh
    y = plan.exec(torch.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.fir_batch_f32(x, h, nx)
    for c in range(nchan):
        check_blocks(y[c], want[c], x[c], h, (nh, fft_n, nx, c))


@pytest.mark.parametrize("fft_n", [0, 16384, 32768])
def test_ols_8191_taps_on_the_bench_inputs(fft_n):
    """BASELINE config 4's filter (8191 windowed-sinc taps) on uniform input: every steady-state
    4096-sample block within 1e-6 of the double sum of fft/src/main.c:193-204."""
    nh, nx = 8191, 150000
    rng = np.random.default_rng(8191)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (1, nx)).astype(np.float32)
    try:
        plan = dsp_amd.OlsPlan(h, fft_n)
    except dsp_amd.DspmiError:
        pytest.skip("no %d-point overlap-save kernel" % fft_n)
    y = plan.exec(torch.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.fir_batch_f32(x, h, nx)
    worst = check_blocks(y[0], want[0], x[0], h, (nh, plan.fft_length))
    assert 0 < worst <= TOL_FIR


def test_ols_history_and_agreement_with_direct_kernel(monkeypatch):
    rng = np.random.default_rng(4)
    nh, nchan, nx = 2049, 3, 30000
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, 2 * nx)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    ols = dsp_amd.OlsPlan(h)
    monkeypatch.setenv("DSPMI_FIR_VARIANT", "mfma")   # the plan alone would run overlap-save here too
    direct = dsp_amd.FirPlan(h, "f32")
    assert "mfma" in direct.variant
    first = ols.exec(xd[:, :nx].contiguous())
    hist = xd[:, nx - (nh - 1):nx].contiguous()
    second = ols.exec(xd[:, nx:].contiguous(), hist=hist)
    whole = direct.exec(xd)
    got = torch.cat([first, second], dim=1)
    err = ((got - whole).double().norm(dim=1) / whole.double().norm(dim=1)).max().item()
    assert err <= TOL_FIR
    want = oracle.fir_batch_f32(x[:1, :6000], h, 6000)
    check_blocks(got[0, :6000].cpu().numpy(), want[0], x[0, :6000], h, "history")


def test_ols_long_signal_round_properties():
    """One 2^24-sample signal, 8191 taps: impulse response and linearity (no oracle sweep)."""
    nh, nx = 8191, 1 << 24
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = dsp_amd.OlsPlan(h)
    x = torch.zeros((1, nx), device="cuda")
    x[0, 12345] = 2.0
    y = plan.exec(x).cpu().numpy()[0]
    np.testing.assert_allclose(y[12345:12345 + nh], 2.0 * h, atol=2e-7)
    assert np.abs(y[:12345]).max() < 2e-7 and np.abs(y[12345 + nh:]).max() < 2e-7


@pytest.mark.parametrize("nh", [2049, 5000, 8191])
def test_fir_plan_takes_the_long_block_path_with_history(nh):
    """An f32 FIR plan of more than 2048 taps filters long rows through 32768-point blocks; rows
    cut in chunks with carried history give the row's outputs, ragged lengths included."""
    rng = np.random.default_rng(nh)
    nchan, chunk = 3, 70001
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, 2 * chunk)).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    assert plan.variant == "fir_f32_overlap_save_n32768"
    xd = torch.from_numpy(x).cuda()
    first = plan.exec(xd[:, :chunk].contiguous())
    second = plan.exec(xd[:, chunk:].contiguous(), hist=xd[:, chunk - (nh - 1):chunk].contiguous())
    got = torch.cat([first, second], dim=1).cpu().numpy()
    want = oracle.fir_batch_f32(x, h, 2 * chunk)
    for c in range(nchan):
        check_blocks(got[c], want[c], x[c], h, (nh, c))


@pytest.mark.parametrize("nh", [900, 1024, 2048])
def test_fir_plan_of_a_thousand_taps_by_row_length(nh):
    """From 900 taps an f32 FIR plan holds two overlap-save plans: rows of 65536 samples and more
    go through 32768-sample blocks, shorter rows through 16384-sample blocks; both with carried
    history against the oracle."""
    rng = np.random.default_rng(nh)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    assert plan.variant == "fir_f32_overlap_save_n32768"
    for nchan, chunk in ((2, 70001), (3, 30000)):
        x = rng.uniform(-1, 1, (nchan, 2 * chunk)).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        first = plan.exec(xd[:, :chunk].contiguous())
        second = plan.exec(xd[:, chunk:].contiguous(), hist=xd[:, chunk - (nh - 1):chunk].contiguous())
        got = torch.cat([first, second], dim=1).cpu().numpy()
        want = oracle.fir_batch_f32(x, h, 2 * chunk)
        for c in range(nchan):
            check_blocks(got[c], want[c], x[c], h, (nh, chunk, c))


@pytest.mark.parametrize("nh,nchan,nx", [(900, 5, 300), (900, 5, 3000), (1536, 4, 20000), (2049, 5, 1000), (2049, 3, 40000),
                                         (8191, 5, 300), (8191, 5, 4096), (8191, 3, 40000), (8192, 2, 16385),
                                         (12000, 2, 20000), (16384, 2, 3000)])
def test_long_filters_on_short_rows(nh, nchan, nx):
    """Rows shorter than two 32768-sample blocks: from 900 taps an f32 FIR plan filters them through
    16384-sample blocks (one real block per 8192-point transform; 32768-sample blocks beyond 8192
    taps) once a row holds enough work, and through the direct kernel below that; either way the
    oracle's outputs, with carried history."""
    rng = np.random.default_rng(nh + nx)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    assert "overlap_save" in plan.variant
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)
    got = plan.exec(torch.from_numpy(x).cuda(), hist=torch.from_numpy(hist).cuda()).cpu().numpy()
    full = np.concatenate([hist, x], axis=1)
    want = oracle.fir_batch_f32(full, h, full.shape[1])[:, nh - 1:]
    for c in range(nchan):
        for lo in range(0, nx, BLK):
            w = want[c, lo:lo + BLK]
            assert np.linalg.norm(got[c, lo:lo + BLK] - w) <= TOL_FIR * np.linalg.norm(w), (nh, nx, c, lo)
    # and from zero state (start-up blocks are held to the absolute bound, see check_blocks)
    got0 = plan.exec(torch.from_numpy(x).cuda()).cpu().numpy()
    want0 = oracle.fir_batch_f32(x, h, nx)
    for c in range(nchan):
        check_blocks(got0[c], want0[c], x[c], h, (nh, nx, c))


@pytest.mark.parametrize("nh", [306, 511, 800])
@pytest.mark.parametrize("nchan,nx", [(9, 100), (7, 256), (5, 1500), (4, 4096), (3, 9000), (2, 40000)])
def test_mid_length_filters_by_row_length(nh, nchan, nx):
    """306 .. 899 taps: an f32 FIR plan holds the transform of four times the tap count and the
    shortest one that fits, and a call runs whichever costs fewer transforms for its row length
    (rows of a few hundred samples included, which the direct kernel's 4096-output tiles serve
    badly); the oracle's outputs either way, with carried history and from zero state."""
    rng = np.random.default_rng(nh + nx)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    assert "overlap_save" in plan.variant
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)
    got = plan.exec(torch.from_numpy(x).cuda(), hist=torch.from_numpy(hist).cuda()).cpu().numpy()
    full = np.concatenate([hist, x], axis=1)
    want = oracle.fir_batch_f32(full, h, full.shape[1])[:, nh - 1:]
    for c in range(nchan):
        for lo in range(0, nx, BLK):
            w = want[c, lo:lo + BLK]
            assert np.linalg.norm(got[c, lo:lo + BLK] - w) <= TOL_FIR * np.linalg.norm(w), (nh, nx, c, lo)
    got0 = plan.exec(torch.from_numpy(x).cuda()).cpu().numpy()
    want0 = oracle.fir_batch_f32(x, h, nx)
    for c in range(nchan):
        check_blocks(got0[c], want0[c], x[c], h, (nh, nx, c))


@pytest.mark.parametrize("nh,fft_n", [(127, 1024), (1000, 4096), (8191, 16384), (8191, 32768)])
def test_fused_kernels_agree_with_the_five_kernel_composition(nh, fft_n, monkeypatch):
    """DSPMI_OLS_COMPOSED=1 runs the same arithmetic as five kernels through HBM (gather, batched
    transform, multiply, batched transform, scatter): the A/B reference of the one-kernel forms."""
    rng = np.random.default_rng(nh + fft_n)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (2, 3 * fft_n + 777)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    fused = dsp_amd.OlsPlan(h, fft_n)
    monkeypatch.setenv("DSPMI_OLS_COMPOSED", "1")
    composed = dsp_amd.OlsPlan(h, fft_n)
    monkeypatch.delenv("DSPMI_OLS_COMPOSED")
    a, b = fused.exec(xd).cpu().numpy(), composed.exec(xd).cpu().numpy()
    want = oracle.fir_batch_f32(x, h, x.shape[1])
    for c in range(2):
        check_blocks(a[c], want[c], x[c], h, ("fused", nh, fft_n, c))
        check_blocks(b[c], want[c], x[c], h, ("composed", nh, fft_n, c))


@pytest.mark.parametrize("nh,nx,nchan,with_hist", [(8191, 150000, 2, False), (8191, 70001, 3, True), (8192, 90000, 2, True),
                                                   (2049, 100000, 2, True), (16, 40000, 2, False), (16384, 120001, 1, True),
                                                   (5000, 3 * 27769, 2, False), (8191, 1, 1, False), (8191, 8190, 1, True)])
@pytest.mark.parametrize("grid", [0, 3])
def test_long_blocks_one_real_block_per_transform(nh, nx, nchan, with_hist, grid, monkeypatch):
    """32768-sample blocks run one real block through a 16384-point complex transform (even samples
    as real parts, odd samples as imaginary parts; the default), or two blocks through a
    32768-point transform (DSPMI_OLS_REAL=0).  Both against the oracle per 4096-sample block where
    it can be swept, and against each other: odd and even tap counts (an even count starts blocks
    on odd samples, which takes the kernel's element-by-element path, as do rows of odd stride),
    carried history, ragged ends, rows shorter than the filter, a single sample; with one block
    per workgroup and with three workgroups walking all blocks (DSPMI_OLS_GRID)."""
    rng = np.random.default_rng(nh + nx + grid)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32) if with_hist else None
    xd = torch.from_numpy(x).cuda()
    hd = torch.from_numpy(hist).cuda() if with_hist else None
    one = dsp_amd.OlsPlan(h, 32768)
    monkeypatch.setenv("DSPMI_OLS_REAL", "0")
    two = dsp_amd.OlsPlan(h, 32768)
    monkeypatch.delenv("DSPMI_OLS_REAL")
    if grid:
        monkeypatch.setenv("DSPMI_OLS_GRID", str(grid))
    a = one.exec(xd, hist=hd).cpu().numpy()
    b = two.exec(xd, hist=hd).cpu().numpy()
    scale = np.linalg.norm(h.astype(np.float64)) * np.linalg.norm(x.astype(np.float64), axis=1)
    for c in range(nchan):
        # two roundings apart at most: each is within 1e-6 of the double sum
        assert np.linalg.norm(a[c].astype(np.float64) - b[c]) <= 2 * TOL_FIR * max(scale[c], np.linalg.norm(b[c].astype(np.float64)))
    if nh <= 8192:
        full = np.concatenate([hist, x], axis=1) if with_hist else x
        want = oracle.fir_batch_f32(full, h, full.shape[1])[:, full.shape[1] - nx:]
        for c in range(nchan):
            if with_hist:
                # every block is steady state once the history is there: hold all of them to the block bound
                for lo in range(0, nx, BLK):
                    w = want[c, lo:lo + BLK]
                    assert np.linalg.norm(a[c, lo:lo + BLK] - w) <= TOL_FIR * np.linalg.norm(w), (nh, nx, c, lo)
            else:
                check_blocks(a[c], want[c], x[c], h, (nh, nx, c))


@pytest.mark.parametrize("nh", [2049, 8191, 16384])
@pytest.mark.parametrize("grid", [1, 4, 7])
def test_long_blocks_with_the_next_pair_requested_early(nh, grid, monkeypatch):
    """The kernel that runs block pairs (DSPMI_OLS_REAL=0).
    32768-point blocks, few workgroups (DSPMI_OLS_GRID) so that each walks several block pairs:
    from its second interior pair on a workgroup takes its input from the requests it issued
    ahead of the previous pair's stores (even block through LDS, odd block through registers).
    Checked against the oracle per block, and bit for bit against the same launch with the early
    requests switched off (DSPMI_OLS_PREFETCH=0) -- the arithmetic is the same."""
    rng = np.random.default_rng(nh + grid)
    L = 32768 - (nh - 1)
    nchan = 2
    nx = 9 * L + 12345                      # ten blocks per row: five pairs, the last one ragged
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)
    xd, hd = torch.from_numpy(x).cuda(), torch.from_numpy(hist).cuda()
    monkeypatch.setenv("DSPMI_OLS_REAL", "0")
    plan = dsp_amd.OlsPlan(h, 32768)
    monkeypatch.setenv("DSPMI_OLS_GRID", str(grid))
    early = plan.exec(xd, hist=hd).cpu().numpy()
    monkeypatch.setenv("DSPMI_OLS_PREFETCH", "0")
    plain = plan.exec(xd, hist=hd).cpu().numpy()
    assert np.array_equal(early, plain)
    if nh <= 8191:
        full = np.concatenate([hist, x], axis=1)
        want = oracle.fir_batch_f32(full, h, nh - 1 + nx)[:, nh - 1:]
        for c in range(nchan):
            check_blocks(early[c], want[c], x[c], h, (nh, grid, c))

