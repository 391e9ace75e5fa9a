"""Overlap-save fast convolution against the oracle's direct sum and against the direct kernel."""
import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu

# f32 FFT-domain filtering: the forward + backward transform error is relative to the norm of
# the transformed block (input x taps), not to the output it lands on, so the bound is the FFT
# tolerance of the path, 1e-5, against max(|output row|, |taps| * |input row|) -- a start-up
# transient that is itself ~1e-6 of the input cannot be resolved to 1e-5 of its own size.
TOL_OLS = 1e-5


def rel_l2(got, want):
    den = np.linalg.norm(want)
    return np.linalg.norm(np.asarray(got, np.float64) - want) / den if den > 0 else np.linalg.norm(got)


@pytest.mark.parametrize("nh,fft_n", [(16, 64), (127, 1024), (300, 2048), (1000, 4096), (8191, 16384), (8191, 0), (33, 0)])
@pytest.mark.parametrize("nx,nchan", [(1, 1), (5000, 3), (40000, 2)])
def test_ols_matches_oracle(nh, fft_n, nx, nchan):
    rng = np.random.default_rng(nh + nx)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    plan = dsp_amd.OlsPlan(h, fft_n)
    assert plan.fft_length >= 2 * nh
    y = plan.exec(torch.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.fir_batch_f32(x, h, nx)
    for c in range(nchan):
        scale = np.linalg.norm(h.astype(np.float64)) * np.linalg.norm(x[c].astype(np.float64))
        assert np.linalg.norm(y[c] - want[c]) <= TOL_OLS * max(np.linalg.norm(want[c]), scale), (nh, fft_n, nx, c)


def test_ols_history_and_agreement_with_direct_kernel():
    rng = np.random.default_rng(4)
    nh, nchan, nx = 2049, 3, 30000
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, 2 * nx)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    ols, direct = dsp_amd.OlsPlan(h), dsp_amd.FirPlan(h, "f32")
    first = ols.exec(xd[:, :nx].contiguous())
    hist = xd[:, nx - (nh - 1):nx].contiguous()
    second = ols.exec(xd[:, nx:].contiguous(), hist=hist)
    whole = direct.exec(xd)
    got = torch.cat([first, second], dim=1)
    err = ((got - whole).double().norm(dim=1) / whole.double().norm(dim=1)).max().item()
    assert err <= TOL_OLS
    want = oracle.fir_batch_f32(x[:1, :6000], h, 6000)
    assert rel_l2(got[0, :6000].cpu().numpy(), want[0]) <= TOL_OLS


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
