"""Rows and batches whose element and byte offsets pass 2^31 and 2^32: one channel of
2^31 + 5000 samples through every FIR kernel, and a c2c batch of more than 2^32 complex points.
Checked on windows around the offsets where a 32-bit byte or element index would wrap."""
import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu


def rel_l2(got, want):
    return float(np.linalg.norm(got - want) / np.linalg.norm(want))


@pytest.mark.parametrize("nh,path,nx", [(16, "direct", (1 << 31) + 5000), (127, "overlap_save", (1 << 31) + 5000),
                                        (127, "direct", (1 << 31) + 5000), (127, "bf16", (1 << 31) + 5000),
                                        (127, "f16", (1 << 31) + 5000),
                                        # the longest row the 32-bit sample count can express
                                        (127, "overlap_save", (1 << 32) - 3), (16, "direct", (1 << 32) - 3),
                                        (127, "f16", (1 << 32) - 3),
                                        # 32768-sample blocks, one per 16384-point transform
                                        (8191, "overlap_save", (1 << 31) + 5000), (8191, "overlap_save", (1 << 32) - 3)])
def test_fir_row_longer_than_2_31_samples(nh, path, nx, monkeypatch):
    if torch.cuda.mem_get_info()[0] < 60 << 30:
        pytest.skip("needs 60 GiB of free device memory")
    monkeypatch.setenv("DSPMI_FIR_VARIANT", {"direct": "mfma", "overlap_save": "ols", "bf16": "bf16", "f16": "f16"}[path])
    h = (np.asarray(oracle.TAPS16) if nh == 16 else oracle.windowed_sinc_taps(nh)).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    assert ("overlap_save" in plan.variant) == (path == "overlap_save") and ("bf16" in plan.variant) == (path == "bf16") and ("f16x2" in plan.variant) == (path == "f16")
    g = torch.Generator(device="cuda").manual_seed(nh)
    x = torch.rand((1, nx), device="cuda", generator=g) * 2 - 1
    y = plan.exec(x)
    torch.cuda.synchronize()
    width = 6000
    for s in (0, (1 << 29) - 3000, (1 << 30) - 3000, (1 << 31) - 3000, nx - (1 << 20), nx - width):
        lo = max(0, s - (nh - 1))
        win = x[0, lo:s + width].cpu().numpy()
        want = oracle.fir_batch_f32(win[None, :], h, len(win))[0][s - lo:]
        got = y[0, s:s + width].cpu().numpy()
        assert rel_l2(got, want) <= 1e-6, (s, rel_l2(got, want))
    del x, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nx", [1000, (1 << 29) - 7])
def test_fir_long_zero_tail_behind_a_short_row(nx, monkeypatch):
    """Outputs far past the input (nxx > 2^30 samples behind a row of at most 2^29): tiles whose
    window starts beyond 2^30 samples have byte offsets that do not fit 32 bits; they must
    produce zeros, not samples from the start of the row."""
    if torch.cuda.mem_get_info()[0] < 16 << 30:
        pytest.skip("needs 16 GiB of free device memory")
    monkeypatch.setenv("DSPMI_FIR_VARIANT", "mfma")
    nh = 16
    h = np.asarray(oracle.TAPS16, dtype=np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    nxx = (1 << 31) + 4096 + 5
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand((1, nx), device="cuda", generator=g) * 2 - 1
    y = torch.full((1, nxx), 7.0, device="cuda")
    plan.exec(x, nxx, y=y)
    torch.cuda.synchronize()
    head = min(nx, 5000)
    want = oracle.fir_batch_f32(x[:, :head].cpu().numpy(), h, head)[0]
    assert rel_l2(y[0, :head].cpu().numpy(), want) <= 1e-6
    tail = x[0, nx - 64:].cpu().numpy()
    want_tail = oracle.fir_batch_f32(tail[None, :], h, 64 + nh)[0][64:]
    np.testing.assert_allclose(y[0, nx:nx + nh].cpu().numpy(), want_tail, rtol=0, atol=1e-6)
    assert not bool(y[0, nx + nh:].any())     # everything past the filter's reach is an empty sum
    del x, y
    torch.cuda.empty_cache()


def test_c2c_batch_of_more_than_2_32_points():
    if torch.cuda.mem_get_info()[0] < 80 << 30:
        pytest.skip("needs 80 GiB of free device memory")
    n = 1024
    batch = (1 << 22) + 3                     # 2^32 + 3072 complex points, 32 GiB each way
    g = torch.Generator(device="cuda").manual_seed(1)
    z = torch.view_as_complex(torch.rand((batch, n, 2), device="cuda", generator=g) * 2 - 1)
    plan = dsp_amd.FftPlan(n, "f32")
    out = plan.c2c(z)
    torch.cuda.synchronize()
    for f in (0, (1 << 18) - 1, 1 << 18, (1 << 21), (1 << 22) - 1, 1 << 22, batch - 1):
        want = oracle.dft(z[f].cpu().numpy().astype(np.complex128), -1, "direct")
        assert rel_l2(out[f].cpu().numpy(), want) <= 1e-5, (f, rel_l2(out[f].cpu().numpy(), want))
    del z, out
    torch.cuda.empty_cache()
