"""Parity of the HIP FFT path against the oracle, through the C ABI.

"Parity unpinned": the reference's transforms are FFTW3's, which is not
available; the checker is the oracle's long-double DFT (committed vectors) and
its double radix-2.  f32 tolerance: relative L2 <= 1e-5 per frame.
"""
import os

import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu

TOL_FFT = 1e-5   # f32, relative L2 per frame
TOL_F64 = 1e-13  # double wrappers, relative L2


def rel_l2(got, want):
    den = np.linalg.norm(want)
    return np.linalg.norm(got - want) / den if den > 0 else np.linalg.norm(got)


POW2 = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]


@pytest.mark.parametrize("n", POW2)
def test_c2c_f32_against_oracle(n):
    rng = np.random.default_rng(n)
    batch = 7 if n <= 4096 else 3
    z = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))).astype(np.complex64)
    plan = dsp_amd.FftPlan(n, "f32")
    zd = torch.from_numpy(z).cuda()
    fwd = plan.c2c(zd, dsp_amd.FORWARD).cpu().numpy()
    bwd = plan.c2c(zd, dsp_amd.BACKWARD).cpu().numpy()
    method = "direct" if n <= 2048 else "radix2"
    for f in range(batch):
        zz = z[f].astype(np.complex128)
        assert rel_l2(fwd[f], oracle.dft(zz, -1, method)) <= TOL_FFT, (plan.variant, f)
        assert rel_l2(bwd[f], oracle.dft(zz, +1, method)) <= TOL_FFT, (plan.variant, f)


@pytest.mark.parametrize("n", [8, 64, 1024, 4096])
def test_c2c_golden_vectors(golden_dir, n):
    g = np.load(os.path.join(golden_dir, "fft_cases.npz"))
    z = g["in_%d" % n]
    plan = dsp_amd.FftPlan(n, "f32")
    zd = torch.from_numpy(z.astype(np.complex64)[None, :]).cuda()
    assert rel_l2(plan.c2c(zd, dsp_amd.FORWARD).cpu().numpy()[0], g["fwd_%d" % n]) <= TOL_FFT
    assert rel_l2(plan.c2c(zd, dsp_amd.BACKWARD).cpu().numpy()[0], g["bwd_%d" % n]) <= TOL_FFT
    # double wrappers
    assert rel_l2(dsp_amd.dft(z), g["fwd_%d" % n]) <= TOL_F64
    assert rel_l2(dsp_amd.idft(z), g["bwd_%d" % n]) <= TOL_F64


def test_c2c_in_place_and_odd_batches():
    n = 1024
    plan = dsp_amd.FftPlan(n, "f32")
    for batch in (1, 3, 4, 5, 1000, 4097):
        g = torch.Generator(device="cuda").manual_seed(batch)
        z = torch.view_as_complex(torch.rand((batch, n, 2), device="cuda", generator=g) * 2 - 1)
        want = plan.c2c(z)
        buf = z.clone()
        plan.c2c(buf, out=buf)
        assert torch.equal(buf, want)
        ref = torch.fft.fft(z.to(torch.complex128))
        err = ((want.to(torch.complex128) - ref).abs().pow(2).sum(1).sqrt() / ref.abs().pow(2).sum(1).sqrt()).max()
        assert err.item() <= TOL_FFT, batch


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n", [2, 8, 16, 32, 64, 128])
def test_short_frames_ragged_batches(n, dtype):
    """Frames shorter than 16 lanes travel through LDS as one contiguous run per wave: batches
    that end inside a wave's run, in place and out of place, nothing written past the batch."""
    plan = dsp_amd.FftPlan(n, dtype)
    ct = torch.complex64 if dtype == "f32" else torch.complex128
    tol = TOL_FFT if dtype == "f32" else TOL_F64
    for batch in (1, 3, 63, 64, 65, 1000, 4099):
        g = torch.Generator(device="cuda").manual_seed(batch + n)
        z = torch.view_as_complex(torch.rand((batch, n, 2), device="cuda", generator=g, dtype=torch.float64) * 2 - 1).to(ct)
        big = torch.full((batch + 70, n), 7.0 + 0j, dtype=ct, device="cuda")
        plan.c2c(z, out=big[:batch])
        assert bool((big[batch:] == 7.0).all()), batch
        buf = z.clone()
        plan.c2c(buf, out=buf)
        assert torch.equal(buf, big[:batch])
        ref = torch.fft.fft(z.to(torch.complex128))
        err = ((big[:batch].to(torch.complex128) - ref).abs().pow(2).sum(1).sqrt() / ref.abs().pow(2).sum(1).sqrt()).max()
        assert err.item() <= tol, (batch, err.item())


def test_known_answers():
    n = 1024
    plan = dsp_amd.FftPlan(n, "f32")
    x = torch.zeros((3, n), dtype=torch.complex64, device="cuda")
    x[0, 0] = 1                                     # impulse -> all ones
    x[1, :] = 1                                     # constant -> n at bin 0
    k = 37
    t = torch.arange(n, device="cuda")
    x[2] = torch.exp(2j * np.pi * k * t / n).to(torch.complex64) * 0.5   # single bin tone
    y = plan.c2c(x).cpu().numpy()
    assert np.allclose(y[0], 1.0, atol=1e-6)
    assert abs(y[1, 0] - n) < 1e-3 and np.abs(y[1, 1:]).max() < 1e-3
    assert abs(y[2, k] - 0.5 * n) < 1e-2 and np.abs(np.delete(y[2], k)).max() < 2e-2


def test_round_trip_and_parseval_at_scale():
    """Properties that do not need the oracle, on a batch the oracle could not sweep."""
    n, batch = 1024, 1 << 16
    plan = dsp_amd.FftPlan(n, "f32")
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.view_as_complex(torch.rand((batch, n, 2), device="cuda", generator=g) * 2 - 1)
    y = plan.c2c(z, dsp_amd.FORWARD)
    back = plan.c2c(y, dsp_amd.BACKWARD) / n     # c2c backward is unnormalised (fftc/main.c:164 divides by nx)
    err = ((back - z).abs().pow(2).sum(1).sqrt() / z.abs().pow(2).sum(1).sqrt()).max()
    assert err.item() <= TOL_FFT
    e_t = z.abs().double().pow(2).sum(1)
    e_f = y.abs().double().pow(2).sum(1) / n
    assert ((e_t - e_f).abs() / e_t).max().item() < 1e-5
    # spot frames against the oracle
    for f in (0, batch // 2, batch - 1):
        assert rel_l2(y[f].cpu().numpy(), oracle.dft(z[f].cpu().numpy().astype(np.complex128), -1, "direct")) <= TOL_FFT


@pytest.mark.parametrize("n", [4, 16, 1024, 4096, 8192])
def test_r2c_c2r_f32(n):
    rng = np.random.default_rng(n + 1)
    x = rng.uniform(-1, 1, (5, n)).astype(np.float32)
    plan = dsp_amd.FftPlan(n, "f32")
    xd = torch.from_numpy(x).cuda()
    bins = plan.r2c(xd)
    got = bins.cpu().numpy()
    for f in range(5):
        assert rel_l2(got[f], oracle.fft_r2c(x[f].astype(np.float64))) <= TOL_FFT
    back = plan.c2r(bins).cpu().numpy() / n
    for f in range(5):
        assert rel_l2(back[f], x[f]) <= TOL_FFT
        want = oracle.fft_c2r(got[f].astype(np.complex128), n)
        assert rel_l2(back[f] * n, want) <= TOL_FFT


@pytest.mark.parametrize("n,batch", [(2, 3), (4, 70), (8, 5), (32, 257), (128, 33), (2048, 37), (16384, 3)])
def test_c2r_f32_against_oracle(n, batch):
    """Arbitrary bins (not the transform of a real frame): the imaginary parts of bins 0 and n/2
    are ignored, as fftw's c2r does (util/filter/filter.c:98-103); ragged batches."""
    rng = np.random.default_rng(n + batch)
    bins = (rng.uniform(-1, 1, (batch, n // 2 + 1)) + 1j * rng.uniform(-1, 1, (batch, n // 2 + 1))).astype(np.complex64)
    plan = dsp_amd.FftPlan(n, "f32")
    bd = torch.from_numpy(bins).cuda()
    keep = bd.clone()
    got = plan.c2r(bd).cpu().numpy()
    assert torch.equal(bd, keep)                          # the input is left alone
    assert got.shape == (batch, n)
    for f in sorted({0, 1, batch // 2, batch - 1}):
        want = oracle.fft_c2r(bins[f].astype(np.complex128), n)
        assert rel_l2(got[f], want) <= TOL_FFT, (f, rel_l2(got[f], want))
    # output over the input buffer (frames of bins are longer than frames of samples)
    flat = torch.zeros(batch * (n // 2 + 1) * 2, dtype=torch.float32, device="cuda")
    src = torch.view_as_complex(flat.view(batch, n // 2 + 1, 2))
    src.copy_(bd)
    out = flat[: batch * n].view(batch, n)
    plan.c2r(src, out=out)
    assert rel_l2(out.cpu().numpy(), got) <= 1e-6


@pytest.mark.parametrize("n", [64, 2048])
def test_r2c_over_its_own_input_and_shifted_c2c(n):
    """Bins written over the samples they come from (frames of bins are longer than frames of
    samples, so this cannot run as one kernel); a c2c whose output is the input shifted by a
    frame is an argument error."""
    rng = np.random.default_rng(n)
    batch, nb = 9, n // 2 + 1
    x = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    plan = dsp_amd.FftPlan(n, "f32")
    want = plan.r2c(torch.from_numpy(x).cuda()).cpu().numpy()
    flat = torch.zeros(batch * nb * 2, dtype=torch.float32, device="cuda")
    src = flat[: batch * n].view(batch, n)
    src.copy_(torch.from_numpy(x))
    out = torch.view_as_complex(flat.view(batch, nb, 2))
    plan.r2c(src, out=out)
    assert rel_l2(out.cpu().numpy(), want) <= 1e-6
    z = torch.view_as_complex(torch.rand(batch + 1, n, 2, device="cuda"))
    with pytest.raises(dsp_amd.DspmiError, match="overlap"):
        plan.c2c(z[:batch], out=z[1:])


def test_double_wrappers(golden_dir):
    g = np.load(os.path.join(golden_dir, "fft_cases.npz"))
    x = g["filter_x"]
    assert rel_l2(dsp_amd.fft_r2c(x), g["r2c_4096"]) <= TOL_F64
    assert rel_l2(dsp_amd.filter(x, g["filter_gains"]), g["filter_y"]) <= TOL_F64
    assert rel_l2(dsp_amd.filter(x, np.ones(2049)), x) <= TOL_F64
    z = g["in_1024"]
    assert rel_l2(dsp_amd.fft_mag(z), oracle.ps_abs(g["fwd_1024"])) <= TOL_F64


# ---- lengths that are not powers of two (Bluestein) ------------------------------

@pytest.mark.parametrize("n", [3, 5, 12, 100, 480, 482, 1000, 3000])
def test_c2c_any_length_f32(n):
    rng = np.random.default_rng(n)
    z = (rng.uniform(-1, 1, (4, n)) + 1j * rng.uniform(-1, 1, (4, n))).astype(np.complex64)
    plan = dsp_amd.FftPlan(n, "f32")
    assert "bluestein" in plan.variant
    zd = torch.from_numpy(z).cuda()
    fwd = plan.c2c(zd, dsp_amd.FORWARD).cpu().numpy()
    bwd = plan.c2c(zd, dsp_amd.BACKWARD).cpu().numpy()
    for f in range(4):
        zz = z[f].astype(np.complex128)
        assert rel_l2(fwd[f], oracle.dft(zz, -1, "direct")) <= TOL_FFT
        assert rel_l2(bwd[f], oracle.dft(zz, +1, "direct")) <= TOL_FFT


def test_demo_lengths_double_wrappers(golden_dir):
    """480 (fftc/main.c:137) and 482 (fir/src/main.c:154) through dft/idft."""
    g = np.load(os.path.join(golden_dir, "fft_cases.npz"))
    for n in (480, 482):
        z = g["in_%d" % n]
        assert rel_l2(dsp_amd.dft(z), g["fwd_%d" % n]) <= 1e-12
        assert rel_l2(dsp_amd.idft(z), g["bwd_%d" % n]) <= 1e-12
    # the fftc demo's own data: forward, then backward / nx gives the input back
    x = g["fftc_x"]
    y = dsp_amd.dft(x)
    assert rel_l2(y, g["fftc_y"]) <= 1e-12
    assert rel_l2(dsp_amd.idft(y) / 480, x) <= 1e-12


@pytest.mark.parametrize("n", [6, 9, 100, 481, 1000, 3000])
def test_real_transforms_any_length(n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, n)
    bins = dsp_amd.fft_r2c(x)
    assert rel_l2(bins, oracle.fft_r2c(x)) <= 1e-12
    gains = rng.uniform(0, 1, n // 2 + 1)
    assert rel_l2(dsp_amd.filter(x, gains), oracle.filter(x, gains)) <= 1e-12
    assert rel_l2(dsp_amd.filter(x, np.ones(n // 2 + 1)), x) <= 1e-12


@pytest.mark.parametrize("dtype,n", [("f32", 4096), ("f32", 16), ("f32", 1024), ("f32", 16384), ("f32", 8), ("f32", 480),
                                     ("f32", 1001), ("f64", 4096), ("f64", 482)])
def test_batched_filter_op(dtype, n):
    """`filter` (util/filter/filter.c:73-105) as a batched device op, in place and out of place."""
    rng = np.random.default_rng(n)
    batch = 5
    x = rng.uniform(-1, 1, (batch, n))
    gains = np.where(np.arange(n // 2 + 1) < n // 8, 1.0, np.where(np.arange(n // 2 + 1) < n // 6, 0.5, 0.0))
    rt = torch.float32 if dtype == "f32" else torch.float64
    plan = dsp_amd.FftPlan(n, dtype)
    xd = torch.from_numpy(x).to(rt).cuda()
    gd = torch.from_numpy(gains).to(rt).cuda()
    got = plan.filter(xd, gd).cpu().numpy()
    buf = xd.clone()
    plan.filter(buf, gd, out=buf)
    assert np.array_equal(buf.cpu().numpy(), got)
    tol = TOL_FFT if dtype == "f32" else TOL_F64
    for f in range(batch):
        want = oracle.filter(xd[f].double().cpu().numpy(), gains)
        assert rel_l2(got[f], want) <= tol, (f, rel_l2(got[f], want))


@pytest.mark.parametrize("n,batch", [(1024, 1), (1024, 2), (2048, 7), (256, 64), (32, 33)])
def test_filter_one_kernel_form_matches_the_composition(n, batch, monkeypatch):
    """Power-of-two f32 frames take the one-kernel form (two real frames per complex transform, the
    gains as a real even spectrum); the r2c / gain / c2r / scale composition is the same arithmetic."""
    rng = np.random.default_rng(n + batch)
    x = torch.from_numpy(rng.uniform(-1, 1, (batch, n)).astype(np.float32)).cuda()
    gains = torch.from_numpy(rng.uniform(0, 2, n // 2 + 1).astype(np.float32)).cuda()
    plan = dsp_amd.FftPlan(n, "f32")
    fused = plan.filter(x, gains).cpu().numpy()
    monkeypatch.setenv("DSPMI_FILTER_COMPOSED", "1")
    composed = plan.filter(x, gains).cpu().numpy()
    for f in range(batch):
        want = oracle.filter(x[f].double().cpu().numpy(), gains.double().cpu().numpy())
        assert rel_l2(fused[f], want) <= TOL_FFT, (f, rel_l2(fused[f], want))
        assert rel_l2(composed[f], want) <= TOL_FFT


def test_lengths_no_kernel_covers_are_refused():
    """Plan creation answers -ENOSYS (with a message) instead of planning something it cannot run:
    beyond 2^24 points (2^22 when the length is not a power of two)."""
    import ctypes as C
    L = dsp_amd.lib()
    for n, prec in (((1 << 24) + 2, 0), (1 << 25, 1), ((1 << 22) + 1, 0), ((1 << 31) - 100, 0), ((1 << 31) + 1, 1)):
        h = C.c_void_p()
        rc = L.dspmi_fft_create(C.byref(h), n, prec, -1)
        assert rc == -38 and not h.value, (n, prec, rc)
        assert b"fft_create" in L.dspmi_last_error()
    h = C.c_void_p()
    assert L.dspmi_fft_create(C.byref(h), 0, 0, -1) == -22
    assert L.dspmi_fft_create(C.byref(h), 1024, 7, -1) == -22
    assert L.dspmi_fft_create(None, 1024, 0, -1) == -22
    # the longest lengths that do exist
    for n, dtype in ((16384, "f32"), (8192, "f64"), (8191, "f32")):
        plan = dsp_amd.FftPlan(n, dtype)
        z = torch.view_as_complex(torch.rand(2, n, 2, device="cuda", dtype=torch.float32 if dtype == "f32" else torch.float64))
        got = plan.c2c(z).cpu().numpy()
        want = np.fft.fft(z.cpu().numpy().astype(np.complex128), axis=1)
        assert rel_l2(got, want) <= (TOL_FFT if dtype == "f32" else TOL_F64)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n,batch", [(5, 3), (9, 70), (100, 33), (129, 5), (300, 7), (480, 17), (482, 4), (1000, 9), (3000, 5), (8191, 2),
                                     (12000, 3), (16384 - 5, 2)])
def test_any_length_one_kernel_form(n, batch, dtype, monkeypatch):
    """Lengths that are not powers of two run Bluestein's convolution in one kernel (f32 up to
    M = 32768 points of convolution, f64 for M = 512 .. 16384); the five separate kernels are the
    same arithmetic.  Both directions, ragged batches, in place."""
    TOL_FFT = 1e-5 if dtype == "f32" else TOL_F64   # noqa: N806 - shadows the module's f32 tolerance for this dtype
    rng = np.random.default_rng(n * 7 + batch)
    z = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))).astype(np.complex64 if dtype == "f32" else np.complex128)
    plan = dsp_amd.FftPlan(n, dtype)
    if dtype == "f32" or 256 < n <= 8192:
        assert "one_kernel" in plan.variant
    zd = torch.from_numpy(z).cuda()
    fwd = plan.c2c(zd, dsp_amd.FORWARD)
    bwd = plan.c2c(zd, dsp_amd.BACKWARD)
    buf = zd.clone()
    plan.c2c(buf, dsp_amd.FORWARD, out=buf)
    assert torch.equal(buf, fwd)
    monkeypatch.setenv("DSPMI_BLUESTEIN_COMPOSED", "1")
    fwd_c = plan.c2c(zd, dsp_amd.FORWARD).cpu().numpy()
    monkeypatch.delenv("DSPMI_BLUESTEIN_COMPOSED")
    fwd, bwd = fwd.cpu().numpy(), bwd.cpu().numpy()
    for f in sorted({0, batch // 2, batch - 1}):
        zz = z[f].astype(np.complex128)
        want = np.fft.fft(zz) if n > 2048 else oracle.dft(zz, -1, "direct")
        assert rel_l2(fwd[f], want) <= TOL_FFT, (f, rel_l2(fwd[f], want))
        assert rel_l2(fwd_c[f], want) <= TOL_FFT
        assert rel_l2(bwd[f], np.conj(np.fft.fft(np.conj(zz)))) <= TOL_FFT


@pytest.mark.parametrize("dtype,n,batch", [("f32", 6, 5), ("f32", 480, 9), ("f32", 481, 4), ("f32", 1000, 33), ("f32", 6000, 3),
                                           ("f64", 482, 7), ("f64", 1000, 2)])
def test_batched_real_transforms_any_length(dtype, n, batch):
    """r2c / c2r of lengths that are not powers of two, batched on the device: even lengths through
    the half-length complex plan, odd ones through the full complex transform."""
    rng = np.random.default_rng(n + batch)
    rt = torch.float32 if dtype == "f32" else torch.float64
    tol = TOL_FFT if dtype == "f32" else TOL_F64
    x = rng.uniform(-1, 1, (batch, n))
    plan = dsp_amd.FftPlan(n, dtype)
    xd = torch.from_numpy(x).to(rt).cuda()
    bins = plan.r2c(xd)
    back = plan.c2r(bins).cpu().numpy() / n
    got = bins.cpu().numpy()
    for f in sorted({0, batch // 2, batch - 1}):
        xf = xd[f].double().cpu().numpy()
        assert rel_l2(got[f], oracle.fft_r2c(xf)) <= tol, (f, rel_l2(got[f], oracle.fft_r2c(xf)))
        assert rel_l2(back[f], xf) <= tol


# ---- lengths beyond the longest single-workgroup kernel: n = n1 * n2 in two batched passes ----

@pytest.mark.parametrize("dtype,n,batch", [("f32", 65536, 3), ("f32", 1 << 17, 2), ("f32", 1 << 20, 1),
                                           ("f64", 32768, 3), ("f64", 65536, 2), ("f64", 1 << 20, 1)])
def test_c2c_long_powers_of_two(dtype, n, batch):
    rng = np.random.default_rng(n + batch)
    ct = np.complex128 if dtype == "f64" else np.complex64
    z = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))).astype(ct)
    plan = dsp_amd.FftPlan(n, dtype)
    assert "_as_" in plan.variant
    zd = torch.from_numpy(z).cuda()
    tol = TOL_F64 if dtype == "f64" else TOL_FFT
    for sign in (dsp_amd.FORWARD, dsp_amd.BACKWARD):
        got = plan.c2c(zd, sign).cpu().numpy()
        for f in range(batch):
            want = oracle.dft(z[f].astype(np.complex128), sign, "radix2")
            assert rel_l2(got[f], want) <= tol, (dtype, n, sign, f, rel_l2(got[f], want))
    # in place, and the round trip n * identity
    buf = zd.clone()
    plan.c2c(buf, dsp_amd.FORWARD, out=buf)
    plan.c2c(buf, dsp_amd.BACKWARD, out=buf)
    assert rel_l2(buf.cpu().numpy() / n, z) <= 4 * tol


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n,batch", [(1 << 14, 9), (1 << 15, 4), (1 << 16, 5), (1 << 17, 3), (1 << 18, 2), (1 << 19, 2), (1 << 20, 3),
                                     (1 << 21, 3), (1 << 22, 2), (1 << 23, 1), (1 << 24, 1)])
def test_c2c_long_transforms_in_two_or_three_passes(n, batch, dtype, monkeypatch):
    """Beyond the single-kernel lengths: two passes over memory up to 2^20 points, the transposes
    folded into the transforms' loads and stores, and three such passes (n = t1 t2 t3, factors of
    128 or 256 points) from 2^21 to 2^24; DSPMI_FFT_FIVE_PASS=1 takes the five-pass route.  Both
    against the oracle on one frame, forward and backward, and against each other on every frame."""
    ct, tol, same = (np.complex64, TOL_FFT, 1e-6) if dtype == "f32" else (np.complex128, TOL_F64, 1e-14)
    rng = np.random.default_rng(n)
    z = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))).astype(ct)
    zd = torch.from_numpy(z).cuda()
    plan = dsp_amd.FftPlan(n, dtype)
    for sign in (dsp_amd.FORWARD, dsp_amd.BACKWARD):
        monkeypatch.delenv("DSPMI_FFT_FIVE_PASS", raising=False)
        two = plan.c2c(zd, sign).cpu().numpy()
        monkeypatch.setenv("DSPMI_FFT_FIVE_PASS", "1")
        five = plan.c2c(zd, sign).cpu().numpy()
        want = oracle.dft(z[batch - 1].astype(np.complex128), sign, "radix2")
        assert rel_l2(two[batch - 1], want) <= tol and rel_l2(five[batch - 1], want) <= tol
        for f in range(batch):
            assert rel_l2(two[f], five[f].astype(np.complex128)) <= same, (n, sign, f)
    # in place, and a batch that does not fill the last group of work frames
    monkeypatch.delenv("DSPMI_FFT_FIVE_PASS", raising=False)
    buf = zd.clone()
    plan.c2c(buf, out=buf)
    assert torch.equal(torch.view_as_real(buf), torch.view_as_real(plan.c2c(zd)))


@pytest.mark.parametrize("dtype,n", [("f64", 9000), ("f64", 12000), ("f32", 17000), ("f32", 20000)])
def test_c2c_any_length_over_a_split_convolution(dtype, n):
    """Lengths whose chirp-z convolution (M >= 2n-1 points) is itself beyond a single kernel."""
    rng = np.random.default_rng(n)
    ct = np.complex128 if dtype == "f64" else np.complex64
    z = (rng.uniform(-1, 1, (2, n)) + 1j * rng.uniform(-1, 1, (2, n))).astype(ct)
    plan = dsp_amd.FftPlan(n, dtype)
    assert "split" in plan.variant
    zd = torch.from_numpy(z).cuda()
    for sign in (dsp_amd.FORWARD, dsp_amd.BACKWARD):
        got = plan.c2c(zd, sign).cpu().numpy()
        want = oracle.dft(z[1].astype(np.complex128), sign, "direct")
        assert rel_l2(got[1], want) <= (TOL_F64 if dtype == "f64" else TOL_FFT)


@pytest.mark.parametrize("n", [20000, 32768, 48000, 100000])
def test_double_wrappers_on_long_windows(n):
    """util/fft and util/filter transform whatever window the input table holds -- a second at
    48 kHz is 48000 samples (util/fft/fft.c:321-352, util/filter/filter.c:319-365)."""
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, n)
    bins = dsp_amd.fft_r2c(x)
    if n <= 20000:
        want = oracle.fft_r2c(x)        # the checker's own transform (O(n^2) in long double off powers of two)
    elif n & (n - 1) == 0:
        want = oracle.dft(x.astype(np.complex128), -1, "radix2")[:n // 2 + 1]
    else:
        want = np.fft.rfft(x)           # numpy's pocketfft as the cross-check at sizes the O(n^2) checker takes minutes for
    assert rel_l2(bins, want) <= TOL_F64
    gains = np.where(np.arange(n // 2 + 1) < n // 8, 1.0, 0.25)
    y = dsp_amd.filter(x, gains)
    want_y = oracle.filter(x, gains) if n <= 20000 else np.fft.irfft(np.fft.rfft(x) * gains, n)
    assert rel_l2(y, want_y) <= TOL_F64
    # properties that need no checker: the filter is linear and gain 1 everywhere returns the input
    assert rel_l2(dsp_amd.filter(x, np.ones(n // 2 + 1)), x) <= TOL_F64


def test_fftpy_spectrum_vectors_from_numpy_rfft(golden_dir):
    """fftpy/main.py:30-45 on the device: real transform of an odd or even window, magnitude, division
    by the sum -- against vectors that numpy.fft.rfft itself produced (the reference's FFT for that
    path; tests/golden/make_golden_fftpy.py).  f32 through the one-call spectrum op, double through
    the drop-in r2c wrapper."""
    g = np.load(os.path.join(golden_dir, "fftpy_spectrum.npz"))
    for i in range(7):
        x, want = g["x%d" % i], g["y%d" % i]
        n = len(x)
        plan = dsp_amd.FftPlan(n, "f32")
        xd = torch.from_numpy(x.astype(np.float32)).cuda()[None, :].contiguous()
        got32 = plan.r2c_spectrum(xd, nbin=n // 2 + 1, mode=dsp_amd.PS_NORM).cpu().numpy()[0]
        assert rel_l2(got32, want) <= TOL_FFT, (i, n, rel_l2(got32, want))
        mag = np.abs(dsp_amd.fft_r2c(x))
        assert rel_l2(mag / mag.sum(), want) <= TOL_F64, (i, n)


@pytest.mark.parametrize("n", [16384, 32768])
def test_c2c_longest_single_kernel_frames_in_place(n):
    """The longest frames one workgroup holds in its registers: more frames than CUs, an odd count,
    in place, both signs."""
    batch = 300 + 17
    g = torch.Generator(device="cuda").manual_seed(n)
    z = torch.view_as_complex(torch.rand((batch, n, 2), device="cuda", generator=g) * 2 - 1)
    plan = dsp_amd.FftPlan(n, "f32")
    assert "reg32x32" in plan.variant
    out = plan.c2c(z)
    for f in (0, 1, 255, 256, 257, batch - 1):
        want = oracle.dft(z[f].cpu().numpy().astype(np.complex128), -1, "radix2")
        assert rel_l2(out[f].cpu().numpy(), want) <= TOL_FFT, (n, f)
    buf = z.clone()
    plan.c2c(buf, dsp_amd.FORWARD, out=buf)
    assert torch.equal(torch.view_as_real(buf), torch.view_as_real(out))
    plan.c2c(buf, dsp_amd.BACKWARD, out=buf)
    err = (buf / n - z).abs().max().item()
    assert err <= 2e-5
