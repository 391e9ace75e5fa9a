"""Parity of the HIP FIR / convolution path against the oracle, through the C ABI.

f32 device API: relative L2 error <= 1e-6 per output block (each channel row),
against the oracle's double accumulation of the same f32 inputs.  The kernel
sums with f32 FMAs in ascending tap order; the reference sums in double.
f64 host wrappers: bit-exact against the oracle / the reference's own vectors.
"""
import json
import os

import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu

TOL_FIR = 1e-6  # relative L2 per output block, f32 vs the double oracle


def rel_l2(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    den = np.linalg.norm(want)
    return np.linalg.norm(got - want) / den if den > 0 else np.linalg.norm(got)


def run_f32(x, h, nxx, hist=None, flags=0):
    plan = dsp_amd.FirPlan(h, "f32")
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    hd = None if hist is None else torch.from_numpy(np.ascontiguousarray(hist, dtype=np.float32)).cuda()
    y = plan.exec(xd, nxx, hist=hd, flags=flags)
    torch.cuda.synchronize()
    return y.cpu().numpy(), plan.variant


# ---- host wrappers (double): exact ---------------------------------------------

def test_kat9_wrappers(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "conv_kat9.json")))
    x, h = np.array(kat["x"]), np.array(kat["h"])
    assert np.array_equal(dsp_amd.conv0(x, h), np.array(kat["y0"]))
    assert np.array_equal(dsp_amd.conv1(x, h), np.array(kat["y1"]))
    assert np.allclose(dsp_amd.conv0(x, h), kat["y0_printed"], atol=5e-7, rtol=0)
    assert np.allclose(dsp_amd.conv1(x, h), kat["y1_printed"], atol=5e-7, rtol=0)


def test_wrappers_bit_exact_against_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "conv_ref_cases.npz"))
    for i in range(9):
        x, h = g["x%d" % i], g["h%d" % i]
        assert np.array_equal(dsp_amd.conv0(x, h), g["y0_%d" % i]), i
        assert np.array_equal(dsp_amd.conv1(x, h), g["y1_%d" % i]), i
        full = oracle.convolve(x, h)
        assert np.array_equal(dsp_amd.convolve_alloc(x, h), full), i
        assert np.array_equal(dsp_amd.convolve(x, h, len(x)), full[:len(x)]), i
        assert np.array_equal(dsp_amd.convolve(x, h, len(x) + 2 * len(h)), oracle.convolve(x, h, len(x) + 2 * len(h))), i


def test_fir_wrapper_tone_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "tone_fir.npz"))
    got = dsp_amd.fir(g["tone"])  # default table: alsa/main.c:389-407
    assert np.array_equal(got, g["y16"])
    dsp_amd.fir_set_taps(g["taps21"])
    try:
        assert np.array_equal(dsp_amd.fir(g["tone"]), g["y21"])
    finally:
        dsp_amd.fir_set_taps(g["taps16"])


def test_wrapper_long_kernel_multi_segment():
    """More taps than one LDS staging round holds (fir/src/main.c:324 convolves with a
    964-point kernel; go beyond the 2048-tap segment too)."""
    rng = np.random.default_rng(5)
    for nx, nh in [(2000, 964), (3000, 2500), (700, 5000)]:
        x, h = rng.standard_normal(nx), rng.standard_normal(nh)
        assert np.array_equal(dsp_amd.convolve(x, h, nx), oracle.convolve(x, h, nx))
        assert np.array_equal(dsp_amd.conv0(x, h), oracle.conv0(x, h))


def test_wrappers_keep_plans_between_calls_and_release_them():
    """The wrappers keep the plans and staging buffers of the last few tap sets and transform
    lengths (the reference calls them once per period with the same taps: alsa/main.c:494-516).
    More tap sets and lengths than are kept, in an order that evicts and revisits; growing and
    shrinking signals through the same staging buffers; every result bit for bit (FIR) or to the
    double tolerance (transforms) as on a first call; dspmi_dropin_release gives the memory back."""
    rng = np.random.default_rng(77)
    taps = [rng.standard_normal(nh) for nh in (16, 21, 5, 127, 16, 964)]   # two different 16-tap sets
    free0 = None
    for rnd in range(3):
        if rnd == 1:   # the first pass also loaded the kernels and set up the library's own state: measure from here
            torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info()
        for i in (0, 1, 2, 3, 4, 5, 0, 4, 1):
            nx = int(rng.integers(1, 5000))
            x = rng.standard_normal(nx)
            assert np.array_equal(dsp_amd.conv0(x, taps[i]), oracle.conv0(x, taps[i])), (rnd, i, nx)
            assert np.array_equal(dsp_amd.convolve(x, taps[i], nx + len(taps[i])),
                                  oracle.convolve(x, taps[i], nx + len(taps[i]))), (rnd, i, nx)
        for n in (64, 480, 1024, 482, 4096, 100, 8, 2048, 64, 480, 1024):
            z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            want = oracle.dft(z, -1, "direct")
            assert np.linalg.norm(dsp_amd.dft(z) - want) <= 1e-13 * np.linalg.norm(want), (rnd, n)
            xr = rng.standard_normal(n)
            wr = oracle.fft_r2c(xr)
            assert np.linalg.norm(dsp_amd.fft_r2c(xr) - wr) <= 1e-13 * np.linalg.norm(wr), (rnd, n)
        dsp_amd.lib().dspmi_dropin_release()
        if free0 is not None:
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info()
            assert free1 >= free0 - (4 << 20), (rnd, free0, free1)   # nothing of the wrappers' is left on the device


# ---- f32 device API -----------------------------------------------------------------

@pytest.fixture(params=["auto", "mfma", "ols"])
def fir_path(request, monkeypatch):
    """A plan picks its kernel by tap count (the f32 matrix kernel for the shortest filters, the
    f16 split kernel in the middle, overlap-save for long filters on long rows); "mfma" pins
    the f32 matrix kernel and "ols" overlap-save from 64 taps up, so all three are checked on
    the same cases."""
    if request.param != "auto":
        monkeypatch.setenv("DSPMI_FIR_VARIANT", request.param)
    return request.param


@pytest.mark.parametrize("nh", [1, 9, 16, 21, 32, 33, 127, 128, 300])
@pytest.mark.parametrize("nx,nchan", [(1, 1), (100, 3), (4096, 2), (5000, 4), (70000, 2)])
def test_f32_parity(nh, nx, nchan, fir_path):
    rng = np.random.default_rng(nh * 1000 + nx)
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    h = (oracle.windowed_sinc_taps(nh) if nh >= 9 else rng.uniform(-1, 1, nh)).astype(np.float32)
    for nxx in (nx, nx + nh):
        got, variant = run_f32(x, h, nxx)
        want = oracle.fir_batch_f32(x, h, nxx)
        # the absolute term is the floor of the 16-bit kernels' parts (see assert_blocks): the one
        # output of the 1-sample row under the 33-tap window is x[0] times an edge tap of 3e-19
        floor = 2.0 ** -37 * np.sqrt(nh) * np.abs(h).max() * np.abs(x).max() if "mfma16x16x32" in variant else 0.0
        for c in range(nchan):
            err = np.linalg.norm(got[c].astype(np.float64) - want[c])
            assert err <= TOL_FIR * np.linalg.norm(want[c]) + floor * np.sqrt(nxx), (variant, nh, nx, nxx, c, err)
        if nxx == nx + nh:
            assert np.all(got[:, -1] == 0)  # the extra element of fft/src/main.c:190


def block_errors(got, want, blk=4096):
    """Relative L2 error of every 4096-sample output block of one row (a last block of fewer
    than 1024 samples is counted with the one before it)."""
    edges = list(range(0, len(want), blk)) + [len(want)]
    if len(edges) > 2 and edges[-1] - edges[-2] < blk // 4:
        del edges[-2]
    return [rel_l2(got[a:b], want[a:b]) for a, b in zip(edges, edges[1:])]


@pytest.mark.parametrize("nh", [2049, 4096, 4097, 8191])
@pytest.mark.parametrize("nx,nchan", [(5000, 2), (12345, 1), (20481, 2)])
def test_f32_direct_kernel_beyond_one_tap_segment(nh, nx, nchan, monkeypatch):
    """More than 2048 taps: the direct MFMA kernel walks the taps in LDS-sized segments, carrying
    its accumulators from one to the next and re-staging the tap table each time -- the
    8191-tap direct sum of fft/src/main.c:193-204.  Pinned, because the plan alone would
    pick overlap-save; checked per 4096-sample block, with and without the nx + nh tail."""
    monkeypatch.setenv("DSPMI_FIR_VARIANT", "mfma")
    rng = np.random.default_rng(nh + nx)
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    for nxx in (nx, nx + nh):
        got, variant = run_f32(x, h, nxx)
        assert "mfma" in variant
        want = oracle.fir_batch_f32(x, h, nxx)
        for c in range(nchan):
            worst = max(block_errors(got[c, :nx + nh - 1], want[c, :nx + nh - 1]))
            assert worst <= TOL_FIR, (variant, nh, nx, nxx, c, worst)
        if nxx == nx + nh:
            assert np.all(got[:, -1] == 0)


@pytest.mark.parametrize("nh", [2049, 8191])
def test_f32_direct_kernel_long_filter_with_history(nh, monkeypatch):
    """The same segments with carried history: a long row cut in three chunks gives the row's
    outputs (the window of a tile then reaches into hist for several segments)."""
    monkeypatch.setenv("DSPMI_FIR_VARIANT", "mfma")
    rng = np.random.default_rng(nh)
    nchan, chunk = 2, 9000
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, 3 * chunk)).astype(np.float32)
    whole = oracle.fir_batch_f32(x, h, 3 * chunk)
    plan = dsp_amd.FirPlan(h, "f32")
    assert "mfma" in plan.variant
    xd = torch.from_numpy(x).cuda()
    hist = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")
    outs = []
    for k in range(3):
        outs.append(plan.exec(xd[:, k * chunk:(k + 1) * chunk], chunk, hist=hist).cpu().numpy())
        lo = (k + 1) * chunk - (nh - 1)
        hist = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")
        hist[:, max(0, -lo):] = xd[:, max(0, lo):(k + 1) * chunk]
    got = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert max(block_errors(got[c], whole[c])) <= TOL_FIR, (nh, c)


def test_f32_reference_taps_and_tone(golden_dir):
    g = np.load(os.path.join(golden_dir, "tone_fir.npz"))
    tone = g["tone"].astype(np.float32)[None, :]
    for taps, key in ((g["taps16"], "y16"), (g["taps21"], "y21")):
        got, _ = run_f32(tone, taps.astype(np.float32), len(g[key]))
        assert rel_l2(got[0], oracle.fir_batch_f32(tone, taps.astype(np.float32), len(g[key]))[0]) <= TOL_FIR
        # and against the double-precision vector itself: input rounding (2^-24) dominates
        assert rel_l2(got[0], g[key]) <= 5e-7


def test_f32_drop_first_is_conv1():
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (3, 777)).astype(np.float32)
    h = rng.uniform(-1, 1, 19).astype(np.float32)
    got, _ = run_f32(x, h, 777, flags=dsp_amd.FIR_DROP_FIRST)
    want = oracle.fir_batch_f32(x, h, 777, drop_first=True)
    for c in range(3):
        assert rel_l2(got[c], want[c]) <= TOL_FIR
        ref = oracle.conv1(x[c].astype(np.float64), h.astype(np.float64))
        assert rel_l2(got[c], ref) <= TOL_FIR


def test_f32_streaming_history_matches_one_block(fir_path):
    rng = np.random.default_rng(21)
    nchan, nh = 5, 127
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (nchan, 3 * 8192)).astype(np.float32)
    whole = oracle.fir_batch_f32(x, h, x.shape[1])
    plan = dsp_amd.FirPlan(h, "f32")
    xd = torch.from_numpy(x).cuda()
    hist = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")
    outs = []
    for k in range(3):
        chunk = xd[:, k * 8192:(k + 1) * 8192]
        outs.append(plan.exec(chunk, 8192, hist=hist).cpu().numpy())
        hist = chunk[:, -(nh - 1):].contiguous()
    got = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert rel_l2(got[c], whole[c]) <= TOL_FIR


def test_f32_strided_rows_and_unaligned_views(fir_path):
    rng = np.random.default_rng(2)
    h = rng.uniform(-1, 1, 16).astype(np.float32)
    big = torch.from_numpy(rng.uniform(-1, 1, (4, 1000)).astype(np.float32)).cuda()
    x = big[:, 3:903]  # row stride 1000, start 12 bytes off a 16-byte boundary
    ybig = torch.zeros((4, 1003), dtype=torch.float32, device="cuda")
    y = ybig[:, 1:901]
    plan = dsp_amd.FirPlan(h, "f32")
    plan.exec(x, 900, y=y)
    want = oracle.fir_batch_f32(x.cpu().numpy(), h, 900)
    got = y.cpu().numpy()
    for c in range(4):
        assert rel_l2(got[c], want[c]) <= TOL_FIR
    assert torch.all(ybig[:, 0] == 0) and torch.all(ybig[:, 901:] == 0)  # nothing written outside the rows


def test_linearity_and_impulse_at_scale(fir_path):
    """Size-independent properties on a block too large for the oracle to sweep."""
    nh, nchan, nx = 127, 64, 1 << 20
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    assert ("overlap_save" in plan.variant) == (fir_path == "ols")
    assert ("f16x2" in plan.variant) == (fir_path == "auto")
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.rand((nchan, nx), device="cuda", generator=g) * 2 - 1
    b = torch.rand((nchan, nx), device="cuda", generator=g) * 2 - 1
    ya, yb, yab = plan.exec(a), plan.exec(b), plan.exec(a + b)
    err = (yab - (ya + yb)).double().norm() / yab.double().norm()
    assert err.item() < 1e-6
    imp = torch.zeros((2, nx), device="cuda")
    imp[0, 0] = 1.0
    imp[1, nx - 200] = 1.0
    yi = plan.exec(imp).cpu().numpy()
    if fir_path == "mfma":
        # one product per output, added to zeros: the taps come back exactly
        assert np.array_equal(yi[0, :nh], h) and not yi[0, nh:].any()
        assert np.array_equal(yi[1, nx - 200:nx - 200 + nh], h) and not yi[1, :nx - 200].any()
    elif fir_path == "auto":
        # the taps as the sum of their two f16 parts (22 bits), and exact zeros elsewhere
        assert rel_l2(yi[0, :nh], h) <= 2.0 ** -22 and not yi[0, nh:].any()
        assert rel_l2(yi[1, nx - 200:nx - 200 + nh], h) <= 2.0 ** -22 and not yi[1, :nx - 200].any()
    else:
        # through the transforms: the taps to rounding, and rounding noise elsewhere
        hn = np.linalg.norm(h)
        assert rel_l2(yi[0, :nh], h) <= TOL_FIR and np.abs(yi[0, nh:]).max() <= 1e-6 * hn
        assert rel_l2(yi[1, nx - 200:nx - 200 + nh], h) <= TOL_FIR and np.abs(yi[1, :nx - 200]).max() <= 1e-6 * hn
    # spot rows against the oracle
    want = oracle.fir_batch_f32(a[:2, :20000].cpu().numpy(), h, 20000)
    got = ya[:2, :20000].cpu().numpy()
    for c in range(2):
        assert rel_l2(got[c], want[c]) <= TOL_FIR


@pytest.mark.parametrize("nh", [300, 127])
def test_many_tiles_per_workgroup_walk(nh, fir_path):
    """Enough tiles that every workgroup walks several of them (the prefetch pipeline's steady
    state), with a tap count between the two staging-register variants; spot rows vs the oracle."""
    nchan, nx = 64, 1 << 21
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    g = torch.Generator(device="cuda").manual_seed(nh)
    x = torch.rand((nchan, nx), device="cuda", generator=g) * 2 - 1
    y = plan.exec(x)
    for c in (0, 31, 63):
        for lo in (0, nx // 2 - 5000, nx - 10000):
            seg = x[c, max(0, lo - nh):lo + 10000].cpu().numpy()
            want = oracle.fir_batch_f32(seg, h, len(seg))[0][lo - max(0, lo - nh):]
            got = y[c, lo:lo + 10000].cpu().numpy()
            assert rel_l2(got, want) <= TOL_FIR, (c, lo)


def test_degenerate_shapes_through_the_abi():
    """Empty and ragged calls: no output, no channel, more taps than samples, zero-length input."""
    import ctypes as C
    L = dsp_amd.lib()
    h = np.arange(1, 6, dtype=np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    x = torch.ones((2, 3), device="cuda")
    # nxx = 0 and nchan = 0 are no-ops that succeed
    assert L.dspmi_fir_exec_f32(plan._h, None, 0, 0, x.data_ptr(), 3, 3, 2, None, 0, 0, None) == 0
    assert L.dspmi_fir_exec_f32(plan._h, None, 0, 5, None, 0, 0, 0, None, 0, 0, None) == 0
    # more taps than samples, output longer than nx + nh: the tail is zeros
    y = plan.exec(x, 12).cpu().numpy()
    want = oracle.fir_batch_f32(np.ones((2, 3), np.float32), h, 12)
    np.testing.assert_allclose(y, want, rtol=0, atol=1e-6)
    assert not y[:, 7:].any()
    # nx = 0: every output is an empty sum
    y0 = torch.full((1, 4), 7.0, device="cuda")
    assert L.dspmi_fir_exec_f32(plan._h, y0.data_ptr(), 4, 4, None, 0, 0, 1, None, 0, 0, None) == 0
    torch.cuda.synchronize()
    assert not y0.cpu().numpy().any()
    # argument errors come back as -EINVAL with a message, nothing is launched
    assert L.dspmi_fir_exec_f32(plan._h, y0.data_ptr(), 2, 4, x.data_ptr(), 3, 3, 1, None, 0, 0, None) == -22
    assert b"stride" in L.dspmi_last_error()
    # host wrappers with empty tap sets / single samples
    assert np.array_equal(dsp_amd.conv0(np.array([2.0]), np.array([3.0])), np.array([6.0]))
    assert np.array_equal(dsp_amd.conv1(np.array([2.0]), np.array([3.0])), np.array([0.0]))
    assert np.array_equal(dsp_amd.convolve(np.array([1.0, 2.0]), np.array([1.0]), 5), np.array([1.0, 2.0, 0, 0, 0]))


def test_in_place_is_refused_interleaved_rows_are_not(fir_path):
    """A tile reads the nh-1 samples before it, which another workgroup may already have
    replaced: output rows that touch input rows are an argument error, for both kernels.  Rows
    of x and y interleaved in one buffer (common stride, disjoint columns) are fine."""
    rng = np.random.default_rng(11)
    h = oracle.windowed_sinc_taps(127).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    nchan, nx = 3, 6000
    buf = torch.zeros((nchan, 2 * nx + 64), device="cuda")
    x = buf[:, :nx]
    x.copy_(torch.from_numpy(rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)))
    with pytest.raises(dsp_amd.DspmiError, match="overlap"):
        plan.exec(x, y=x)
    with pytest.raises(dsp_amd.DspmiError, match="overlap"):
        plan.exec(x, y=buf[:, nx - 1:2 * nx - 1])          # one shared column
    y = plan.exec(x, nx + 64, y=buf[:, nx:])               # interleaved, disjoint, filling the row exactly
    want = oracle.fir_batch_f32(x.cpu().numpy(), h, nx + 64)
    for c in range(nchan):
        assert rel_l2(y[c].cpu().numpy(), want[c]) <= TOL_FIR
    y = plan.exec(x, y=buf[:, nx:2 * nx])
    for c in range(nchan):
        assert rel_l2(y[c].cpu().numpy(), want[c, :nx]) <= TOL_FIR


# ---- the bf16 split kernel (17..305 taps) ---------------------------------------------

@pytest.fixture(params=["bf16", "f16"])
def split_kind(request):
    """The two kernels that split f32 operands for the 16-bit matrix pipe: three bf16 parts, or
    two f16 parts under a per-tile scale."""
    return request.param


def assert_blocks(got, want, h, x, what, blk=4096):
    """Every 4096-sample block within TOL_FIR of its own norm, plus the absolute floor of the
    16-bit parts: a tap or sample more than 2^39 below the largest of its tile is not carried
    (outputs that are sums of such terms alone -- the first output under a window whose edge tap
    is 1e-18 -- come out as zero)."""
    floor = 2.0 ** -37 * np.sqrt(len(h)) * np.abs(h).max() * max(np.abs(x).max(), 1e-300)
    edges = list(range(0, len(want), blk)) + [len(want)]
    if len(edges) > 2 and edges[-1] - edges[-2] < blk // 4:
        del edges[-2]
    for a, b in zip(edges, edges[1:]):
        err = np.linalg.norm(got[a:b].astype(np.float64) - want[a:b])
        assert err <= TOL_FIR * np.linalg.norm(want[a:b]) + floor * np.sqrt(b - a), (what, a, err, np.linalg.norm(want[a:b]))


def split_plan(h, monkeypatch, kind):
    monkeypatch.setenv("DSPMI_FIR_VARIANT", kind)
    plan = dsp_amd.FirPlan(h, "f32")
    assert kind in plan.variant, plan.variant
    return plan


@pytest.mark.parametrize("nh", [17, 18, 48, 49, 100, 127, 177, 178, 256, 305])
def test_split_kernel_every_slice_count(nh, monkeypatch, split_kind):
    """One tap count at and one past the edges of the 32-position slices (nh + 15 = 32 k), ragged
    rows of every length class: shorter than a tile, a tile plus a few, rows that end inside a
    16-byte group; outputs cut at nx and run on to nx + nh."""
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = split_plan(h, monkeypatch, split_kind)
    rng = np.random.default_rng(nh)
    for nx, nchan in [(3, 2), (nh - 1, 1), (4095, 3), (4097, 2), (9001, 3), (3 * 4096 + 2, 2)]:
        x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        for nxx in (nx, nx + nh, nx + nh - 1, max(1, nx - 5)):
            got = plan.exec(xd, nxx).cpu().numpy()
            want = oracle.fir_batch_f32(x, h, nxx)
            for c in range(nchan):
                assert_blocks(got[c], want[c], h, x[c], (nh, nx, nxx, c))


@pytest.mark.parametrize("nh,chunk", [(127, 8192), (100, 5000), (33, 4096), (305, 4097), (64, 300)])
def test_split_kernel_streaming_history(nh, chunk, monkeypatch, split_kind):
    """A row cut in chunks with the previous chunk's tail as history gives the row's outputs; tap
    counts whose history does not line up with the 16-byte groups of the window, chunks shorter
    than the history included."""
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = split_plan(h, monkeypatch, split_kind)
    rng = np.random.default_rng(nh + chunk)
    nchan, nchunk = 3, 4
    x = rng.uniform(-1, 1, (nchan, nchunk * chunk)).astype(np.float32)
    whole = oracle.fir_batch_f32(x, h, x.shape[1])
    xd = torch.from_numpy(x).cuda()
    hist = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")
    outs = []
    for k in range(nchunk):
        piece = xd[:, k * chunk:(k + 1) * chunk]
        outs.append(plan.exec(piece, chunk, hist=hist).cpu().numpy())
        hist = torch.cat([hist, piece], dim=1)[:, -(nh - 1):].contiguous()
    got = np.concatenate(outs, axis=1)
    for c in range(nchan):
        assert max(block_errors(got[c], whole[c])) <= TOL_FIR, (nh, chunk, c)


def test_split_kernel_unaligned_rows(monkeypatch, split_kind):
    """Rows that start 4 and 12 bytes off a 16-byte boundary, row strides that are no multiple of
    four samples, and nothing written outside the output rows."""
    rng = np.random.default_rng(40)
    h = oracle.windowed_sinc_taps(40).astype(np.float32)
    plan = split_plan(h, monkeypatch, split_kind)
    big = torch.from_numpy(rng.uniform(-1, 1, (5, 9003)).astype(np.float32)).cuda()
    x = big[:, 3:9000]
    ybig = torch.zeros((5, 9101), dtype=torch.float32, device="cuda")
    y = ybig[:, 1:9038]
    plan.exec(x, 9037, y=y)
    want = oracle.fir_batch_f32(x.cpu().numpy(), h, 9037)
    got = y.cpu().numpy()
    for c in range(5):
        assert max(block_errors(got[c], want[c])) <= TOL_FIR
    assert torch.all(ybig[:, 0] == 0) and torch.all(ybig[:, 9038:] == 0)


def test_split_kernel_walks_tiles_across_rows(monkeypatch, split_kind):
    """Enough tiles that a workgroup walks several, with rows of 2.5 tiles: walks cross from one
    row into the next, and every row has a ragged last tile."""
    nh, nchan, nx = 127, 1500, 2 * 4096 + 2000
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = split_plan(h, monkeypatch, split_kind)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand((nchan, nx), device="cuda", generator=g) * 2 - 1
    y = plan.exec(x, nx + nh)
    for c in (0, 1, 2, 700, 1498, 1499):
        want = oracle.fir_batch_f32(x[c:c + 1].cpu().numpy(), h, nx + nh)[0]
        assert max(block_errors(y[c].cpu().numpy(), want)) <= TOL_FIR, c


def test_split_kernel_wide_dynamic_range(monkeypatch, split_kind):
    """Samples spread over twelve decades: the three bf16 parts carry a float exactly at any
    exponent, so the per-block error stays that of an f32 sum."""
    nh = 127
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = split_plan(h, monkeypatch, split_kind)
    rng = np.random.default_rng(12)
    nx = 6 * 4096
    x = (rng.uniform(-1, 1, (2, nx)) * 10.0 ** rng.uniform(-6, 6, (2, nx))).astype(np.float32)
    got = plan.exec(torch.from_numpy(x).cuda(), nx).cpu().numpy()
    want = oracle.fir_batch_f32(x, h, nx)
    for c in range(2):
        assert max(block_errors(got[c], want[c])) <= TOL_FIR
    # and tiny values keep their relative accuracy (no flush of the small parts)
    xs = (x * 1e-30).astype(np.float32)
    got = plan.exec(torch.from_numpy(xs).cuda(), nx).cpu().numpy()
    want = oracle.fir_batch_f32(xs, h, nx)
    for c in range(2):
        assert max(block_errors(got[c], want[c])) <= TOL_FIR


def test_split_kernels_degenerate_shapes(monkeypatch, split_kind):
    """No input at all, fewer samples than taps, one output, many one-tile rows: through the ABI."""
    import ctypes as C
    L = dsp_amd.lib()
    nh = 64
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    plan = split_plan(h, monkeypatch, split_kind)
    y0 = torch.full((2, 9), 7.0, device="cuda")
    assert L.dspmi_fir_exec_f32(plan._h, y0.data_ptr(), 9, 9, None, 0, 0, 2, None, 0, 0, None) == 0     # nx = 0: empty sums
    torch.cuda.synchronize()
    assert not y0.cpu().numpy().any()
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (4000, 5)).astype(np.float32)                                                 # 4000 rows of 5 samples
    got = plan.exec(torch.from_numpy(x).cuda(), 5 + nh).cpu().numpy()
    want = oracle.fir_batch_f32(x, h, 5 + nh)
    for c in (0, 1, 1999, 3999):
        assert_blocks(got[c], want[c], h, x[c], c)
    hist = rng.uniform(-1, 1, (4000, nh - 1)).astype(np.float32)
    got = plan.exec(torch.from_numpy(x).cuda(), 5, hist=torch.from_numpy(hist).cuda()).cpu().numpy()      # everything from the history
    want = oracle.fir_batch_f32(np.concatenate([hist, x], axis=1), h, nh - 1 + 5)[:, nh - 1:]
    for c in (0, 1, 1999, 3999):
        assert_blocks(got[c], want[c], h, np.concatenate([hist[c], x[c]]), c)


@pytest.mark.parametrize("nh", [1, 4, 5, 9, 12, 13, 16, 17, 21, 32])
@pytest.mark.parametrize("nx,nchan,stride_pad", [(64, 1, 0), (65, 3, 3), (1023, 5, 1), (2047, 2, 1), (4099, 3, 1), (100000, 2, 0), (100002, 2, 2)])
@pytest.mark.parametrize("with_hist", [False, True])
def test_short_filters_as_a_stream(nh, nx, nchan, stride_pad, with_hist, monkeypatch):
    """Up to 16 taps (up to 32 on rows shorter than the matrix kernel's tiles) an f32 plan runs calls
    with as many outputs as inputs on 16-byte aligned rows through a kernel without tiles (four
    outputs per thread straight from global memory, the ends of every row by a second small kernel);
    other calls, and DSPMI_FIR_STREAM=0, run the matrix kernel.  Both against the oracle per row, and
    against each other: row lengths that are not multiples of four, padded strides, history."""
    rng = np.random.default_rng(nh * 1000 + nx)
    h = (oracle.windowed_sinc_taps(nh) if nh > 1 else np.array([0.75])).astype(np.float32)
    stride = nx + stride_pad
    if stride % 4:
        stride += 4 - stride % 4          # rows start on 16-byte boundaries
    buf = torch.zeros((nchan, stride), dtype=torch.float32, device="cuda")
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    buf[:, :nx] = torch.from_numpy(x).cuda()
    xd = buf[:, :nx]
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32) if with_hist and nh > 1 else None
    hd = None if hist is None else torch.from_numpy(hist).cuda()
    plan = dsp_amd.FirPlan(h, "f32")
    assert plan.variant.startswith("fir_f32_stream") == (nh <= 16)
    ybuf = torch.zeros((nchan, stride), dtype=torch.float32, device="cuda")
    a = plan.exec(xd, nx, y=ybuf[:, :nx], hist=hd).cpu().numpy().copy()
    monkeypatch.setenv("DSPMI_FIR_STREAM", "0")
    b = plan.exec(xd, nx, y=ybuf[:, :nx], hist=hd).cpu().numpy().copy()
    full = x if hist is None else np.concatenate([hist, x], axis=1)
    want = oracle.fir_batch_f32(full, h, full.shape[1])[:, full.shape[1] - nx:]
    for c in range(nchan):
        assert rel_l2(a[c], want[c]) <= TOL_FIR, (nh, nx, c, rel_l2(a[c], want[c]))
        assert rel_l2(b[c], want[c]) <= TOL_FIR
        assert rel_l2(a[c], b[c]) <= TOL_FIR


def test_short_filter_stream_over_many_rows():
    """More rows than one grid dimension holds (65535): the rest goes into a second launch."""
    nh, nx, nchan = 16, 64, 70001
    rng = np.random.default_rng(3)
    h = np.asarray(oracle.TAPS16, dtype=np.float32)
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    y = plan.exec(torch.from_numpy(x).cuda(), nx).cpu().numpy()
    for c in (0, 1, 65534, 65535, 65536, 70000):
        want = oracle.fir_batch_f32(x[c:c + 1], h, nx)[0]
        assert rel_l2(y[c], want) <= TOL_FIR, c


@pytest.mark.parametrize("nh", [1, 3, 9, 12, 16, 21, 32])
@pytest.mark.parametrize("nx,nchan", [(64, 1), (65, 3), (300, 4), (1001, 4), (5001, 2), (70000, 2)])
def test_short_filters_as_a_stream_in_double(nh, nx, nchan, monkeypatch):
    """The double-precision plans behind the host routines run short filters through the same
    kernel without tiles (two outputs per lane, every product and sum rounded on its own in the
    reference's order): bit for bit the tile kernel's outputs (DSPMI_FIR_STREAM=0) with carried
    history, and bit for bit the oracle's conv0 / conv1 / convolve through the host routines."""
    rng = np.random.default_rng(nh * 100 + nx)
    h = rng.uniform(-1, 1, nh)
    stride = (nx + 1) & ~1                      # rows start on 16-byte boundaries
    buf = torch.zeros((nchan, stride), dtype=torch.float64, device="cuda")
    x = rng.uniform(-1, 1, (nchan, nx))
    buf[:, :nx] = torch.from_numpy(x).cuda()
    hist = torch.from_numpy(rng.uniform(-1, 1, (nchan, nh - 1))).cuda() if nh > 1 else None
    plan = dsp_amd.FirPlan(h, "f64")
    ybuf = torch.zeros_like(buf)
    a = plan.exec(buf[:, :nx], nx, y=ybuf[:, :nx], hist=hist).cpu().numpy().copy()
    monkeypatch.setenv("DSPMI_FIR_STREAM", "0")
    b = plan.exec(buf[:, :nx], nx, y=ybuf[:, :nx], hist=hist).cpu().numpy().copy()
    monkeypatch.delenv("DSPMI_FIR_STREAM")
    assert np.array_equal(a, b)
    if nchan == 2:
        xs = x[0]
        assert np.array_equal(dsp_amd.conv0(xs, h), oracle.conv0(xs, h))
        assert np.array_equal(dsp_amd.conv1(xs, h), oracle.conv1(xs, h))
        assert np.array_equal(dsp_amd.convolve(xs, h, nx), oracle.convolve(xs, h, nx))
        assert np.array_equal(dsp_amd.convolve_alloc(xs, h), oracle.convolve(xs, h))


# ---- what the f16 kernel does with samples it has no scale for ---------------------------

def _nonfinite_reference(x, h, nxx):
    """The reference's sum (fft/src/main.c:197-203) on a row with non-finite samples: which
    outputs are finite, and their values (numpy warns about inf * 0; that NaN is the point)."""
    with np.errstate(invalid="ignore", over="ignore"):
        return oracle.fir_batch_f32(x, h, nxx)


@pytest.mark.parametrize("nh", [33, 127, 305])
def test_f16_kernel_confines_non_finite_samples_as_the_reference_does(nh):
    """One NaN and one infinity, in different tiles of a row, near a tile edge and in the middle:
    exactly the outputs whose nh samples include one of them are non-finite, as in the
    reference's sum, and every other output keeps the usual accuracy -- the tile's scale does not
    exist then, so the kernel sums such tiles tap by tap."""
    rng = np.random.default_rng(nh)
    nchan, nx = 3, 5 * 4096
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x[0, 4096 - 7] = np.nan          # reaches into the next tile
    x[0, 3 * 4096 + 1000] = np.inf
    x[1, 0] = -np.inf
    x[1, nx - 1] = np.nan
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)
    hist[2, 5] = np.nan              # the carried history counts as samples too
    plan = dsp_amd.FirPlan(h, "f32")
    assert "f16x2" in plan.variant
    xd, hd = torch.from_numpy(x).cuda(), torch.from_numpy(hist).cuda()
    for nxx, hh in ((nx, hd), (nx + nh, None)):
        got = plan.exec(xd, nxx, hist=hh).cpu().numpy()
        xin = x if hh is None else np.concatenate([hist, x], axis=1)
        want = _nonfinite_reference(xin, h, xin.shape[1] + (nxx - nx))[:, xin.shape[1] - nx:]
        assert want.shape == got.shape
        bad = ~np.isfinite(want)
        assert bad.any() and np.array_equal(~np.isfinite(got), bad), (nh, nxx, (np.isfinite(got) == bad).sum())
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[np.isinf(want)], want[np.isinf(want)].astype(np.float32))
        for c in range(nchan):
            ok = ~bad[c]
            for a in range(0, nxx, 4096):
                sel = ok[a:a + 4096]
                with np.errstate(invalid="ignore"):
                    err = np.linalg.norm((got[c, a:a + 4096] - want[c, a:a + 4096])[sel])
                assert err <= TOL_FIR * np.linalg.norm(want[c, a:a + 4096][sel]), (nh, nxx, c, a, err)


@pytest.mark.parametrize("nh", [33, 127])
def test_f16_kernel_dynamic_range_inside_a_tile(nh, monkeypatch):
    """One scale per tile: a sample 2^40 and more above its neighbours (a 1e6 spike in a 1e-6
    signal, a 1e30 outlier in a signal of order 1) takes the neighbours' precision -- every output
    is within 2^-37 sqrt(nh) max|h| of the TILE's largest sample, not of its own samples; tiles
    without the outlier are unaffected.  A plan created with DSPMI_FIR_FULL_RANGE keeps the
    usual bound relative to each output's own samples."""
    rng = np.random.default_rng(nh + 1)
    nx = 3 * 4096
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    for spike, level in ((1e6, 1e-6), (1e30, 1.0)):
        x = (level * rng.uniform(-1, 1, (1, nx))).astype(np.float32)
        k = 4096 + 2000
        x[0, k] = spike
        want = oracle.fir_batch_f32(x, h, nx)[0]
        near = np.zeros(nx, bool)
        near[k:k + nh] = True                       # outputs that see the spike
        xd = torch.from_numpy(x).cuda()
        half = dsp_amd.FirPlan(h, "f32")
        assert "f16x2" in half.variant
        got = half.exec(xd).cpu().numpy()[0]
        # the outputs under the spike: relative to themselves (they are of the spike's order)
        assert rel_l2(got[near], want[near]) <= TOL_FIR
        # the rest of the spike's tile: the absolute bound of the contract
        bound = 2.0 ** -37 * np.sqrt(nh) * np.abs(h).max() * spike
        tile = np.zeros(nx, bool)
        tile[4096:8192] = True
        # (the window of the next tile starts up to 304 samples early: its scale sees the spike too
        # if the spike lies that close to the edge; here it does not)
        assert np.abs(got - want)[tile & ~near].max() <= bound
        # tiles without the spike: untouched
        for a in (0, 8192):
            assert rel_l2(got[a:a + 4096], want[a:a + 4096]) <= TOL_FIR
        # full range on request: every output relative to the samples under its own taps
        full = dsp_amd.FirPlan(h, "f32", options=dsp_amd.FirPlan.FULL_RANGE)
        assert "bf16x3" in full.variant
        gf = full.exec(xd).cpu().numpy()[0]
        far = tile & ~near
        far[k - 64:k + nh + 64] = False             # (clear of the 16-output blocks around the spike)
        assert rel_l2(gf[far], want[far]) <= TOL_FIR
        assert rel_l2(gf[near], want[near]) <= TOL_FIR


def test_plan_options_select_the_kernels():
    h = oracle.windowed_sinc_taps(127).astype(np.float32)
    assert "f16x2" in dsp_amd.FirPlan(h, "f32").variant
    assert "bf16x3" in dsp_amd.FirPlan(h, "f32", options=dsp_amd.FirPlan.FULL_RANGE).variant
    assert "mfma16x16x4" in dsp_amd.FirPlan(h, "f32", options=dsp_amd.FirPlan.F32_PRODUCTS).variant
    # outside the f16 kernel's range the full-range option changes nothing
    h16 = oracle.TAPS16.astype(np.float32)
    assert dsp_amd.FirPlan(h16, "f32", options=dsp_amd.FirPlan.FULL_RANGE).variant == dsp_amd.FirPlan(h16, "f32").variant
    with pytest.raises(dsp_amd.DspmiError):
        dsp_amd.FirPlan(h, "f32", options=64)


def test_fir_and_spectrum_kernel_with_a_non_finite_sample():
    """The fused filter + spectrum kernel takes the same tap-by-tap path: the frame that holds the
    filtered NaN has no finite spectrum, every other frame is what the two calls give."""
    rng = np.random.default_rng(3)
    nh, nfft = 127, 4096
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    x = rng.uniform(-1, 1, (2, 4 * nfft)).astype(np.float32)
    x[0, nfft + 50] = np.nan
    x[1, 2 * nfft - 3] = np.inf      # its filtered samples reach into frame 2
    fir, fft = dsp_amd.FirPlan(h, "f32"), dsp_amd.FftPlan(nfft, "f32")
    xd = torch.from_numpy(x).cuda()
    got = fir.spectrum(fft, xd).cpu().numpy().reshape(2, 4, -1)
    y = fir.exec(xd)
    want = fft.r2c_spectrum(y.view(-1, nfft)).cpu().numpy().reshape(2, 4, -1)
    for c, f in ((0, 1), (1, 1), (1, 2)):
        assert not np.isfinite(got[c, f]).any()
    for c, f in ((0, 0), (0, 2), (0, 3), (1, 0), (1, 3)):
        assert np.isfinite(got[c, f]).all() and rel_l2(got[c, f], want[c, f]) <= 1e-5, (c, f)


@pytest.mark.parametrize("nh", [306, 417, 529, 530, 700, 964, 1009])
@pytest.mark.parametrize("nx", [700, 4096, 7000])
def test_mid_length_filters_on_short_rows_run_the_f16_kernel(nh, nx):
    """306 .. 1009 taps (11 .. 32 slices; from 18 slices the tap images sit in LDS): the plan is an
    overlap-save plan that also holds the f16 kernel's tap images and runs it on rows below the
    measured crossover -- with carried history, with the nx + nh tail, and against the same plan
    created with DSPMI_FIR_FULL_RANGE, which keeps overlap-save and the f32 kernels
    (fir/src/main.c:258-280 is the sum both restate; :324 applies a 964-point kernel)."""
    rng = np.random.default_rng(nh + nx)
    nchan = 19
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)
    xd, hd = torch.from_numpy(x).cuda(), torch.from_numpy(hist).cuda()
    want_h = oracle.fir_batch_f32(np.concatenate([hist, x], axis=1), h, nh - 1 + nx)[:, nh - 1:]
    want_0 = oracle.fir_batch_f32(x, h, nx + nh)
    plans = [dsp_amd.FirPlan(h, "f32"), dsp_amd.FirPlan(h, "f32", options=dsp_amd.FirPlan.FULL_RANGE)]
    assert all("overlap_save" in pl.variant for pl in plans)
    outs = []
    for pl in plans:
        got_h = pl.exec(xd, nx, hist=hd).cpu().numpy()
        got_0 = pl.exec(xd, nx + nh).cpu().numpy()
        outs.append(got_h)
        for c in (0, nchan // 2, nchan - 1):
            assert_blocks(got_h[c], want_h[c], h, x[c], (pl.variant, nh, nx, c, "history"))
            assert_blocks(got_0[c], want_0[c], h, x[c], (pl.variant, nh, nx, c, "tail"))
    assert not np.array_equal(outs[0], outs[1])   # two different kernels did the work


@pytest.mark.parametrize("nh", [33, 127, 305])
@pytest.mark.parametrize("nx", [100, 256, 1000, 1024, 1025, 1536, 2048, 2100, 3072, 5120, 6000])
def test_f16_kernel_tiles_by_row_length(nh, nx, monkeypatch):
    """Rows shorter than the walker's 4096-output tiles run through tiles of 1024 or 2048 outputs
    (the plan's choice by row length); every tile size on every row length gives the same outputs
    as the oracle per block -- many rows per launch, so that workgroups walk from row to row, with
    carried history and with the nx + nh tail."""
    rng = np.random.default_rng(nh * 7 + nx)
    nchan = 37
    x = rng.uniform(-1, 1, (nchan, nx)).astype(np.float32)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    hist = rng.uniform(-1, 1, (nchan, nh - 1)).astype(np.float32)
    plan = dsp_amd.FirPlan(h, "f32")
    assert "f16x2" in plan.variant
    xd, hd = torch.from_numpy(x).cuda(), torch.from_numpy(hist).cuda()
    want_h = oracle.fir_batch_f32(np.concatenate([hist, x], axis=1), h, nh - 1 + nx)[:, nh - 1:]
    want_0 = oracle.fir_batch_f32(x, h, nx + nh)
    for tile in (None, "1024", "2048", "4096"):
        if tile is None:
            monkeypatch.delenv("DSPMI_FIR_TILE", raising=False)
        else:
            monkeypatch.setenv("DSPMI_FIR_TILE", tile)
        got_h = plan.exec(xd, nx, hist=hd).cpu().numpy()
        got_0 = plan.exec(xd, nx + nh).cpu().numpy()
        for c in (0, 1, nchan // 2, nchan - 1):
            assert_blocks(got_h[c], want_h[c], h, x[c], (tile, nh, nx, c, "history"))
            assert_blocks(got_0[c], want_0[c], h, x[c], (tile, nh, nx, c, "tail"))
