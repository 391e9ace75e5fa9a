"""The C demos (demos/*.c: the reference's mains over include/dsp_dropin.h) build with
gcc, and on the GPU print what the reference prints."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "demos", "bin")


def _build():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "demos")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_demos_compile_as_plain_c():
    _build()
    for name in ("convolution_demo", "fir_demo", "filter_demo", "fftc_demo", "fftw_demo", "firdesign_demo", "fft_tool", "filter_tool",
                 "multi_gpu_demo"):
        assert os.path.exists(os.path.join(BIN, name))


def _run(name):
    _build()
    r = subprocess.run([os.path.join(BIN, name)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.gpu
def test_convolution_demo_prints_the_reference_table(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "conv_kat9.json")))
    want = "".join("%u %f %f %f\n" % (i, kat["x"][i], kat["y0"][i], kat["y1"][i]) for i in range(9))
    assert _run("convolution_demo") == want
    rows = [ln.split() for ln in want.splitlines()]
    assert [float(r[2]) for r in rows] == kat["y0_printed"] and [float(r[3]) for r in rows] == kat["y1_printed"]


@pytest.mark.gpu
def test_fir_demo_matches_oracle_text(golden_dir):
    g = np.load(os.path.join(golden_dir, "tone_fir.npz"))
    want = "".join("%f %f %f\n" % (i / 48000.0, g["y16"][i], g["tone"][i]) for i in range(1000))
    assert _run("fir_demo") == want


@pytest.mark.gpu
def test_filter_demo_matches_oracle():
    t = oracle.Tonegen()
    t.add(48000.0, 1000.0, 1.0, 0.0)
    t.add(48000.0, 12000.0, 0.5, 0.3)
    x = t.read(4096)
    fres = 48000.0 / 4096
    f = np.arange(2049) * fres
    coeffs = np.where(f < 3000, 1.0, np.where(f < 3500, 0.5, 0.0))
    want = oracle.filter(x, coeffs)
    rows = np.array([[float(v) for v in ln.split()] for ln in _run("filter_demo").splitlines()])
    assert rows.shape == (4096, 3)
    np.testing.assert_allclose(rows[:, 1], x, atol=6e-7)
    np.testing.assert_allclose(rows[:, 2], want, atol=6e-7)


@pytest.mark.gpu
def test_fftc_demo_matches_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "fft_cases.npz"))
    y = g["fftc_y"]
    rows = np.array([[float(v) for v in ln.split()] for ln in _run("fftc_demo").splitlines()])
    assert rows.shape == (240, 5)
    np.testing.assert_allclose(rows[:, 0], 100.0 * np.arange(240), atol=1e-6)
    np.testing.assert_allclose(rows[:, 1], oracle.ps_abs(y), atol=2e-6)
    np.testing.assert_allclose(rows[:, 2], y.real[:240] / 240, atol=2e-6)
    np.testing.assert_allclose(rows[:, 3], -y.imag[:240] / 240, atol=2e-6)


@pytest.mark.gpu
def test_fftw_subset_demo_matches_oracle():
    """The reference's wrappers compiled against include/dspmi_fftw3.h instead of <fftw3.h>."""
    out = _run("fftw_demo").splitlines()
    spec = np.array([[float(v) for v in ln.split()] for ln in out[:480]])
    filt = np.array([[float(v) for v in ln.split()] for ln in out[480:]])
    t = oracle.Tonegen()
    t.add(48000, 2000, 100, 0); t.add(48000, 4000, 100, 0.666); t.add(48000, 6000, 100, 0)
    x = t.read_complex(480)
    y = oracle.dft(x, -1, "direct")
    np.testing.assert_allclose(spec[:, 1], x.real, atol=1e-6)
    np.testing.assert_allclose(spec[:, 2], x.real, atol=1e-6)          # backward(forward(x))/nx
    np.testing.assert_allclose(spec[:, 3], y.real, atol=2e-6 * np.abs(y).max())
    np.testing.assert_allclose(spec[:, 4], y.imag, atol=2e-6 * np.abs(y).max())
    t = oracle.Tonegen()
    t.add(48000, 1000, 1.0, 0); t.add(48000, 12000, 0.5, 0.3)
    s = t.read(4096)
    g = np.where(np.arange(2049) * 48000.0 / 4096 < 3000, 1.0, 0.0)
    assert filt.shape == (4096, 3)
    np.testing.assert_allclose(filt[:, 2], oracle.filter(s, g), atol=1e-6)


CUTS = [1000, 2500, 3000, 3200, 6000, 10000, 14000, 20000]


@pytest.mark.gpu
def test_filter_kernel_designer_matches_oracle():
    """make_filter_kernel (fir/src/main.c:149-226) over the device backward DFT (482 points:
    not a power of two) against the oracle's long-double DFT."""
    import dsp_amd
    for nx, cuts in ((241, CUTS), (8, [6000]), (16, [3000, 9000]), (1025, CUTS)):
        fresp = oracle.make_multipass_fresp(nx, cuts, 48000.0)
        want = oracle.make_filter_kernel(fresp)
        got = dsp_amd.make_filter_kernel(fresp)
        assert got.shape == want.shape == (2 * nx,)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        assert not got[32:].any()
    assert np.array_equal(dsp_amd.make_blackman_window(32), oracle.make_blackman_window(32))


@pytest.mark.gpu
def test_firdesign_demo_matches_oracle():
    fresp = oracle.make_multipass_fresp(241, CUTS, 48000.0)
    kern = oracle.make_filter_kernel(fresp)
    achieved = oracle.ps_abs(oracle.dft(kern, -1, "direct"))
    rows = np.array([[float(v) for v in ln.split()] for ln in _run("firdesign_demo").splitlines()])
    assert rows.shape == (241, 5)
    np.testing.assert_allclose(rows[:, 0], 100.0 * np.arange(241), atol=1e-6)
    assert np.array_equal(rows[:, 1], fresp)
    np.testing.assert_allclose(rows[:, 2], kern.real[:241], atol=1e-6)
    np.testing.assert_allclose(rows[:, 3], kern.imag[:241], atol=1e-6)
    np.testing.assert_allclose(rows[:, 4], achieved[:241], atol=1e-6)


def _table(tmp_path, cols):
    """A text table in the tools' format (util/common/csv.c): comment and blank lines ignored."""
    path = os.path.join(str(tmp_path), "table.txt")
    with open(path, "w") as f:
        f.write("# t x\n\n")
        for i, row in enumerate(zip(*cols)):
            f.write(" ".join("%.17g" % v for v in row) + "\n")
            if i == 3:
                f.write("# a comment in the middle\n\n")
    return path


def _tool(name, *args):
    _build()
    return subprocess.run([os.path.join(BIN, name)] + list(args), capture_output=True, text=True, timeout=120)


def test_tools_reject_bad_command_lines(tmp_path):
    """Argument and window checks happen before any device work (util/fft/fft.c:262-326)."""
    t = np.arange(100) / 1000.0
    path = _table(tmp_path, [t, np.sin(t)])
    for tool in ("fft_tool", "filter_tool"):
        assert _tool(tool, "-ifile").returncode == 255                                   # odd count
        assert _tool(tool, "-bogus", "1").returncode == 255
        assert _tool(tool, "-icol", "1", "-fsampl", "1000").returncode == 255            # no -ifile
        assert _tool(tool, "-ifile", path, "-fsampl", "1000").returncode == 255          # no -icol
        assert _tool(tool, "-ifile", path + ".missing", "-icol", "1", "-fsampl", "1000").returncode == 255
        # the default window end is fsampl * rows seconds: longer than any table
        assert _tool(tool, "-ifile", path, "-icol", "1", "-fsampl", "1000").returncode == 255
    assert _tool("fft_tool", "-ifile", path, "-icol", "1", "-fsampl", "1000", "-tsampl_lo", "0.2",
                 "-tsampl_hi", "0.3").returncode == 255                                   # starts past the end


@pytest.mark.gpu
def test_fft_tool_matches_oracle(tmp_path):
    fs = 48000.0
    tg = oracle.Tonegen()
    tg.add(fs, 1000.0, 1.0, 0.0); tg.add(fs, 12000.0, 0.5, 0.3)
    x = tg.read(1200)
    path = _table(tmp_path, [np.arange(1200) / fs, x])
    r = _tool("fft_tool", "-ifile", path, "-icol", "0x1", "-fsampl", "48000", "-tsampl_lo", "0.005", "-tsampl_hi", "0.015")
    assert r.returncode == 0, r.stderr
    rows = np.array([[float(v) for v in ln.split()] for ln in r.stdout.splitlines()])
    n = 480                                                    # ceil(0.01 * 48000): not a power of two
    assert rows.shape == (n // 2 + 1, 2)
    np.testing.assert_allclose(rows[:, 0], np.arange(n // 2 + 1) * fs / n, atol=1e-6)
    want = oracle.power_spectrum(oracle.fft_r2c(x[240:240 + n]))
    np.testing.assert_allclose(rows[:, 1], want, atol=1e-6)


@pytest.mark.gpu
def test_filter_tool_matches_oracle(tmp_path):
    fs = 48000.0
    tg = oracle.Tonegen()
    tg.add(fs, 1000.0, 1.0, 0.0); tg.add(fs, 12000.0, 0.5, 0.3)
    x = tg.read(5000)
    path = _table(tmp_path, [np.arange(5000) / fs, x])
    r = _tool("filter_tool", "-ifile", path, "-icol", "1", "-fsampl", "48000", "-tsampl", "0.01:0.0953333:",
              "-filter", "3000:24000:0:", "-filter", "3000:3500:0.5:")
    assert r.returncode == 0, r.stderr
    rows = np.array([[float(v) for v in ln.split()] for ln in r.stdout.splitlines()])
    first, n = 480, int(np.ceil((0.0953333 - 0.01) * fs))
    assert rows.shape == (n, 3)
    fband = fs / n
    gains = np.ones(n // 2 + 1)
    for lo, hi, g in ((3000.0, 3500.0, 0.5), (3000.0, 24000.0, 0.0)):      # last given, first applied
        gains[int(np.floor(lo / fband)):min(int(np.ceil(hi / fband)), len(gains))] = g
    np.testing.assert_allclose(rows[:, 1], x[first:first + n], atol=6e-7)
    np.testing.assert_allclose(rows[:, 2], oracle.filter(x[first:first + n], gains), atol=6e-7)


def _multi_gpu(tmp_path, mode, nshard, *sizes):
    """demos/multi_gpu_demo.c: one host thread per shard, each with its own plan on its own device
    (round robin: on a one-GPU box both shards share device 0).  Returns (input, output) as f32."""
    path = os.path.join(str(tmp_path), "%s_%d.bin" % (mode, nshard))
    r = _tool("multi_gpu_demo", mode, str(nshard), path, *[str(v) for v in sizes])
    assert r.returncode == 0, r.stderr
    assert ("%d shards" % nshard) in r.stdout
    both = np.fromfile(path, dtype=np.float32)
    return both[:len(both) // 2], both[len(both) // 2:], r.stdout


def _lowpass(nh):
    k = np.arange(nh, dtype=np.float64) - (nh - 1) / 2.0
    h = np.sinc(2.0 * 6000.0 / 48000.0 * k) * (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(nh) / (nh - 1)))
    return (h / h.sum()).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("nshard", [1, 2, 3])
def test_multi_gpu_demo_shards_channels_and_frames_from_c(tmp_path, nshard):
    """Channel and frame sharding through include/dspmi.h alone (no Python, no collective): every
    shard's rows against the oracle, whatever the number of shards."""
    nchan, nx, nh = 7, 20000, 127
    x, y, _ = _multi_gpu(tmp_path, "channels", nshard, nchan, nx, nh)
    want = oracle.fir_batch_f32(x.reshape(nchan, nx), _lowpass(nh), nx)
    for c in range(nchan):
        for a in range(0, nx, 4096):
            b = min(a + 4096, nx)
            err = np.linalg.norm(y.reshape(nchan, nx)[c, a:b] - want[c, a:b]) / np.linalg.norm(want[c, a:b])
            assert err <= 1e-6, (c, a, err)
    nframe, n = 37, 1024
    z, out, _ = _multi_gpu(tmp_path, "frames", nshard, nframe, n)
    z = z.reshape(nframe, n, 2).astype(np.float64)
    out = out.reshape(nframe, n, 2)
    for f in (0, nframe // 2, nframe - 1):
        want = oracle.dft(z[f, :, 0] + 1j * z[f, :, 1], -1, "direct")
        got = out[f, :, 0] + 1j * out[f, :, 1]
        assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-5, f


@pytest.mark.gpu
@pytest.mark.parametrize("nshard,nh", [(2, 8191), (4, 2049), (1, 8191)])
def test_multi_gpu_demo_splits_one_long_signal_from_c(tmp_path, nshard, nh):
    """One signal, one long filter, one output segment per shard: a shard's slice carries the
    nh-1 samples before its segment as history (fft/src/main.c:193-204 is the sum), and the
    segments put together are the filtered signal -- per 4096-sample block against the oracle."""
    nx = 262144 + 5
    x, y, text = _multi_gpu(tmp_path, "split", nshard, nx, nh)
    assert ("outputs [0, %d)" % (nx // nshard if nshard > 1 else nx)) in text
    want = oracle.fir_batch_f32(x[None, :], _lowpass(nh), nx)[0]
    scale = np.linalg.norm(want) / np.sqrt(nx)
    for a in range(0, nx, 4096):
        b = min(a + 4096, nx)
        # (a row's first nh-1 outputs are a start-up transient: held to the steady-state blocks' absolute error)
        err = np.linalg.norm(y[a:b] - want[a:b])
        assert err <= 1e-6 * max(np.linalg.norm(want[a:b]), scale * np.sqrt(b - a)), (a, err)
