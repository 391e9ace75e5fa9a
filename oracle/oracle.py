"""ctypes bindings for oracle/dsp_oracle.c (and oracle/_ref when built).

TEST INFRASTRUCTURE ONLY -- see the header of dsp_oracle.c.  Each wrapper
names the reference routine its C counterpart restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u = C.c_uint


def build(quiet=True):
    """(Re)build liboracle*.so, and oracle/_ref when /root/reference exists."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


_libs = {}


def lib(variant=""):
    """variant: "" (parity checker), "_O0" (as-shipped flags), "_fast"."""
    if variant not in _libs:
        L = _load("liboracle%s.so" % variant)
        L.oracle_conv0.argtypes = [_dp, _u, _dp, _u, _dp]
        L.oracle_conv1.argtypes = [_dp, _u, _dp, _u, _dp]
        L.oracle_convolve.argtypes = [_dp, _u, _dp, _u, _dp, _u]
        L.oracle_fir_batch_f32.argtypes = [_dp, C.c_size_t, _u, _fp, C.c_size_t,
                                           _u, _fp, _u, _u, C.c_int]
        L.oracle_fir_stream_f32.argtypes = [_dp, _fp, _u, _fp, _fp, _u]
        L.oracle_dft_direct.argtypes = [_dp, _u, _dp, C.c_int]
        L.oracle_fft_radix2.argtypes = [_dp, _u, _dp, C.c_int]
        L.oracle_dft.argtypes = [_dp, _u, _dp, C.c_int]
        L.oracle_fft_r2c.argtypes = [_dp, _dp, _u]
        L.oracle_fft_c2r.argtypes = [_dp, _dp, _u]
        L.oracle_filter.argtypes = [_dp, _dp, _u, _dp]
        L.oracle_power_spectrum.argtypes = [_dp, _dp, _u]
        L.oracle_ps_abs.argtypes = [_dp, _u, _dp]
        L.oracle_ps_percent.argtypes = [_dp, _u, _dp]
        L.oracle_compute_phi.argtypes = [_dp, _u, _dp]
        L.oracle_alsa_power_spectrum.argtypes = [_dp, _dp, _u]
        L.oracle_fft_batch.argtypes = [_dp, _dp, _u, _u, C.c_int, C.c_int]
        L.oracle_convolve_batch.argtypes = [_dp, _u, _dp, _u, _dp, _u, _u]
        L.oracle_sin_tonegen_read.argtypes = [_dp, _dp, _u, _dp, _u]
        _sp = C.POINTER(C.c_short)
        L.oracle_alsa_do_fir.argtypes = [_sp, _sp, _u, _dp, _u]
        L.oracle_alsa_do_power_spectrum.argtypes = [_dp, _sp, _u]
        L.oracle_double_to_s16x2.argtypes = [_sp, _dp, _u]
        L.oracle_make_multipass_fresp.argtypes = [_dp, _u, _dp, _u, C.c_double]
        L.oracle_make_blackman_window.argtypes = [_dp, _u]
        L.oracle_make_filter_kernel.argtypes = [_dp, _u, _dp]
        for f in ("oracle_conv0", "oracle_conv1", "oracle_convolve",
                  "oracle_fir_batch_f32", "oracle_fir_stream_f32",
                  "oracle_dft_direct", "oracle_fft_radix2", "oracle_dft",
                  "oracle_fft_r2c", "oracle_fft_c2r", "oracle_filter",
                  "oracle_power_spectrum", "oracle_ps_abs", "oracle_ps_percent",
                  "oracle_compute_phi", "oracle_alsa_power_spectrum",
                  "oracle_fft_batch", "oracle_convolve_batch",
                  "oracle_sin_tonegen_read", "oracle_alsa_do_fir",
                  "oracle_alsa_do_power_spectrum", "oracle_double_to_s16x2",
                  "oracle_make_multipass_fresp", "oracle_make_blackman_window",
                  "oracle_make_filter_kernel"):
            getattr(L, f).restype = None
        _libs[variant] = L
    return _libs[variant]


def ref_lib():
    """The reference's own conv0/conv1 (oracle/_ref/libref_conv.so) or None."""
    path = os.path.join(_HERE, "_ref", "libref_conv.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    for f in ("ref_conv0", "ref_conv1"):
        getattr(L, f).argtypes = [_dp, _u, _dp, _u, _dp]
        getattr(L, f).restype = None
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


# ---- FIR / convolution ---------------------------------------------------

def conv0(x, h, variant=""):
    """convolution/main.c:15-33."""
    x, xp = _d(x); h, hp = _d(h)
    y = np.empty(len(x)); lib(variant).oracle_conv0(xp, len(x), hp, len(h), y.ctypes.data_as(_dp))
    return y


def conv1(x, h, variant=""):
    """convolution/main.c:36-53 (bug-compatible: drops the j == i term)."""
    x, xp = _d(x); h, hp = _d(h)
    y = np.empty(len(x)); lib(variant).oracle_conv1(xp, len(x), hp, len(h), y.ctypes.data_as(_dp))
    return y


def convolve(x, h, nxx=None, variant=""):
    """fir/src/main.c:258-280; nxx=None gives the nx+nh length of fft/src/main.c:190."""
    x, xp = _d(x); h, hp = _d(h)
    if nxx is None:
        nxx = len(x) + len(h)
    y = np.empty(nxx); lib(variant).oracle_convolve(y.ctypes.data_as(_dp), nxx, xp, len(x), hp, len(h))
    return y


def ref_conv0(x, h):
    x, xp = _d(x); h, hp = _d(h)
    y = np.empty(len(x)); ref_lib().ref_conv0(xp, len(x), hp, len(h), y.ctypes.data_as(_dp))
    return y


def ref_conv1(x, h):
    x, xp = _d(x); h, hp = _d(h)
    y = np.empty(len(x)); ref_lib().ref_conv1(xp, len(x), hp, len(h), y.ctypes.data_as(_dp))
    return y


def fir_batch_f32(x, h, nxx, drop_first=False):
    """Double-accumulated checker for the batched f32 device API.
    x: [nchan, nx] float32 (C-contiguous); returns [nchan, nxx] float64."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim == 1:
        x = x[None, :]
    h, hp = _f(h)
    nchan, nx = x.shape
    y = np.empty((nchan, nxx))
    lib().oracle_fir_batch_f32(y.ctypes.data_as(_dp), nxx, nxx, x.ctypes.data_as(_fp), nx, nx,
                               hp, len(h), nchan, int(bool(drop_first)))
    return y


def fir_stream_f32(x, hist, h):
    x, xp = _f(x); hist, hip = _f(hist); h, hp = _f(h)
    assert len(hist) == len(h) - 1
    y = np.empty(len(x))
    lib().oracle_fir_stream_f32(y.ctypes.data_as(_dp), xp, len(x), hip, hp, len(h))
    return y


# ---- DFT ("parity unpinned" for the C demos: FFTW is absent, see dsp_oracle.c; the fftpy
# path, whose FFT is numpy.fft.rfft, is pinned by tests/golden/fftpy_spectrum.npz) -------------

def _c2d(z):
    z = np.ascontiguousarray(z, dtype=np.complex128)
    return z, z.view(np.float64).ctypes.data_as(_dp)


def dft(z, sign=-1, method="auto"):
    """fftw_plan_dft_1d semantics (fftc/main.c:31-43): unnormalised, sign -1 forward."""
    z, zp = _c2d(z)
    out = np.empty(len(z), dtype=np.complex128)
    fn = {"auto": lib().oracle_dft, "direct": lib().oracle_dft_direct,
          "radix2": lib().oracle_fft_radix2}[method]
    fn(zp, len(z), out.view(np.float64).ctypes.data_as(_dp), sign)
    return out


def fft_r2c(x):
    """fft/src/main.c:146-168: bins 0..n/2."""
    x, xp = _d(x)
    out = np.empty(len(x) // 2 + 1, dtype=np.complex128)
    lib().oracle_fft_r2c(out.view(np.float64).ctypes.data_as(_dp), xp, len(x))
    return out


def fft_c2r(bins, n):
    bins, bp = _c2d(bins)
    assert len(bins) == n // 2 + 1
    x = np.empty(n)
    lib().oracle_fft_c2r(x.ctypes.data_as(_dp), bp, n)
    return x


def filter(x, coeffs):  # noqa: A001 - the reference's name
    """util/filter/filter.c:73-105."""
    x, xp = _d(x); coeffs, cp = _d(coeffs)
    assert len(coeffs) == len(x) // 2 + 1
    xx = np.empty(len(x))
    lib().oracle_filter(xx.ctypes.data_as(_dp), xp, len(x), cp)
    return xx


# ---- spectrum post-ops ----------------------------------------------------

def power_spectrum(bins):
    """fft/src/main.c:120-141."""
    bins, bp = _c2d(bins)
    ps = np.empty(len(bins)); lib().oracle_power_spectrum(ps.ctypes.data_as(_dp), bp, len(bins))
    return ps


def ps_abs(z):
    """fftc/main.c:80-105."""
    z, zp = _c2d(z)
    out = np.empty(len(z) // 2); lib().oracle_ps_abs(zp, len(z), out.ctypes.data_as(_dp))
    return out


def ps_percent(z):
    """fftc/main.c:46-77."""
    z, zp = _c2d(z)
    out = np.empty(len(z) // 2); lib().oracle_ps_percent(zp, len(z), out.ctypes.data_as(_dp))
    return out


def compute_phi(z):
    """fftc/main.c:108-123."""
    z, zp = _c2d(z)
    out = np.empty(len(z) // 2); lib().oracle_compute_phi(zp, len(z), out.ctypes.data_as(_dp))
    return out


def fftpy_power_spectrum(x):
    """compute_power_spectrum of fftpy/main.py:30-45: |rfft(x, norm='ortho')| divided by its sum
    over the n/2+1 bins (the 1/sqrt(n) of 'ortho' cancels), through this checker's own transform."""
    mag = np.abs(fft_r2c(x))
    q = 0.0
    for m in mag:          # the reference's running sum, bin by bin
        q += m
    return mag / q


def alsa_power_spectrum(z):
    """alsa/main.c:473-487 over the first len(z)//2 bins."""
    z, zp = _c2d(z)
    nb = len(z) // 2
    out = np.empty(nb); lib().oracle_alsa_power_spectrum(out.ctypes.data_as(_dp), zp, nb)
    return out


_sp = C.POINTER(C.c_short)


def alsa_do_fir(frames, h):
    """alsa/main.c:518-540 on one block of interleaved stereo s16 frames [nsampl, 2]."""
    frames = np.ascontiguousarray(frames, dtype=np.int16)
    h = np.ascontiguousarray(h, dtype=np.float64)
    out = np.empty_like(frames)
    lib().oracle_alsa_do_fir(out.ctypes.data_as(_sp), frames.ctypes.data_as(_sp), frames.shape[0],
                             h.ctypes.data_as(_dp), len(h))
    return out


def alsa_do_power_spectrum(frames):
    """alsa/main.c:453-492 on one block of interleaved stereo s16 frames [nsampl, 2]."""
    frames = np.ascontiguousarray(frames, dtype=np.int16)
    out = np.empty(frames.shape[0] // 2)
    lib().oracle_alsa_do_power_spectrum(out.ctypes.data_as(_dp), frames.ctypes.data_as(_sp), frames.shape[0])
    return out


def double_to_s16x2(gen):
    """tonegen/main.c:14-24."""
    gen = np.ascontiguousarray(gen, dtype=np.float64)
    out = np.empty((len(gen), 2), dtype=np.int16)
    lib().oracle_double_to_s16x2(out.ctypes.data_as(_sp), gen.ctypes.data_as(_dp), len(gen))
    return out


def make_multipass_fresp(nx, fcuts, fsampl):
    """fir/src/main.c:108-135."""
    fc = np.ascontiguousarray(fcuts, dtype=np.float64)
    x = np.empty(nx)
    lib().oracle_make_multipass_fresp(x.ctypes.data_as(_dp), nx, fc.ctypes.data_as(_dp), len(fc), fsampl)
    return x


def make_blackman_window(nx):
    """fir/src/main.c:138-146."""
    x = np.empty(nx)
    lib().oracle_make_blackman_window(x.ctypes.data_as(_dp), nx)
    return x


def make_filter_kernel(fresp):
    """fir/src/main.c:149-226."""
    f = np.ascontiguousarray(fresp, dtype=np.float64)
    k = np.empty(2 * len(f), dtype=np.complex128)
    lib().oracle_make_filter_kernel(f.ctypes.data_as(_dp), len(f), k.view(np.float64).ctypes.data_as(_dp))
    return k


# ---- tone generator ---------------------------------------------------------

class _Tonegen(C.Structure):
    _fields_ = [("n", C.c_uint), ("a", C.c_double * 32), ("phi", C.c_double * 32),
                ("w", C.c_double * 32), ("dw", C.c_double * 32)]


class Tonegen:
    """tonegen/tonegen.c:5-59 (argument order of add: fsampl, freq, a, phi)."""

    def __init__(self):
        L = lib()
        L.oracle_tonegen_init.argtypes = [C.POINTER(_Tonegen)]
        L.oracle_tonegen_add.argtypes = [C.POINTER(_Tonegen)] + [C.c_double] * 4
        L.oracle_tonegen_read.argtypes = [C.POINTER(_Tonegen), _dp, _u]
        L.oracle_tonegen_read_complex.argtypes = [C.POINTER(_Tonegen), _dp, _u]
        for f in ("init", "add", "read", "read_complex"):
            getattr(L, "oracle_tonegen_" + f).restype = None
        self._L = L
        self.g = _Tonegen()
        L.oracle_tonegen_init(C.byref(self.g))

    def add(self, fsampl, freq, a, phi):
        self._L.oracle_tonegen_add(C.byref(self.g), fsampl, freq, a, phi)

    def read(self, n):
        buf = np.empty(n); self._L.oracle_tonegen_read(C.byref(self.g), buf.ctypes.data_as(_dp), n)
        return buf

    def read_complex(self, n):
        buf = np.empty(n, dtype=np.complex128)
        self._L.oracle_tonegen_read_complex(C.byref(self.g), buf.view(np.float64).ctypes.data_as(_dp), n)
        return buf


def sin_tonegen(freqs, fsampl, n):
    """fft/src/main.c:265-291: unit-amplitude sines, phase recurrence from 0."""
    w = np.zeros(len(freqs)); dw = np.array([(2.0 * np.pi * f) / fsampl for f in freqs])
    buf = np.empty(n)
    lib().oracle_sin_tonegen_read(w.ctypes.data_as(_dp), dw.ctypes.data_as(_dp), len(freqs),
                                  buf.ctypes.data_as(_dp), n)
    return buf


# ---- tap tables that live in the reference as data ---------------------------

# alsa/main.c:389-407 -- the only concrete 16-tap set in the reference
# (fir/lowpass_6000.gmeteor's output, /tmp/bar.h, cannot be regenerated: gmeteor is absent).
TAPS16 = np.array([
    -0.131625603115805, 0.103820148451119, 0.096541374482813, 0.102280926834003,
    0.113689768997426, 0.125850624234278, 0.135740574367069, 0.141133806318268,
    0.141133806318268, 0.135740574367069, 0.125850624234278, 0.113689768997426,
    0.102280926834003, 0.096541374482813, 0.103820148451119, -0.131625603115805])

# fft/src/main.c:216-236 -- the 21-tap table of the disabled branch
TAPS21 = np.array([
    1964.09456418888, -10208.338288738, 16795.2945884697, -0.0, -23281.999649468,
    0.0572589041085133, 37237.1250088098, -0.0, -50515.8159352444, -0.0, 56020.202875443,
    -0.0, -50515.8159352444, -0.0, 37237.1250088098, 0.0572589041085133, -23281.999649468,
    -0.0, 16795.2945884697, -10208.338288738, 1964.09456418888])

# convolution/main.c:83-84 and the table its main() prints (y0, y1 columns)
KAT9_X = np.array([1, -0.5, -0.25, 0, 0, 0, 0, 0, 0], dtype=np.float64)
KAT9_H = np.array([0, -1, -1.25, 2, 1.3, 1.3, 0.75, 0, -0.75], dtype=np.float64)


def windowed_sinc_taps(nh, fc=6000.0, fs=48000.0):
    """Hamming-windowed-sinc low-pass, sum normalised to 1, computed in double.
    Synthetic taps for the 127- and 8191-tap workloads (the reference holds no
    table of that length)."""
    n = np.arange(nh, dtype=np.float64) - (nh - 1) / 2.0
    h = np.sinc(2.0 * fc / fs * n) * (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(nh) / (nh - 1)))
    return h / h.sum()
