"""ctypes bindings for libdspmi.so.

Two groups, mirroring include/dspmi.h:
  * FirPlan / FftPlan: the device-resident batched API, driven with torch CUDA
    tensors (only their data_ptr() and the current stream cross the boundary);
  * conv0, conv1, convolve, fir, dft, idft, fft_r2c, fft_mag, filter: the
    reference's host routines (same names, argument meaning and order as the C
    code they replace), taking and returning numpy float64 arrays.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FORWARD, BACKWARD = -1, +1
F32, F64 = 0, 1
FIR_DROP_FIRST = 1


class DspmiError(RuntimeError):
    pass


def lib_path():
    # DSPMI_LIB: another build of the same library (A/B timing of two kernel versions on one box)
    return os.environ.get("DSPMI_LIB") or os.path.join(_HERE, "libdspmi.so")


_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
_u = C.c_uint
_sz = C.c_size_t

_SIGS = {
    "dspmi_last_error": (C.c_char_p, []),
    "dspmi_abi_version": (C.c_int, []),
    "dspmi_device_count": (C.c_int, []),
    "dspmi_fir_create_f32": (C.c_int, [C.POINTER(_vp), _fp, _u, C.c_int]),
    "dspmi_fir_create_f32_opts": (C.c_int, [C.POINTER(_vp), _fp, _u, C.c_int, _u]),
    "dspmi_fir_create_f64": (C.c_int, [C.POINTER(_vp), _dp, _u, C.c_int]),
    "dspmi_fir_destroy": (C.c_int, [_vp]),
    "dspmi_fir_exec_f32": (C.c_int, [_vp, _vp, _sz, _u, _vp, _sz, _u, _u, _vp, _sz, _u, _vp]),
    "dspmi_fir_exec_f64": (C.c_int, [_vp, _vp, _sz, _u, _vp, _sz, _u, _u, _vp, _sz, _u, _vp]),
    "dspmi_fir_exec_s16x2": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _u, _u, _vp]),
    "dspmi_fir_variant": (C.c_char_p, [_vp]),
    "dspmi_fft_create": (C.c_int, [C.POINTER(_vp), _u, C.c_int, C.c_int]),
    "dspmi_fft_destroy": (C.c_int, [_vp]),
    "dspmi_fft_exec_c2c": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, _vp]),
    "dspmi_fft_exec_c2c_noalloc": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, _vp]),
    "dspmi_fft_exec_r2c": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "dspmi_fft_exec_c2r": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "dspmi_fft_exec_filter": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "dspmi_fft_variant": (C.c_char_p, [_vp]),
    "dspmi_ols_create_f32": (C.c_int, [C.POINTER(_vp), _fp, _u, _u, C.c_int]),
    "dspmi_ols_destroy": (C.c_int, [_vp]),
    "dspmi_ols_fft_length": (_u, [_vp]),
    "dspmi_ols_exec_f32": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _u, _u, _vp, _sz, _vp]),
    "dspmi_fft_exec_r2c_spectrum": (C.c_int, [_vp, _vp, _vp, _sz, _u, C.c_int, C.c_float, _vp]),
    "dspmi_fir_spectrum_f32": (C.c_int, [_vp, _vp, _vp, _u, C.c_int, C.c_float, _vp, _sz, _u, _u, _vp, _sz, _vp]),
    "dspmi_tone_fir_spectrum_work_bytes": (_sz, [_u, _u, _u]),
    "dspmi_tone_fir_spectrum_f32": (C.c_int, [_vp, _vp, _vp, _u, C.c_int, C.c_float, _u, _u, _u, _vp, _vp, _vp, _vp, C.c_int,
                                              _vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    "dspmi_fft_exec_r2c_spectrum_s16x2": (C.c_int, [_vp, _vp, _vp, _sz, _u, C.c_int, C.c_float, _vp]),
    "dspmi_power_spectrum_f32": (C.c_int, [_vp, _vp, _u, _u, _sz, C.c_int, C.c_float, _vp]),
    "dspmi_tonegen_f32": (C.c_int, [_vp, _sz, _u, _u, _u, _vp, _vp, _vp, _vp, C.c_int, _u, _vp]),
    "dspmi_tonegen_c32": (C.c_int, [_vp, _sz, _u, _u, _u, _vp, _vp, _vp, _vp, C.c_int, _u, _vp]),
    "dspmi_tonegen_s16x2": (C.c_int, [_vp, _sz, _u, _u, _u, _vp, _vp, _vp, _vp, C.c_int, _u, _vp]),
    "dspmi_dropin_release": (None, []),
    "dsp_conv0": (None, [_dp, _u, _dp, _u, _dp]),
    "dsp_conv1": (None, [_dp, _u, _dp, _u, _dp]),
    "dsp_convolve": (None, [_dp, _u, _dp, _u, _dp, _u]),
    "dsp_convolve_alloc": (None, [C.POINTER(_dp), C.POINTER(_u), _dp, _u, _dp, _u]),
    "dsp_fir_set_taps": (None, [_dp, _u]),
    "dsp_fir": (None, [C.POINTER(_dp), C.POINTER(_u), _dp, _u]),
    "dsp_dft": (None, [_dp, _u, _dp]),
    "dsp_idft": (None, [_dp, _u, _dp]),
    "dsp_fft_r2c": (None, [_dp, _dp, _u]),
    "dsp_fft_mag": (None, [_dp, _u, _dp]),
    "dsp_filter": (None, [_dp, _dp, _u, _dp]),
    "dsp_ps_percent": (None, [_dp, _u, _dp]),
    "dsp_ps_abs": (None, [_dp, _u, _dp]),
    "dsp_compute_phi": (None, [_dp, _u, _dp]),
    "dspmi_phase_f32": (C.c_int, [_vp, _vp, _u, _u, _sz, _vp]),
    "dsp_make_blackman_window": (None, [_dp, _u]),
    "dsp_make_filter_kernel": (None, [_dp, _u, _dp]),
}


def exported_symbols():
    """Every entry point include/dspmi.h declares."""
    return sorted(_SIGS)


def lib():
    """Load libdspmi.so (once).  Raises if it is missing: nothing here falls back to the CPU."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise DspmiError("%s not found: build it with `python -m dsp_amd.build` "
                             "(or __graft_entry__.build())" % path)
        # This harness hands torch's device buffers to the library, so both must talk to one HIP
        # runtime: torch brings its own copy, and whichever copy is loaded second finds no device.
        # Importing torch first lets the library's libamdhip64 dependency resolve to the copy
        # already in the process.  (A C program linking libdspmi.so has one runtime anyway.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if not os.environ.get("DSPMI_LIB"):
                    raise
                continue   # another build, for A/B timing: it may predate an entry point
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise DspmiError("%s failed (%d): %s" % (what, rc, lib().dspmi_last_error().decode()))


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------------------
# device-resident batched API
# ---------------------------------------------------------------------------

class FirPlan:
    """Taps resident on the device; exec() runs the batched FIR on torch CUDA tensors."""

    FULL_RANGE, F32_PRODUCTS = 1, 2   # creation options of f32 plans (dspmi.h)

    def __init__(self, taps, dtype="f32", device=-1, options=0):
        self.f64 = dtype in ("f64", np.float64)
        self.nh = len(taps)
        self._h = C.c_void_p()
        if self.f64:
            t = np.ascontiguousarray(taps, dtype=np.float64)
            _check(lib().dspmi_fir_create_f64(C.byref(self._h), t.ctypes.data_as(_dp), self.nh, device), "fir_create_f64")
        else:
            t = np.ascontiguousarray(taps, dtype=np.float32)
            if options == 0:
                _check(lib().dspmi_fir_create_f32(C.byref(self._h), t.ctypes.data_as(_fp), self.nh, device), "fir_create_f32")
            else:
                _check(lib().dspmi_fir_create_f32_opts(C.byref(self._h), t.ctypes.data_as(_fp), self.nh, device, options),
                       "fir_create_f32_opts")

    @property
    def variant(self):
        return lib().dspmi_fir_variant(self._h).decode()

    def exec(self, x, nxx=None, y=None, hist=None, flags=0):
        """x: [nchan, nx] CUDA tensor (last dim contiguous).  Returns y [nchan, nxx]."""
        import torch
        want = torch.float64 if self.f64 else torch.float32
        assert x.is_cuda and x.dtype == want and x.dim() == 2 and x.stride(1) == 1
        nchan, nx = x.shape
        if nxx is None:
            nxx = nx
        if y is None:
            y = torch.empty((nchan, nxx), dtype=want, device=x.device)
        assert y.is_cuda and y.dtype == want and y.shape == (nchan, nxx) and y.stride(1) == 1
        hp, hs = None, 0
        if hist is not None:
            assert hist.is_cuda and hist.dtype == want and hist.shape == (nchan, self.nh - 1) and hist.stride(1) == 1
            hp, hs = hist.data_ptr(), hist.stride(0) if nchan > 1 else self.nh - 1
        fn = lib().dspmi_fir_exec_f64 if self.f64 else lib().dspmi_fir_exec_f32
        _check(fn(self._h, y.data_ptr(), y.stride(0) if nchan > 1 else nxx, nxx,
                  x.data_ptr(), x.stride(0) if nchan > 1 else nx, nx, nchan, hp, hs, flags, _stream()), "fir_exec")
        return y

    def spectrum(self, fft, x, hist=None, nbin=None, mode=0, scale=1.0, out=None):
        """Filter, then the normalised magnitude spectrum of every 4096 filtered samples, in one
        kernel (the filtered samples are never written): x [nchan, nx], nx a multiple of 4096,
        fft an f32 FftPlan(4096) -> [nchan * nx / 4096, nbin].  Plans of up to 305 taps."""
        import torch
        assert not self.f64 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        nchan, nx = x.shape
        if nbin is None:
            nbin = fft.n // 2 + 1
        if out is None:
            out = torch.empty((nchan * (nx // fft.n), nbin), dtype=torch.float32, device=x.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == nchan * (nx // fft.n) * nbin
        hp, hs = None, 0
        if hist is not None:
            assert hist.is_cuda and hist.dtype == torch.float32 and hist.shape == (nchan, self.nh - 1) and hist.stride(1) == 1
            hp, hs = hist.data_ptr(), hist.stride(0) if nchan > 1 else self.nh - 1
        _check(lib().dspmi_fir_spectrum_f32(self._h, fft._h, out.data_ptr(), nbin, mode, scale, x.data_ptr(),
                                            x.stride(0) if nchan > 1 else nx, nx, nchan, hp, hs, _stream()), "fir_spectrum")
        return out

    def tone_spectrum(self, fft, gen, nx, hist=None, hist_out=None, nbin=None, mode=0, scale=1.0, out=None, work=None):
        """The next nx samples of every channel of the ToneGen `gen`, filtered, as normalised
        magnitude spectra of 4096 samples each -- generator, filter and transform in one kernel,
        no sample ever written to memory.  hist / hist_out: [nchan, nh-1], the samples before this
        call (None: zeros) and the ones the next call continues from.  work: a uint8 CUDA tensor of
        tone_spectrum_work_bytes() bytes (allocated when None).  Returns [nchan * nx / 4096, nbin]."""
        import torch
        assert not self.f64 and nx % fft.n == 0
        nchan = gen.nchan
        dev = gen.w.device
        if nbin is None:
            nbin = fft.n // 2 + 1
        if out is None:
            out = torch.empty((nchan * (nx // fft.n), nbin), dtype=torch.float32, device=dev)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == nchan * (nx // fft.n) * nbin
        need = lib().dspmi_tone_fir_spectrum_work_bytes(nx, nchan, gen.ntone)
        if work is None:
            work = torch.empty((max(need, 1),), dtype=torch.uint8, device=dev)
        assert work.is_cuda and work.dtype == torch.uint8 and work.is_contiguous()
        ptrs = []
        for h in (hist, hist_out):
            if h is None:
                ptrs += [None, 0]
            else:
                assert h.is_cuda and h.dtype == torch.float32 and h.shape == (nchan, self.nh - 1) and h.stride(1) == 1
                ptrs += [h.data_ptr(), h.stride(0) if nchan > 1 else self.nh - 1]
        _check(lib().dspmi_tone_fir_spectrum_f32(self._h, fft._h, out.data_ptr(), nbin, mode, scale, nx, nchan, gen.ntone,
                                                 gen.w.data_ptr(), gen.dw.data_ptr(), gen.amp.data_ptr(), gen.phi.data_ptr(),
                                                 int(gen.use_sin), ptrs[0], ptrs[1], ptrs[2], ptrs[3], work.data_ptr(),
                                                 work.numel(), _stream()), "tone_fir_spectrum")
        return out

    def exec_s16x2(self, frames, out=None):
        """frames: [nblock, nsampl, 2] int16 CUDA tensor of stereo capture blocks; every block is
        filtered from zero state and comes back as int16 in both channels (f64 plans only)."""
        import torch
        assert frames.is_cuda and frames.dtype == torch.int16 and frames.dim() == 3 and frames.shape[2] == 2
        assert frames.stride(2) == 1 and frames.stride(1) == 2 and frames.stride(0) % 2 == 0
        nblock, nsampl = frames.shape[0], frames.shape[1]
        if out is None:
            out = torch.empty_like(frames)
        assert out.is_cuda and out.dtype == torch.int16 and out.shape == frames.shape
        assert out.stride(2) == 1 and out.stride(1) == 2 and out.stride(0) % 2 == 0
        _check(lib().dspmi_fir_exec_s16x2(self._h, out.data_ptr(), out.stride(0) // 2 if nblock > 1 else nsampl,
                                          frames.data_ptr(), frames.stride(0) // 2 if nblock > 1 else nsampl,
                                          nsampl, nblock, _stream()), "fir_exec_s16x2")
        return out

    def close(self):
        if self._h:
            lib().dspmi_fir_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FftPlan:
    """Twiddles + kernel choice for one length; exec on torch CUDA complex/real tensors."""

    def __init__(self, n, dtype="f32", device=-1):
        self.f64 = dtype in ("f64", np.float64)
        self.n = n
        self._h = C.c_void_p()
        _check(lib().dspmi_fft_create(C.byref(self._h), n, F64 if self.f64 else F32, device), "fft_create")

    @property
    def variant(self):
        return lib().dspmi_fft_variant(self._h).decode()

    def _types(self):
        import torch
        return (torch.complex128, torch.float64) if self.f64 else (torch.complex64, torch.float32)

    def c2c(self, x, sign=FORWARD, out=None):
        """x: [batch, n] complex CUDA tensor, contiguous.  Unnormalised both ways."""
        import torch
        ct, _ = self._types()
        assert x.is_cuda and x.dtype == ct and x.is_contiguous() and x.shape[-1] == self.n
        if out is None:
            out = torch.empty_like(x)
        assert out.is_cuda and out.dtype == ct and out.is_contiguous() and out.shape == x.shape
        batch = x.numel() // self.n
        _check(lib().dspmi_fft_exec_c2c(self._h, x.data_ptr(), out.data_ptr(), batch, sign, _stream()), "fft_exec_c2c")
        return out

    def c2c_noalloc(self, x, sign=FORWARD, out=None):
        """c2c through the entry point that allocates no device memory (`out` must be given:
        allocating it here would defeat the purpose)."""
        ct, _ = self._types()
        assert x.is_cuda and x.dtype == ct and x.is_contiguous() and x.shape[-1] == self.n
        assert out is not None and out.is_cuda and out.dtype == ct and out.is_contiguous() and out.shape == x.shape
        _check(lib().dspmi_fft_exec_c2c_noalloc(self._h, x.data_ptr(), out.data_ptr(), x.numel() // self.n, sign, _stream()),
               "fft_exec_c2c_noalloc")
        return out

    def r2c(self, x, out=None):
        import torch
        ct, rt = self._types()
        assert x.is_cuda and x.dtype == rt and x.is_contiguous() and x.shape[-1] == self.n
        batch = x.numel() // self.n
        if out is None:
            out = torch.empty(x.shape[:-1] + (self.n // 2 + 1,), dtype=ct, device=x.device)
        _check(lib().dspmi_fft_exec_r2c(self._h, x.data_ptr(), out.data_ptr(), batch, _stream()), "fft_exec_r2c")
        return out

    def r2c_spectrum(self, x, nbin=None, mode=0, scale=1.0, out=None):
        """Real frames -> normalised magnitude spectrum in one pass (f32 plans)."""
        import torch
        assert not self.f64 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == self.n
        batch = x.numel() // self.n
        if nbin is None:
            nbin = self.n // 2 + 1
        if out is None:
            out = torch.empty(x.shape[:-1] + (nbin,), dtype=torch.float32, device=x.device)
        _check(lib().dspmi_fft_exec_r2c_spectrum(self._h, x.data_ptr(), out.data_ptr(), batch, nbin, mode, scale, _stream()),
               "fft_exec_r2c_spectrum")
        return out

    def r2c_spectrum_s16x2(self, frames, nbin=None, mode=1, scale=1.0, out=None):
        """frames: [batch, n, 2] int16 stereo blocks -> [batch, nbin] magnitude spectrum of the mono mix."""
        import torch
        assert frames.is_cuda and frames.dtype == torch.int16 and frames.is_contiguous()
        assert frames.dim() == 3 and frames.shape[1] == self.n and frames.shape[2] == 2
        batch = frames.shape[0]
        if nbin is None:
            nbin = self.n // 2
        if out is None:
            out = torch.empty((batch, nbin), dtype=torch.float32, device=frames.device)
        assert out.is_contiguous() and out.shape == (batch, nbin) and out.dtype == torch.float32
        _check(lib().dspmi_fft_exec_r2c_spectrum_s16x2(self._h, frames.data_ptr(), out.data_ptr(), batch, nbin, mode, scale,
                                                       _stream()), "fft_exec_r2c_spectrum_s16x2")
        return out

    def filter(self, x, gains, out=None):
        """x: [batch, n] real frames; gains: [n//2 + 1] real per-bin gains -> filtered frames (util/filter/filter.c:73-105)."""
        import torch
        _, rt = self._types()
        assert x.is_cuda and x.dtype == rt and x.is_contiguous() and x.shape[-1] == self.n
        assert gains.is_cuda and gains.dtype == rt and gains.is_contiguous() and gains.numel() == self.n // 2 + 1
        if out is None:
            out = torch.empty_like(x)
        assert out.is_cuda and out.dtype == rt and out.is_contiguous() and out.shape == x.shape
        _check(lib().dspmi_fft_exec_filter(self._h, x.data_ptr(), out.data_ptr(), gains.data_ptr(), x.numel() // self.n,
                                           _stream()), "fft_exec_filter")
        return out

    def c2r(self, x, out=None):
        import torch
        ct, rt = self._types()
        assert x.is_cuda and x.dtype == ct and x.is_contiguous() and x.shape[-1] == self.n // 2 + 1
        batch = x.numel() // (self.n // 2 + 1)
        if out is None:
            out = torch.empty(x.shape[:-1] + (self.n,), dtype=rt, device=x.device)
        _check(lib().dspmi_fft_exec_c2r(self._h, x.data_ptr(), out.data_ptr(), batch, _stream()), "fft_exec_c2r")
        return out

    def close(self):
        if self._h:
            lib().dspmi_fft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PS_NORM_GUARDED, PS_NORM, PS_ABS, PS_PERCENT = 0, 1, 2, 3


def power_spectrum(bins, nbin=None, mode=PS_NORM_GUARDED, scale=1.0, out=None):
    """bins: [batch, nspec] complex64 CUDA tensor; uses the first nbin bins of each frame."""
    import torch
    assert bins.is_cuda and bins.dtype == torch.complex64 and bins.dim() == 2 and bins.is_contiguous()
    batch, nspec = bins.shape
    if nbin is None:
        nbin = nspec
    if out is None:
        out = torch.empty((batch, nbin), dtype=torch.float32, device=bins.device)
    _check(lib().dspmi_power_spectrum_f32(out.data_ptr(), bins.data_ptr(), nbin, nspec, batch, mode, scale, _stream()),
           "power_spectrum")
    return out


def phase(bins, nbin=None, out=None):
    """compute_phi (fftc/main.c:108-123) over the first nbin bins of each frame."""
    import torch
    assert bins.is_cuda and bins.dtype == torch.complex64 and bins.dim() == 2 and bins.is_contiguous()
    batch, nspec = bins.shape
    if nbin is None:
        nbin = nspec
    if out is None:
        out = torch.empty((batch, nbin), dtype=torch.float32, device=bins.device)
    _check(lib().dspmi_phase_f32(out.data_ptr(), bins.data_ptr(), nbin, nspec, batch, _stream()), "phase")
    return out


class ToneGen:
    """Per-channel tone generators with device-resident running phases (tonegen/tonegen.c)."""

    def __init__(self, fsampl, freqs, amps=None, phis=None, use_sin=False, device="cuda"):
        import torch
        freqs = np.atleast_2d(np.asarray(freqs, dtype=np.float64))
        self.nchan, self.ntone = freqs.shape
        self.use_sin = use_sin
        dw = (2.0 * np.pi * freqs) / fsampl          # tonegen/tonegen.c:22
        amps = np.ones_like(freqs) if amps is None else np.array(np.broadcast_to(np.asarray(amps, np.float64), freqs.shape))
        phis = np.zeros_like(freqs) if phis is None else np.array(np.broadcast_to(np.asarray(phis, np.float64), freqs.shape))
        self.dw = torch.from_numpy(np.ascontiguousarray(dw)).to(device)
        self.amp = torch.from_numpy(np.ascontiguousarray(amps)).to(device)
        self.phi = torch.from_numpy(np.ascontiguousarray(phis)).to(device)
        self.w = torch.zeros_like(self.dw)

    def read(self, n, out=None, sequential=False):
        """Next n samples of every channel.  sequential=True walks each channel with one lane;
        the default jumps to every chunk's exact starting angle and fills the chunks in parallel
        -- same bits."""
        import torch
        if out is None:
            out = torch.empty((self.nchan, n), dtype=torch.float32, device=self.w.device)
        assert out.shape == (self.nchan, n) and out.stride(1) == 1
        _check(lib().dspmi_tonegen_f32(out.data_ptr(), out.stride(0) if self.nchan > 1 else n, n, self.nchan, self.ntone,
                                       self.w.data_ptr(), self.dw.data_ptr(), self.amp.data_ptr(), self.phi.data_ptr(),
                                       int(self.use_sin), 1 if sequential else 0, _stream()), "tonegen")
        return out

    def read_complex(self, n, sequential=False):
        """Next n samples of every channel as complex64 with zero imaginary part (tonegen_read_complex)."""
        import torch
        out = torch.empty((self.nchan, n), dtype=torch.complex64, device=self.w.device)
        _check(lib().dspmi_tonegen_c32(out.data_ptr(), n, n, self.nchan, self.ntone,
                                       self.w.data_ptr(), self.dw.data_ptr(), self.amp.data_ptr(), self.phi.data_ptr(),
                                       int(self.use_sin), 1 if sequential else 0, _stream()), "tonegen_c32")
        return out

    def read_s16x2(self, n, sequential=False):
        """Next n samples of every channel as stereo int16 frames [nchan, n, 2] (both channels equal)."""
        import torch
        out = torch.empty((self.nchan, n, 2), dtype=torch.int16, device=self.w.device)
        _check(lib().dspmi_tonegen_s16x2(out.data_ptr(), n, n, self.nchan, self.ntone,
                                         self.w.data_ptr(), self.dw.data_ptr(), self.amp.data_ptr(), self.phi.data_ptr(),
                                         int(self.use_sin), 1 if sequential else 0, _stream()), "tonegen_s16x2")
        return out


class OlsPlan:
    """Overlap-save fast convolution for long filters (f32); same result as FirPlan.exec(x, nx)."""

    def __init__(self, taps, fft_n=0, device=-1):
        self.nh = len(taps)
        self._h = C.c_void_p()
        t = np.ascontiguousarray(taps, dtype=np.float32)
        _check(lib().dspmi_ols_create_f32(C.byref(self._h), t.ctypes.data_as(_fp), self.nh, fft_n, device), "ols_create")

    @property
    def fft_length(self):
        return lib().dspmi_ols_fft_length(self._h)

    def exec(self, x, y=None, hist=None):
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        nchan, nx = x.shape
        if y is None:
            y = torch.empty((nchan, nx), dtype=torch.float32, device=x.device)
        assert y.is_cuda and y.dtype == torch.float32 and y.shape == (nchan, nx) and y.stride(1) == 1
        hp, hs = None, 0
        if hist is not None:
            assert hist.is_cuda and hist.dtype == torch.float32 and hist.shape == (nchan, self.nh - 1) and hist.stride(1) == 1
            hp, hs = hist.data_ptr(), hist.stride(0) if nchan > 1 else self.nh - 1
        _check(lib().dspmi_ols_exec_f32(self._h, y.data_ptr(), y.stride(0) if nchan > 1 else nx,
                                        x.data_ptr(), x.stride(0) if nchan > 1 else nx, nx, nchan, hp, hs, _stream()),
               "ols_exec")
        return y

    def close(self):
        if self._h:
            lib().dspmi_ols_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# host drop-in routines (numpy float64 in / out), named as in the reference
# ---------------------------------------------------------------------------

def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def conv0(x, h):
    """convolution/main.c:15-33 -- y[nx]."""
    x, xp = _d(x); h, hp = _d(h)
    y = np.empty(len(x))
    lib().dsp_conv0(xp, len(x), hp, len(h), y.ctypes.data_as(_dp))
    return y


def conv1(x, h):
    """convolution/main.c:36-53 -- y[nx], bug-compatible."""
    x, xp = _d(x); h, hp = _d(h)
    y = np.empty(len(x))
    lib().dsp_conv1(xp, len(x), hp, len(h), y.ctypes.data_as(_dp))
    return y


def convolve(x, h, nxx):
    """fir/src/main.c:258-280 -- caller chooses the output length nxx."""
    x, xp = _d(x); h, hp = _d(h)
    xx = np.empty(nxx)
    lib().dsp_convolve(xx.ctypes.data_as(_dp), nxx, xp, len(x), hp, len(h))
    return xx


def _take(ptr, n):
    """Copy a malloc'd double buffer into numpy and free() it, as the reference's caller does."""
    out = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.cast(ptr, C.c_void_p))
    return out


def convolve_alloc(x, h):
    """fft/src/main.c:180-205 -- library-allocated result of nx+nh doubles."""
    x, xp = _d(x); h, hp = _d(h)
    ptr, n = _dp(), _u()
    lib().dsp_convolve_alloc(C.byref(ptr), C.byref(n), xp, len(x), hp, len(h))
    return _take(ptr, n.value)


def fir_set_taps(h):
    h, hp = _d(h)
    lib().dsp_fir_set_taps(hp, len(h))


def fir(x):
    """fft/src/main.c:207-244 -- convolve with the bound tap table."""
    x, xp = _d(x)
    ptr, n = _dp(), _u()
    lib().dsp_fir(C.byref(ptr), C.byref(n), xp, len(x))
    return _take(ptr, n.value)


def _c(z):
    z = np.ascontiguousarray(z, dtype=np.complex128)
    return z, z.view(np.float64).ctypes.data_as(_dp)


def dft(z):
    """fftc/main.c:31-36."""
    z, zp = _c(z)
    out = np.empty(len(z), dtype=np.complex128)
    lib().dsp_dft(zp, len(z), out.view(np.float64).ctypes.data_as(_dp))
    return out


def idft(z):
    """fftc/main.c:38-43 (unnormalised)."""
    z, zp = _c(z)
    out = np.empty(len(z), dtype=np.complex128)
    lib().dsp_idft(zp, len(z), out.view(np.float64).ctypes.data_as(_dp))
    return out


def fft_r2c(x):
    """fft/src/main.c:146-168 -- bins 0..n/2."""
    x, xp = _d(x)
    out = np.empty(len(x) // 2 + 1, dtype=np.complex128)
    lib().dsp_fft_r2c(out.view(np.float64).ctypes.data_as(_dp), xp, len(x))
    return out


def fft_mag(z):
    """fir/src/main.c:229-255 -- 2|X|/nx over nx/2 bins."""
    z, zp = _c(z)
    out = np.empty(len(z) // 2)
    lib().dsp_fft_mag(zp, len(z), out.ctypes.data_as(_dp))
    return out


def _readout(fn, z):
    z, zp = _c(z)
    out = np.empty(len(z) // 2)
    fn(zp, len(z), out.ctypes.data_as(_dp))
    return out


def ps_percent(z):
    """fftc/main.c:46-77 -- (re - im) over nx/2 bins, divided by their sum when it reaches 1e-4."""
    return _readout(lib().dsp_ps_percent, z)


def ps_abs(z):
    """fftc/main.c:80-105 -- 2|X|/nx over nx/2 bins of a transform."""
    return _readout(lib().dsp_ps_abs, z)


def compute_phi(z):
    """fftc/main.c:108-123 -- atan(im/re) where |re| >= 1e-5, else 0."""
    return _readout(lib().dsp_compute_phi, z)


def filter(x, coeffs):  # noqa: A001 - the reference's name
    """util/filter/filter.c:73-105."""
    x, xp = _d(x); coeffs, cp = _d(coeffs)
    assert len(coeffs) == len(x) // 2 + 1
    xx = np.empty(len(x))
    lib().dsp_filter(xx.ctypes.data_as(_dp), xp, len(x), cp)
    return xx


def make_blackman_window(nx):
    """fir/src/main.c:138-146."""
    w = np.empty(nx)
    lib().dsp_make_blackman_window(w.ctypes.data_as(_dp), nx)
    return w


def make_filter_kernel(fresp):
    """fir/src/main.c:149-226 -- complex kernel of 2*len(fresp) values (at most 32 non-zero)."""
    f, fp = _d(fresp)
    k = np.empty(2 * len(f), dtype=np.complex128)
    lib().dsp_make_filter_kernel(fp, len(f), k.view(np.float64).ctypes.data_as(_dp))
    return k
