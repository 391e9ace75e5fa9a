"""The C-ABI library loads without a GPU and exports what include/dspmi.h declares."""
import ctypes
import os
import re

import pytest

import dsp_amd
import importlib

L = importlib.import_module('dsp_amd.lib')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dspmi?_[a-z0-9_]+|dsp_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = dsp_amd.lib()
    names = _declared("dspmi.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    # and the Python binding table covers exactly the header
    assert sorted(L.exported_symbols()) == names


def test_abi_version_and_error_channel():
    lib = dsp_amd.lib()
    assert lib.dspmi_abi_version() == 1
    h = ctypes.c_void_p()
    # argument errors are reported without touching a device
    rc = lib.dspmi_fir_create_f32(ctypes.byref(h), None, 0, -1)
    assert rc == -22 and b"fir_create" in lib.dspmi_last_error()
    rc = lib.dspmi_fft_create(ctypes.byref(h), 0, 0, -1)
    assert rc == -22
    rc = lib.dspmi_fft_create(ctypes.byref(h), 1000, 7, -1)
    assert rc == -22


def test_no_oracle_or_cpu_path_in_the_product():
    """The shipped package must not import the checker."""
    for root, _, files in os.walk(os.path.join(ROOT, "dsp_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt and "dsp_oracle" not in txt, f


def test_fftw_subset_symbols_exported():
    lib = dsp_amd.lib()
    names = [n for n in _declared("dspmi_fftw3.h") if n.startswith("dspmi_fftw_")]
    assert sorted(names) == sorted(["dspmi_fftw_malloc", "dspmi_fftw_free", "dspmi_fftw_plan_dft_1d",
                                    "dspmi_fftw_plan_dft_r2c_1d", "dspmi_fftw_plan_dft_c2r_1d",
                                    "dspmi_fftw_execute", "dspmi_fftw_destroy_plan"])
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    # host-only entry points work without a device
    lib.dspmi_fftw_malloc.restype = ctypes.c_void_p
    lib.dspmi_fftw_malloc.argtypes = [ctypes.c_size_t]
    lib.dspmi_fftw_free.argtypes = [ctypes.c_void_p]
    p = lib.dspmi_fftw_malloc(1024)
    assert p and p % 64 == 0
    lib.dspmi_fftw_free(p)


@pytest.mark.gpu
def test_library_and_torch_share_one_hip_runtime_in_either_import_order():
    """torch ships its own HIP runtime; a second copy in the process finds no device.  The
    binding loads the library after torch whatever the caller's import order."""
    import subprocess
    import sys
    code = ("import dsp_amd\nL = dsp_amd.lib()\nn = L.dspmi_device_count()\nimport torch\n"
            "assert torch.cuda.is_available() and n >= 1 and L.dspmi_device_count() == n\n"
            "x = torch.ones(1, 64, device='cuda')\n"
            "y = dsp_amd.FirPlan([1.0, 1.0]).exec(x)\nassert float(y.sum()) == 127.0\nprint('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-800:]
