"""Builds dsp_amd/libdspmi.so in-tree with hipcc for gfx950.

The library is plain HIP + a C ABI (no torch, no pybind): it is what a C host
links against, and what the Python harness loads through ctypes.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libdspmi.so")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]

# source -> extra flags.  fir_direct keeps products and sums as written (its
# fused multiply-adds are explicit), which the bit-exact double path relies on;
# MFMA accumulators stay in ordinary VGPRs (gfx950 has one unified file), which
# keeps hipcc from shuffling them through AGPR copies inside the tap loop (fft_stockham holds the
# FIR + spectrum kernel, hence the same flag there).
# fft_stockham is built without the SLP vectoriser: a packed f32 instruction issues at half the
# rate of a plain one on gfx950 (no throughput to gain), and pairing the butterflies' scalars
# costs a register move for every third instruction and 30-40 more VGPRs.
SOURCES = {
    "runtime.cpp": ["-x", "hip"],
    "fir_direct.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"],
    "fft_stockham.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
    "fft_plan.hip": [],
    "ols.hip": [],
    "spectrum.hip": [],
    "wrappers.cpp": ["-x", "hip"],
    "fftw_shim.cpp": ["-x", "hip"],
}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _compile(job):
    src, obj, flags = job
    cmd = [_hipcc()] + COMMON + flags + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r.stderr


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(HERE, "..", "include")   # the whole public contract: dspmi.h, the FFTW subset, the name map
    headers += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    jobs, objs = [], []
    for name, flags in SOURCES.items():
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj, flags))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(_compile, jobs):
                if verbose and warn:
                    sys.stderr.write(warn)
    if jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
