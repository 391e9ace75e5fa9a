"""The entry point that allocates nothing (dspmi_fft_exec_c2c_noalloc) against dspmi_fft_exec_c2c,
through the C ABI: two calls in a row on one GPU, the second with a smaller input, so stale
contents of the shared static scratch would show."""
import numpy as np
import pytest
import torch

import dsp_amd
import oracle

pytestmark = pytest.mark.gpu


def _frames(batch, n, dtype, seed):
    rng = np.random.default_rng(seed)
    z = rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))
    return torch.from_numpy(z.astype(dtype)).cuda()


@pytest.mark.parametrize("n,dtype", [(480, "f64"), (482, "f64"), (1000, "f64"), (100, "f64"), (480, "f32"), (1024, "f64"), (32768, "f64"),
                                     (65536, "f32"), (1 << 20, "f64"), (1 << 21, "f32")])
def test_no_alloc_c2c_matches_the_allocating_entry_point(n, dtype):
    ct = np.complex128 if dtype == "f64" else np.complex64
    plan = dsp_amd.FftPlan(n, dtype)
    big = n >= 1 << 20   # two and three column passes, one work frame that fills the static array
    for call, (batch, sign) in enumerate([(7 if big else 37, dsp_amd.FORWARD), (2 if big else 5, dsp_amd.BACKWARD)]):
        z = _frames(batch, n, ct, 100 * n + call)
        want = plan.c2c(z, sign)
        got = torch.full_like(z, float("nan"))
        plan.c2c_noalloc(z, sign, out=got)
        torch.cuda.synchronize()
        # the same kernels on the same inputs: bit for bit
        assert torch.equal(torch.view_as_real(got), torch.view_as_real(want)), (n, dtype, call)
        # and both against the oracle's DFT
        ref = oracle.dft(z[0].cpu().numpy().astype(np.complex128), sign, "radix2" if n & (n - 1) == 0 else "auto")
        err = np.linalg.norm(got[0].cpu().numpy() - ref) / np.linalg.norm(ref)
        assert err <= (1e-13 if dtype == "f64" else 1e-5), (n, dtype, call, err)


def test_no_alloc_c2c_in_place_and_limits():
    n = 100   # double precision, M = 256: the five-kernel any-length transform with its work frames
    plan = dsp_amd.FftPlan(n, "f64")   # (from M = 512 to 16384 one kernel does it all and needs none)
    assert "via" in plan.variant
    z = _frames(9, n, np.complex128, 7)
    want = plan.c2c(z)
    buf = z.clone()
    plan.c2c_noalloc(buf, out=buf)
    assert torch.equal(torch.view_as_real(buf), torch.view_as_real(want))
    # 100 points ride 256-point work frames of 4 KiB: the 16 MiB array holds 4096 of them
    L = dsp_amd.lib()
    big = torch.empty((4097, n), dtype=torch.complex128, device="cuda")
    rc = L.dspmi_fft_exec_c2c_noalloc(plan._h, big.data_ptr(), big.data_ptr(), 4097, -1, None)
    assert rc == -22 and b"static" in L.dspmi_last_error()
    ok = torch.empty((4096, n), dtype=torch.complex128, device="cuda").normal_()
    assert L.dspmi_fft_exec_c2c_noalloc(plan._h, ok.data_ptr(), ok.data_ptr(), 4096, -1, None) == 0
    torch.cuda.synchronize()
    # a smaller call right after the largest one
    small = z[:2].clone()
    plan.c2c_noalloc(small, out=small)
    assert torch.equal(torch.view_as_real(small), torch.view_as_real(want[:2]))


@pytest.mark.parametrize("n,dtype,batch", [(100, "f64", 40), (65536, "f32", 6)])
def test_no_alloc_c2c_on_a_fresh_plan_leaves_device_memory_alone(n, dtype, batch):
    """The first execute of a plan, before any dspmi_fft_exec_c2c has grown the plan's own work
    buffers: the amount of free device memory is the same before and after (an entry point that fell
    back to the plan's buffer would have had to allocate it), and the result is that of a second
    plan's allocating execute, bit for bit."""
    ct = np.complex128 if dtype == "f64" else np.complex64
    z = _frames(batch, n, ct, n + batch)
    got = torch.empty_like(z)
    fresh = dsp_amd.FftPlan(n, dtype)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    fresh.c2c_noalloc(z, dsp_amd.FORWARD, out=got)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, (free_before, free_after)
    want = dsp_amd.FftPlan(n, dtype).c2c(z, dsp_amd.FORWARD)
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(want))


def test_no_alloc_c2c_refuses_frames_longer_than_the_static_array():
    plan = dsp_amd.FftPlan(1 << 22, "f32")      # 32 MiB per work frame
    z = torch.zeros((1, 1 << 22), dtype=torch.complex64, device="cuda")
    rc = dsp_amd.lib().dspmi_fft_exec_c2c_noalloc(plan._h, z.data_ptr(), z.data_ptr(), 1, -1, None)
    assert rc == -22 and b"static" in dsp_amd.lib().dspmi_last_error()
