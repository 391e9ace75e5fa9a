"""The N > 1 path: world_size-2 gloo processes shard channels (no collective) and split one long
signal into halo'd segments that one all_gather reassembles.  On CPU the per-rank compute is the
oracle (index arithmetic of the split only); the GPU test runs the same split over the
overlap-save and direct kernels, two ranks sharing cuda:0, and checks the gathered signal
against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dsp_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nx, nh, out_dir):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)          # every rank builds the same global input
        x = rng.uniform(-1, 1, nx).astype(np.float32)
        h = oracle.windowed_sinc_taps(nh).astype(np.float32)

        # --- one long signal: segment + halo, local filter, one all_gather ----------------
        lo, hi, in_lo = parallel.segment_plan(nx, nh, world)[rank]
        hist, body = parallel.split_history(torch.from_numpy(x[in_lo:hi]), lo, in_lo, nh)
        y_local = oracle.fir_stream_f32(body.numpy(), hist.numpy(), h)
        y = parallel.gather_segments(torch.from_numpy(y_local), nx, world)
        np.save(os.path.join(out_dir, "long_%d.npy" % rank), y.numpy())

        # --- independent channels: contiguous block per rank, no collective ------------------
        nchan = 5
        xc = rng.uniform(-1, 1, (nchan, 300)).astype(np.float32)
        c_lo, c_hi = parallel.shard_range(nchan, world, rank)
        yc = oracle.fir_batch_f32(xc[c_lo:c_hi], h[:31], 300) if c_hi > c_lo else np.zeros((0, 300))
        np.save(os.path.join(out_dir, "chan_%d.npy" % rank), yc)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 8, 4096, 1000003):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_segment_plan_cuts_equal_pieces():
    for nx in (1, 9, 10, 4096, 5000, 1000003):
        for world in (1, 2, 3, 4, 8):
            for nh in (1, 16, 8191):
                plan = parallel.segment_plan(nx, nh, world)
                seg = parallel.segment_length(nx, world)
                assert plan[0][0] == 0 and plan[-1][1] == nx
                assert all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
                assert all(hi - lo == seg for lo, hi, _ in plan[:-1] if hi < nx)       # equal pieces up to the last
                assert all(0 <= hi - lo <= seg for lo, hi, _ in plan)
                assert all(in_lo == max(0, lo - (nh - 1)) for lo, _, in_lo in plan)


@pytest.mark.parametrize("world,nx", [(2, 5001), (3, 5000)])
def test_gloo_ranks_segments_and_channels(tmp_path, world, nx):
    import oracle
    nh = 257
    mp.spawn(_worker, args=(world, _free_port(), nx, nh, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    x = rng.uniform(-1, 1, nx).astype(np.float32)
    h = oracle.windowed_sinc_taps(nh).astype(np.float32)
    want = oracle.fir_batch_f32(x, h, nx)[0]
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "long_%d.npy" % r))
        assert got.shape == (nx,)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)   # both ranks hold the full output
    xc = rng.uniform(-1, 1, (5, 300)).astype(np.float32)
    want_c = oracle.fir_batch_f32(xc, h[:31], 300)
    got_c = np.concatenate([np.load(os.path.join(str(tmp_path), "chan_%d.npy" % r)) for r in range(world)])
    np.testing.assert_allclose(got_c, want_c, rtol=0, atol=1e-12)


def _gpu_worker(rank, world, port, nx, nh, out_dir):
    import dsp_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)                       # both ranks share the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(77)
        x = rng.uniform(-1, 1, nx).astype(np.float32)
        k = np.arange(nh) - (nh - 1) / 2.0
        h = (np.sinc(0.25 * k) * np.hamming(nh)).astype(np.float32)
        h /= h.sum()
        lo, hi, in_lo = parallel.segment_plan(nx, nh, world)[rank]
        mine = torch.from_numpy(x[in_lo:hi]).cuda()
        for name, plan in (("ols", dsp_amd.OlsPlan(h)), ("fir", dsp_amd.FirPlan(h, "f32"))):
            y_local = parallel.filter_segment(plan, mine, lo, in_lo)
            assert y_local.is_cuda and y_local.shape == (hi - lo,)
            full = parallel.gather_segments(y_local, nx, world)
            assert full.is_cuda
            np.save(os.path.join(out_dir, "%s_%d.npy" % (name, rank)), full.cpu().numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nh", [(200001, 8191), (70000, 127)])
def test_split_signal_over_the_device_kernels_two_ranks_one_gpu(tmp_path, nx, nh):
    """BASELINE config 4's multi-GPU shape on one GPU: each rank filters its output segment from a
    slice that carries its own halo (no neighbour exchange), one all_gather reassembles the signal
    (fft/src/main.c:193-204 is the sum every output must equal)."""
    import oracle
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), nx, nh, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, nx).astype(np.float32)
    k = np.arange(nh) - (nh - 1) / 2.0
    h = (np.sinc(0.25 * k) * np.hamming(nh)).astype(np.float32)
    h /= h.sum()
    want = oracle.fir_batch_f32(x, h, nx)[0]
    for name in ("ols", "fir"):
        for r in range(world):
            got = np.load(os.path.join(str(tmp_path), "%s_%d.npy" % (name, r)))
            assert got.shape == (nx,)
            for lo in range(max(4096, (nh + 4095) // 4096 * 4096), nx - 4096, 4096):   # steady-state blocks
                w = want[lo:lo + 4096]
                assert np.linalg.norm(got[lo:lo + 4096] - w) <= 1e-6 * np.linalg.norm(w), (name, r, lo)
            assert np.linalg.norm(got - want) <= 1e-6 * np.linalg.norm(want), (name, r)
