"""The N > 1 path on CPU: world_size-2 gloo processes shard channels (no collective) and split
one long signal into halo'd segments that one all_gather reassembles.  The per-rank compute is
the oracle here (no GPU in this container); on GPUs the same functions wrap FirPlan / OlsPlan."""
import os
import socket

import numpy as np
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


def test_two_rank_gloo_segments_and_channels(tmp_path):
    import oracle
    world, nx, nh = 2, 5001, 257
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
