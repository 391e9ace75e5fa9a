"""One-process-per-GPU sharding of the hot path (torch.distributed; RCCL on GPUs, gloo on CPUs).

Channels (FIR) and frames (FFT) are independent, so a rank simply owns a contiguous block of
them and the data path has no collective.  The one exchange step in the whole design is the
gather of per-rank output segments when a single long signal is split for overlap-save: each
rank receives its output range plus the nh-1 input samples before it (the halo is supplied
with the scatter, so ranks never talk to their neighbours) and one all_gather puts the
segments back together.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous block [lo, hi) of `total` independent units owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def segment_plan(nx, nh, world):
    """Split the outputs [0, nx) of one long signal into `world` segments.

    Returns a list of (out_lo, out_hi, in_lo): rank r computes outputs [out_lo, out_hi) and needs
    inputs [in_lo, out_hi) with in_lo = max(0, out_lo - (nh - 1)); samples before index 0 are the
    zero history of the reference's convolve (fft/src/main.c:199)."""
    plan = []
    for r in range(world):
        lo, hi = shard_range(nx, world, r)
        plan.append((lo, hi, max(0, lo - (nh - 1))))
    return plan


def split_history(x_with_halo, out_lo, in_lo, nh):
    """From a rank's slice x[in_lo:out_hi] build (hist [nh-1], body [out_hi-out_lo]) for an
    exec(..., hist=) call; the part of the history before sample 0 is zero."""
    have = out_lo - in_lo
    body = x_with_halo[..., have:]
    hist = torch.zeros(x_with_halo.shape[:-1] + (nh - 1,), dtype=x_with_halo.dtype, device=x_with_halo.device)
    if have:
        hist[..., nh - 1 - have:] = x_with_halo[..., :have]
    return hist, body


def gather_segments(y_local, nx, world=None, group=None):
    """all_gather of the per-rank output segments into the full-length signal on every rank.

    Segments differ in length by at most one sample, so they are padded to the longest for the
    collective and trimmed afterwards.  One RCCL all_gather over xGMI; never a reduction."""
    if world is None:
        world = dist.get_world_size(group)
    if world == 1:
        return y_local
    longest = -(-nx // world)
    pad = torch.zeros(y_local.shape[:-1] + (longest,), dtype=y_local.dtype, device=y_local.device)
    pad[..., :y_local.shape[-1]] = y_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_range(nx, world, r)
        out.append(p[..., :hi - lo])
    return torch.cat(out, dim=-1)
