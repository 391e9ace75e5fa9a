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


def filter_segment(plan, x_with_halo, out_lo, in_lo):
    """One rank's share of a long signal through a FIR / overlap-save plan on its GPU.

    x_with_halo: 1-D CUDA tensor x[in_lo:out_hi] (the rank's outputs' inputs plus the nh-1 samples
    before them, as segment_plan lays them out); plan: FirPlan or OlsPlan with .nh and
    .exec(x[nchan, nx], hist=).  Returns y[out_lo:out_hi]."""
    hist, body = split_history(x_with_halo, out_lo, in_lo, plan.nh)
    return plan.exec(body[None, :].contiguous(), hist=hist[None, :].contiguous())[0]


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
    # gloo (CPU rehearsals of the multi-rank path, ranks sharing one GPU) has no all_gather of
    # device tensors: stage through the host there; RCCL gathers device to device over xGMI
    staged = pad.is_cuda and dist.get_backend(group) == "gloo"
    send = pad.cpu() if staged else pad
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_range(nx, world, r)
        out.append(p[..., :hi - lo])
    full = torch.cat(out, dim=-1)
    return full.to(y_local.device) if staged else full
