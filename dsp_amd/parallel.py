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


def segment_length(nx, world):
    """Outputs per rank when one signal of nx samples is cut for `world` ranks: every rank the same
    count (the last one whatever remains), so that the gather is one all_gather of equal pieces
    straight into the output buffer."""
    return -(-nx // world)


def segment_plan(nx, nh, world):
    """Split the outputs [0, nx) of one long signal into `world` segments of segment_length(nx, world)
    samples (the last one shorter, possibly empty).

    Returns a list of (out_lo, out_hi, in_lo): rank r computes outputs [out_lo, out_hi) and needs
    inputs [in_lo, out_hi) with in_lo = max(0, out_lo - (nh - 1)); samples before index 0 are the
    zero history of the reference's convolve (fft/src/main.c:199)."""
    seg = segment_length(nx, world)
    plan = []
    for r in range(world):
        lo, hi = min(r * seg, nx), min((r + 1) * seg, nx)
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
    if body.shape[-1] == 0:
        return body.clone()
    return plan.exec(body[None, :].contiguous(), hist=hist[None, :].contiguous())[0]


def gather_segments(y_local, nx, world=None, group=None):
    """The per-rank output segments (1-D, cut by segment_plan) put together on every rank: ONE
    all_gather of equal pieces, received directly in the output buffer -- no staging copies, no
    concatenation.  Returns a view of the first nx samples of that buffer.  Only the last rank,
    whose segment may be shorter, sends a padded copy of its piece.  RCCL over xGMI under the nccl
    backend; never a reduction.  (gloo, used to rehearse the multi-rank path on fewer GPUs than
    ranks, has no collectives on device tensors: the pieces go through the host there.)"""
    if world is None:
        world = dist.get_world_size(group)
    if world == 1:
        return y_local
    assert y_local.dim() == 1
    seg = segment_length(nx, world)
    send = y_local.contiguous()
    if send.shape[0] != seg:
        send = torch.zeros((seg,), dtype=y_local.dtype, device=y_local.device)
        send[:y_local.shape[0]] = y_local
    if y_local.is_cuda and dist.get_backend(group) == "gloo":
        full = torch.empty((world * seg,), dtype=y_local.dtype)
        dist.all_gather(list(full.view(world, seg).unbind(0)), send.cpu(), group=group)
        return full[:nx].to(y_local.device)
    full = torch.empty((world * seg,), dtype=y_local.dtype, device=y_local.device)
    if y_local.is_cuda:
        dist.all_gather_into_tensor(full, send, group=group)
    else:
        dist.all_gather(list(full.view(world, seg).unbind(0)), send, group=group)
    return full[:nx]
