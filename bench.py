#!/usr/bin/env python3
"""Benchmark of the FIR / FFT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload fft1024|fir127|fir16|ols8191|fir8191|chain|ols8191_split|c1|fft_len --fft-n N --fft-dtype f32|f64]

One "step" is one pass of the hot path over one batch of synthetic input that
is already resident in HBM.  The default workload is BASELINE.json configs[1]:
batched complex f32 FFT, N = 1024, 2^20 frames per GPU (8 GiB in, 8 GiB out).
With N > 1 ranks (one process per GPU, launched by torch.distributed.run) every
rank transforms its own 2^20 frames: frames are independent, so the data path
has no collective ("scaling": "weak"); the only communication is the timing
barrier and the max-over-ranks reduction.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel against the HBM (or FP32 vector) roof, from
                 HIP events recorded on the launch stream around the timed region
  cpu_baseline : the oracle's CPU port timed on a bounded sample of the same
                 workload on this box's host cores (rank 0, N = 1 only)
  secondary    : short measurements of the other BASELINE configs, each with its own roofline
                 block: the two FIR regimes (16 and 127 taps), the 8191-tap signal by
                 overlap-save and by the direct kernel, and the tonegen -> FIR -> FFT chain
  pcie_inclusive : the headline workload with host (pinned) buffers in and out, on a bounded sample

--workload ols8191_split is the one workload with a data-path collective: one 2^26-sample signal
is cut into one output segment per rank, each rank filters its segment (its input slice carries
the nh-1 samples of halo, so ranks never exchange neighbours' samples) and one all_gather puts
the output together on every rank ("scaling": "strong").
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# RCCL between processes needs dmabuf IPC on these hosts (legacy IPC fails with
# hipIpcGetMemHandle: invalid argument); keep the setting if the launcher did not export it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

import dsp_amd

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec
FP32_PEAK_TFLOPS = 157.3   # MI355X FP32 vector spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default: 200 launches of ~3 ms, so the timed region (0.6 s) is long enough for the clocks
    # to settle inside it and for a sampling monitor to see the GPU busy
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fft1024", choices=["fft1024", "fir127", "fir127_direct", "fir16", "c1", "fir8191", "ols8191",
                                                             "ols8191_split", "chain", "fft_len"])
    ap.add_argument("--fft-n", type=int, default=16384, help="fft_len workload: points per frame (1 GiB of frames per GPU)")
    ap.add_argument("--fft-dtype", default="f32", choices=["f32", "f64"], help="fft_len workload: precision")
    ap.add_argument("--ols-n", type=int, default=0, help="8191-tap overlap-save: transform length (0 = the plan's choice)")
    ap.add_argument("--chain-two-kernels", action="store_true", help="chain workload: FIR and real transform as two kernels (A/B)")
    ap.add_argument("--chain-separate-generator", action="store_true",
                    help="chain workload: the generator as a kernel of its own in front of the fused filter + spectrum kernel (A/B)")
    ap.add_argument("--frames", type=int, default=1 << 20, help="FFT frames per GPU")
    ap.add_argument("--channels", type=int, default=512, help="FIR channels per GPU (4096 / 8 GPUs)")
    ap.add_argument("--samples", type=int, default=1 << 20, help="FIR samples per channel")
    ap.add_argument("--long-samples", type=int, default=1 << 26, help="samples of the single long signal (8191-tap workloads)")
    ap.add_argument("--chain-taps", type=int, default=127, choices=[16, 127], help="chain workload: the 16-tap table of alsa/main.c or the 127-tap low-pass")
    ap.add_argument("--chunks", type=int, default=4, help="chain workload: chunks the 2^20 samples per channel are streamed in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    return ap.parse_args()


def setup_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DSPMI_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than
    # ranks (ranks then share devices; RCCL refuses that)
    backend = os.environ.get("DSPMI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    return world, rank, local


def barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def uniform(shape, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.rand(shape, device=device, generator=g, dtype=torch.float32) * 2 - 1


def timed(fn, steps, warmup, world):
    """Returns (wall seconds over `steps`, max over ranks; mean event ms per launch on this rank;
    median of the per-launch event times).  The events are recorded on torch's current stream,
    which is the stream every launch is given (dsp_amd.lib passes torch.cuda.current_stream())."""
    for _ in range(warmup):
        fn()
    barrier(world)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    barrier(world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    median = per[steps // 2] if steps % 2 else 0.5 * (per[steps // 2 - 1] + per[steps // 2])
    return dt, ev[0].elapsed_time(ev[steps]) / steps, median


# ---------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------

def make_fft(args, rank):
    n = 1024
    # Input and output are the two halves of one allocation: the same placement in every process
    # (the kernel visits the frames from two fronts so that its rate does not depend on the
    # distance between the buffers, tools/fft_offsets.py; this keeps the rest equal too).
    both = torch.empty((2, args.frames, n, 2), dtype=torch.float32, device="cuda")
    both[0].uniform_(-1.0, 1.0, generator=torch.Generator(device="cuda").manual_seed(0xD5B0 + 2 + rank))
    z = torch.view_as_complex(both[0])
    out = torch.view_as_complex(both[1])
    plan = dsp_amd.FftPlan(n, "f32")
    flop_per_frame = 5.0 * n * math.log2(n)     # nominal radix-2 count
    bytes_per_frame = 2.0 * n * 8               # read once + write once
    info = dict(kernel=plan.variant, units=args.frames, unit_name="frames", n=n,
                flop_per_unit=flop_per_frame, bytes_per_unit=bytes_per_frame, bound="hbm",
                samples_per_unit=n,
                workload="batched c2c f32 FFT, N=1024, %d frames per GPU, out of place (output buffer directly behind the input)" % args.frames)
    return (lambda: plan.c2c(z, dsp_amd.FORWARD, out=out)), info, (plan, z, out, both)


def make_fft_len(n, dtype):
    """Secondary lines: other transform lengths and double precision (the precision of the
    reference's own fftw calls, fftc/main.c:31-43), 1 GiB of frames in and 1 GiB out."""
    esz = 8 if dtype == "f32" else 16
    frames = (1 << 30) // (n * esz)
    both = torch.empty((2, frames, n, 2), dtype=torch.float32 if dtype == "f32" else torch.float64, device="cuda")
    both[0].uniform_(-1.0, 1.0, generator=torch.Generator(device="cuda").manual_seed(0xD5B0 + n))
    z, out = torch.view_as_complex(both[0]), torch.view_as_complex(both[1])
    plan = dsp_amd.FftPlan(n, dtype)
    info = dict(kernel=plan.variant, units=frames, unit_name="frames", n=n, flop_per_unit=5.0 * n * math.log2(n),
                bytes_per_unit=2.0 * n * esz, bound="hbm", samples_per_unit=n,
                workload="batched c2c %s FFT, N=%d, %d frames, out of place" % (dtype, n, frames))
    passes = plan.variant.split("_as_")[1].count("x") + 1 if "_as_" in plan.variant else 1
    if passes > 1:
        info["note"] = ("%d passes over memory (one work frame between them): `achieved` counts the algorithmic bytes of "
                        "one read and one write, each pass moves them once" % passes)
    return (lambda: plan.c2c(z, dsp_amd.FORWARD, out=out)), info, (plan, z, out, both)


def sinc_taps(nh):
    k = np.arange(nh, dtype=np.float64) - (nh - 1) / 2.0
    h = np.sinc(2.0 * 6000.0 / 48000.0 * k) * (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(nh) / (nh - 1)))
    return (h / h.sum()).astype(np.float32)


TAPS16 = np.array([  # alsa/main.c:389-407 of the reference, cast to f32
    -0.131625603115805, 0.103820148451119, 0.096541374482813, 0.102280926834003,
    0.113689768997426, 0.125850624234278, 0.135740574367069, 0.141133806318268,
    0.141133806318268, 0.135740574367069, 0.125850624234278, 0.113689768997426,
    0.102280926834003, 0.096541374482813, 0.103820148451119, -0.131625603115805], dtype=np.float32)


def make_fir(args, rank, nh, direct=False):
    """Batched FIR through the plan's own choice of kernel (overlap-save from 64 taps up on long
    rows, the direct MFMA kernel otherwise), or with the direct kernel pinned."""
    taps = TAPS16 if nh == 16 else sinc_taps(nh)
    # input and output from one allocation, as in make_fft: the same placement in every process
    both = torch.empty((2, args.channels, args.samples), dtype=torch.float32, device="cuda")
    both[0].uniform_(-1.0, 1.0, generator=torch.Generator(device="cuda").manual_seed(0xD5B0 + 3 + rank))
    x, y = both[0], both[1]
    if direct:
        os.environ["DSPMI_FIR_VARIANT"] = "mfma"
    try:
        plan = dsp_amd.FirPlan(taps, "f32")
    finally:
        if direct:
            del os.environ["DSPMI_FIR_VARIANT"]
    units = args.channels * args.samples
    # the f32 matrix kernel is graded against the f32 rate once 2*nh flop per 8 bytes pass the ridge;
    # the kernels that do not execute those f32 flops (16-bit matrix pipe, overlap-save) against HBM
    compute_bound = 2.0 * nh / 8.0 > FP32_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS and "mfma16x16x4" in plan.variant
    info = dict(kernel=plan.variant, units=units, unit_name="samples", nh=nh,
                flop_per_unit=2.0 * nh, bytes_per_unit=8.0, bound="fp32" if compute_bound else "hbm",
                samples_per_unit=1,
                workload="batched FIR, %d taps, %d channels x %d samples per GPU%s"
                         % (nh, args.channels, args.samples, ", direct kernel pinned" if direct else ""))
    if "f16x2" in plan.variant:
        info["note"] = ("f32 samples and taps enter the f16 matrix pipe as two parts each under a per-tile scale (3 MFMAs per "
                        "32 window positions, f32 accumulation); one pass over HBM, 8 B per sample")
    if "overlap_save" in plan.variant:
        n = int(plan.variant.rsplit("_n", 1)[1])
        info["note"] = ("the plan computes these outputs by overlap-save (FFT length %d, %d new outputs per block): "
                        "graded against HBM at the algorithmic 8 B per sample; the kernel itself moves %.2f B per sample"
                        % (n, n - nh + 1, 4.0 * n / (n - nh + 1) + 4.0))
    return (lambda: plan.exec(x, args.samples, y=y)), info, (plan, x, y, both)


def make_long(args, rank, use_ols):
    """BASELINE config 4: one long real signal, 8191 taps, direct MFMA kernel vs overlap-save."""
    nh = 8191
    taps = sinc_taps(nh)
    x = uniform((1, args.long_samples), 0xD5B0 + 4 + rank, "cuda")
    y = torch.empty_like(x)
    if use_ols:
        plan = dsp_amd.OlsPlan(taps, args.ols_n)
        kernel = "ols_fft%d" % plan.fft_length
        # A launch moves 2 x 256 MiB, the size of the last-level cache: launched again and again on
        # one pair of buffers it would find part of its input there.  The launches rotate over
        # several pairs instead (1 GiB of input, 1 GiB of output), so each reads from HBM.
        nbuf = 4
        xs = [x] + [uniform((1, args.long_samples), 0xD5B0 + 40 + i + 8 * rank, "cuda") for i in range(1, nbuf)]
        ys = [y] + [torch.empty_like(x) for _ in range(1, nbuf)]
        turn = [0]

        def fn():
            i = turn[0] = (turn[0] + 1) % nbuf
            plan.exec(xs[i], y=ys[i])
    else:
        os.environ["DSPMI_FIR_VARIANT"] = "mfma"   # pin the direct kernel (the plan alone would pick overlap-save)
        try:
            plan = dsp_amd.FirPlan(taps, "f32")
        finally:
            del os.environ["DSPMI_FIR_VARIANT"]
        kernel = plan.variant
        fn = lambda: plan.exec(x, args.long_samples, y=y)
    info = dict(kernel=kernel, units=args.long_samples, unit_name="samples", nh=nh,
                flop_per_unit=2.0 * nh, bytes_per_unit=8.0, bound="hbm" if use_ols else "fp32", samples_per_unit=1,
                workload="one %d-sample signal, %d taps, %s" % (args.long_samples, nh,
                                                                "overlap-save" if use_ols else "direct"))
    if use_ols:
        n = plan.fft_length
        info["note"] = ("one kernel, one pass over HBM: 8 B per sample algorithmic; blocks of %d samples overlap by %d, so the "
                        "kernel itself reads %.2f B and writes 4 B per output; launches rotate over %d input/output pairs"
                        % (n, nh - 1, 4.0 * n / (n - nh + 1), nbuf))
        return fn, info, (plan, xs, ys)
    return fn, info, (plan, x, y)


def make_long_split(args, rank, world):
    """BASELINE config 4 over several GPUs: the 2^26-sample signal's outputs are cut into one
    segment per rank; a rank's input slice already holds the nh-1 samples before its segment, so
    the only exchange is the all_gather of the outputs (RCCL over xGMI; gloo in rehearsals)."""
    from dsp_amd import parallel
    nh, nx = 8191, args.long_samples
    taps = sinc_taps(nh)
    lo, hi, in_lo = parallel.segment_plan(nx, nh, world)[rank]
    # every rank draws the same signal and keeps its own slice (synthetic stand-in for the scatter)
    g = torch.Generator(device="cuda").manual_seed(0xD5B0 + 4)
    whole = torch.rand((nx,), device="cuda", generator=g, dtype=torch.float32) * 2 - 1
    mine = whole[in_lo:hi].clone()
    del whole
    torch.cuda.empty_cache()
    plan = dsp_amd.OlsPlan(taps, args.ols_n)
    state = {}

    def local_only():
        state["y"] = parallel.filter_segment(plan, mine, lo, in_lo)

    def step():
        local_only()
        state["full"] = parallel.gather_segments(state["y"], nx, world)

    info = dict(kernel="ols_fft%d + all_gather" % plan.fft_length, units=nx / world, unit_name="samples", nh=nh,
                flop_per_unit=2.0 * nh, bytes_per_unit=8.0, bound="hbm", samples_per_unit=1, local_only=local_only,
                workload="one %d-sample signal, %d taps, overlap-save, split into %d halo'd segments, outputs all-gathered"
                         % (nx, nh, world))
    return step, info, (plan, mine, state)


def make_chain(args, rank):
    """BASELINE config 5: per-channel tones -> FIR low-pass -> N=4096 real FFT -> normalised
    magnitude spectrum, streamed in 4 chunks; the generator's phases and the FIR history are
    carried from chunk to chunk.  Every stage runs inside the timed step; the time of the
    FIR + FFT stages alone (tones pre-generated) is reported beside it."""
    nchan, nfft, nchunk = 1024, 4096, args.chunks
    chunk = (1 << 20) // nchunk
    nh = args.chain_taps
    taps = TAPS16 if nh == 16 else sinc_taps(nh)
    freqs = np.array([[100.0 * (c + 1 + 1024 * rank), 12000.0] for c in range(nchan)])
    gen = dsp_amd.ToneGen(48000.0, freqs, use_sin=True)
    fir = dsp_amd.FirPlan(taps, "f32")
    fft = dsp_amd.FftPlan(nfft, "f32")
    x = torch.empty((nchan, chunk), dtype=torch.float32, device="cuda")
    y = torch.empty((nchan, chunk), dtype=torch.float32, device="cuda")
    ps = torch.empty((nchan * chunk // nfft, nfft // 2 + 1), dtype=torch.float32, device="cuda")
    hist = torch.zeros((nchan, nh - 1), dtype=torch.float32, device="cuda")

    fused = nh <= 305 and not args.chain_two_kernels
    one_kernel = fused and not args.chain_separate_generator
    hists = [hist, torch.empty_like(hist)]
    work = None
    if one_kernel:
        work = torch.empty((dsp_amd.lib().dspmi_tone_fir_spectrum_work_bytes(chunk, nchan, gen.ntone),), dtype=torch.uint8, device="cuda")

    def filter_and_transform():
        if fused:
            fir.spectrum(fft, x, hist=hist, out=ps)      # FIR -> r2c -> |X| / sum in one kernel: y never reaches HBM
        else:
            fir.exec(x, chunk, y=y, hist=hist)
            fft.r2c_spectrum(y.view(-1, nfft), out=ps)   # r2c + |X| + normalise in one pass
        hist.copy_(x[:, chunk - (nh - 1):])

    def step():
        for _ in range(nchunk):
            if one_kernel:
                # generator -> FIR -> r2c -> |X| / sum in one kernel: neither x nor y reaches HBM; the
                # kernel leaves the chunk's last nh-1 samples for the next call
                fir.tone_spectrum(fft, gen, chunk, hist=hists[0], hist_out=hists[1], out=ps, work=work)
                hists.reverse()
            else:
                gen.read(chunk, out=x)
                filter_and_transform()

    gen.read(chunk, out=x)
    for _ in range(2):
        filter_and_transform()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nchunk * 3):
        filter_and_transform()
    torch.cuda.synchronize()
    fir_fft_ms = (time.perf_counter() - t0) * 1e3 / 3

    units = nchan * chunk * nchunk
    if one_kernel:
        kernel = "tone_fir_spectrum(tones + f16x2 matrix pipe, %d taps, n%d)" % (nh, nfft)
        how = ("one kernel: the samples are generated into the filter's staging registers and the filtered samples stay on "
               "chip, so the kernel itself moves only the 2 B of spectrum per sample (plus ~1.5 % of tables)")
    elif fused:
        kernel = "tonegen + fir_spectrum(f16x2 matrix pipe, %d taps, n%d)" % (nh, nfft)
        how = "the generator hands its samples on through HBM (4 B written + 4 B read per sample), the filtered samples stay on chip"
    else:
        kernel = "tonegen + %s + r2c_spectrum(%s)" % (fir.variant, fft.variant)
        how = "the stages hand their results on through HBM, which the roof does not allow for"
    info = dict(kernel=kernel, units=units, unit_name="samples",
                nh=nh, flop_per_unit=2.0 * nh + 2.5 * math.log2(nfft), bytes_per_unit=4.0 + 2.0, bound="hbm",
                samples_per_unit=1, fir_fft_ms=fir_fft_ms,
                note="algorithmic bytes: 4 B per generated sample + 2 B of spectrum per sample (N/2+1 floats per N samples); " + how,
                workload="chain: tones -> %d-tap FIR -> N=%d r2c -> |X|/sum, %d channels x %d samples in %d chunks"
                         % (nh, nfft, nchan, chunk * nchunk, nchunk))
    return step, info, (gen, x, fir, fft, y, ps, hists, work)


def flops_key(info):
    """`gflops` where the kernel executes the counted operations (the direct form's 2*nh per
    output in f32, or the transform's nominal 5*N*log2(N)); the f16 matrix-pipe kernel, the
    overlap-save kernels and the chain compute the same outputs another way, so their figure is
    what the direct form WOULD have executed."""
    k = info["kernel"]
    executes = k.startswith("fft_") or "mfma16x16x4" in k or "stream" in k
    return "gflops" if executes else "direct_form_equivalent_gflops"


def measured_traffic(info):
    """HBM bytes per launch of this workload from the committed rocprofv3 PMC passes
    (profiles/traffic.json, filled from separate --pmc FETCH_SIZE / WRITE_SIZE runs), or None
    when the workload was not profiled at this size."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    key = "%s | %s" % (info["kernel"], info["workload"])
    ent = tab.get(key)
    return ent["hbm_bytes"] if ent else None


TRAFFIC_SOURCE = ("profiles/traffic.json: a table filled from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                  "of this workload (not collected in this run)")


def roofline(info, ms_per_launch, ms_median=None):
    """`achieved` uses the mean event time per launch over the timed region; the median of the
    per-launch event times is reported beside it (`launch_ms_median`)."""
    sec = ms_per_launch * 1e-3
    traffic = measured_traffic(info)
    common = {"launch_ms": round(ms_per_launch, 4),
              "launch_ms_median": None if ms_median is None else round(ms_median, 4),
              "traffic": traffic, "traffic_source": TRAFFIC_SOURCE if traffic is not None else None}
    if info["bound"] == "hbm":
        achieved = info["bytes_per_unit"] * info["units"] / sec / 1e9
        out = {"bound": "hbm", "kernel": info["kernel"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), **common,
               "algorithmic_bytes_per_launch": info["bytes_per_unit"] * info["units"]}
        if info["kernel"].startswith("fir_f32_") and "mfma16x16x32" in info["kernel"] and info.get("nh", 0) * 2.0 / 8.0 > FP32_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS:
            # a direct-form kernel past the ridge of the f32 vector roof: that fraction too (it can exceed 1, the
            # products run on the 16-bit matrix pipe)
            out["fp32_vector_frac"] = round(info["flop_per_unit"] * info["units"] / sec / 1e12 / FP32_PEAK_TFLOPS, 4)
        if "note" in info:
            out["note"] = info["note"]
        return out
    achieved = info["flop_per_unit"] * info["units"] / sec / 1e12
    out = {"bound": "mfma", "dtype_peak": "f32 MFMA dense = FP32 vector peak", "kernel": info["kernel"], "achieved": round(achieved, 2), "peak": FP32_PEAK_TFLOPS,
           "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 4), **common,
           "hbm_frac": round(info["bytes_per_unit"] * info["units"] / sec / 1e9 / HBM_PEAK_GBS, 4),
           "algorithmic_flop_per_launch": info["flop_per_unit"] * info["units"]}
    if "note" in info:
        out["note"] = info["note"]
    return out


# ---------------------------------------------------------------------------
# CPU baseline (the oracle's port; rank 0, N = 1 only)
# ---------------------------------------------------------------------------

def cpu_baseline_fft(n=1024, target_s=12.0):
    """The reference's FFTs are FFTW3's, which is not installed, so the baseline is the
    oracle's double radix-2 (fft/doc/notes:8-22), one thread, twiddles from a table,
    built -O3 ("port").  Timed on a bounded number of frames."""
    import ctypes as C
    import oracle
    L = oracle.lib("_fast")
    L.oracle_set_threads.argtypes = [C.c_int]
    L.oracle_set_threads.restype = C.c_int
    L.oracle_set_threads(1)   # the library is an OpenMP build: pin the single-core figure to one thread
    rng = np.random.default_rng(0)
    probe = 2048
    z = rng.uniform(-1, 1, (probe, n, 2))
    out = np.empty_like(z)
    dp = C.POINTER(C.c_double)
    t0 = time.perf_counter()
    L.oracle_fft_batch(z.ctypes.data_as(dp), out.ctypes.data_as(dp), n, probe, -1, 1)
    per = (time.perf_counter() - t0) / probe
    frames = int(max(probe, min(1 << 20, target_s / per)))
    reps = max(1, frames // probe)
    t0 = time.perf_counter()
    for _ in range(reps):
        L.oracle_fft_batch(z.ctypes.data_as(dp), out.ctypes.data_as(dp), n, probe, -1, 1)
    dt = time.perf_counter() - t0
    done = reps * probe
    gflops = 5.0 * n * math.log2(n) * done / dt / 1e9
    # the same loop spread over the host cores this job may use (frames are independent)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthr = L.oracle_set_threads(max(1, min(16, avail)))
    big = np.tile(z, (8, 1, 1))
    bout = np.empty_like(big)
    L.oracle_fft_batch(big.ctypes.data_as(dp), bout.ctypes.data_as(dp), n, len(big), -1, 1)
    t0 = time.perf_counter()
    for _ in range(3):
        L.oracle_fft_batch(big.ctypes.data_as(dp), bout.ctypes.data_as(dp), n, len(big), -1, 1)
    mt = 3 * len(big) / (time.perf_counter() - t0)
    L.oracle_set_threads(1)
    return {"value": round(gflops, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "frames_per_s": round(done / dt, 1),
            "all_cores": {"value": round(5.0 * n * math.log2(n) * mt / 1e9, 3), "unit": "GFLOP/s", "cores": nthr,
                          "frames_per_s": round(mt, 1)},
            "sample": "%d frames of N=%d (double radix-2 with twiddle table, gcc -O3, 1 thread; "
                      "FFTW3, which the reference links, is not installed)" % (done, n)}


def cpu_baseline_fir(nh, target_s=8.0):
    """One thread on the host.  With the reference's own conv0 at hand (oracle/_ref, compiled from
    convolution/main.c where that tree was mounted) and a filter it was written for (tens of taps),
    that routine is timed ("reference"); otherwise `convolve` restated in the oracle, -O3 ("port")."""
    import ctypes as C
    import oracle
    rng = np.random.default_rng(0)
    nx = 1 << 16
    x = rng.uniform(-1, 1, nx)
    h = np.asarray(TAPS16 if nh == 16 else sinc_taps(nh), dtype=np.float64)
    y = np.empty(nx)
    dp = C.POINTER(C.c_double)
    ref = oracle.ref_lib() if nh <= 64 else None
    if ref is not None:
        kind, what = "reference", "the reference's conv0 (convolution/main.c:15-33, gcc -O2)"
        run = lambda: ref.ref_conv0(x.ctypes.data_as(dp), nx, h.ctypes.data_as(dp), nh, y.ctypes.data_as(dp))
    else:
        L = oracle.lib("_fast")
        kind, what = "port", "convolve restated (double, gcc -O3)"
        run = lambda: L.oracle_convolve(y.ctypes.data_as(dp), nx, x.ctypes.data_as(dp), nx, h.ctypes.data_as(dp), nh)
    t0 = time.perf_counter()
    run()
    per = time.perf_counter() - t0
    reps = int(max(1, min(4096, target_s / per)))
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    dt = time.perf_counter() - t0
    out = {"value": float("%.4g" % (reps * nx / dt / 1e9)), "unit": "Gsamples/s", "cores": 1, "kind": kind,
           "sample": "%d blocks of %d samples, %d taps, %s, 1 thread" % (reps, nx, nh, what)}
    # the same sums spread over the host cores this job may use (channels are independent):
    # the restated convolve, -O3, OpenMP over channels
    try:
        L = oracle.lib("_fast")
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_threads.restype = C.c_int
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        nthr = L.oracle_set_threads(max(1, min(16, avail)))
        nchan = 4 * nthr
        xs = rng.uniform(-1, 1, (nchan, nx))
        ys = np.empty_like(xs)
        call = lambda: L.oracle_convolve_batch(ys.ctypes.data_as(dp), nx, xs.ctypes.data_as(dp), nx, h.ctypes.data_as(dp), nh, nchan)
        t0 = time.perf_counter()
        call()
        per = time.perf_counter() - t0
        reps2 = int(max(1, min(256, 4.0 / per)))
        t0 = time.perf_counter()
        for _ in range(reps2):
            call()
        dt2 = time.perf_counter() - t0
        L.oracle_set_threads(1)
        out["all_cores"] = {"value": float("%.4g" % (reps2 * nchan * nx / dt2 / 1e9)), "unit": "Gsamples/s", "cores": nthr,
                            "kind": "port", "sample": "%d x %d channels of %d samples, convolve restated, gcc -O3, OpenMP" % (reps2, nchan, nx)}
    except (OSError, AttributeError) as e:   # a serial build of the checker: single-core figure only
        out["all_cores"] = None
        print("cpu_baseline: all-cores figure skipped (%s)" % e, file=sys.stderr)
    return out


def cpu_as_shipped(target_s=2.0):
    """The reference's FIR loops built the way the reference builds them: gcc -Wall, no -O
    (convolution/do_build.sh:2, fft/src/do_build.sh:5), one thread.  conv0 with the 16-tap table
    and convolve with 127 taps, restated in the oracle and compiled with those flags."""
    import ctypes as C
    import oracle
    L = oracle.lib("_O0")
    rng = np.random.default_rng(0)
    nx = 1 << 16
    x = rng.uniform(-1, 1, nx)
    y = np.empty(nx)
    dp = C.POINTER(C.c_double)
    out = {}
    for name, nh, call in (("conv0_16_taps", 16, "oracle_conv0"), ("convolve_127_taps", 127, "oracle_convolve")):
        h = np.asarray(TAPS16 if nh == 16 else sinc_taps(nh), dtype=np.float64)
        if call == "oracle_conv0":
            run = lambda: L.oracle_conv0(x.ctypes.data_as(dp), nx, h.ctypes.data_as(dp), nh, y.ctypes.data_as(dp))
        else:
            run = lambda: L.oracle_convolve(y.ctypes.data_as(dp), nx, x.ctypes.data_as(dp), nx, h.ctypes.data_as(dp), nh)
        t0 = time.perf_counter()
        run()
        per = time.perf_counter() - t0
        reps = int(max(1, min(1024, target_s / per)))
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        dt = time.perf_counter() - t0
        out[name] = {"value": float("%.4g" % (reps * nx / dt / 1e9)), "unit": "Gsamples/s", "cores": 1, "kind": "port",
                     "flags": "gcc -Wall (no -O), as the reference's build scripts",
                     "sample": "%d blocks of %d samples" % (reps, nx)}
    return out


def pcie_inclusive_fft(frames=1 << 16, n=1024, reps=5):
    """The headline transform with HOST buffers on both sides, as the drop-in boundary hands them
    over: pinned host -> device, transform, device -> pinned host, one stream, no overlap."""
    plan = dsp_amd.FftPlan(n, "f32")
    hin = torch.empty((frames, n, 2), dtype=torch.float32).uniform_(-1, 1).pin_memory()
    hout = torch.empty((frames, n, 2), dtype=torch.float32).pin_memory()
    din = torch.empty((frames, n, 2), dtype=torch.float32, device="cuda")
    dout = torch.empty_like(din)

    def once():
        din.copy_(hin, non_blocking=True)
        plan.c2c(torch.view_as_complex(din), dsp_amd.FORWARD, out=torch.view_as_complex(dout))
        hout.copy_(dout, non_blocking=True)
        torch.cuda.synchronize()

    once()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    dt = (time.perf_counter() - t0) / reps
    return {"gflops": round(5.0 * n * math.log2(n) * frames / dt / 1e9, 1),
            "host_gbytes_per_s_each_way": round(frames * n * 8 / dt / 1e9, 1),
            "ms": round(dt * 1e3, 3),
            "sample": "%d frames of N=%d: pinned host -> device, transform, device -> pinned host, serialised" % (frames, n)}


# ---------------------------------------------------------------------------

def main():
    args = parse()
    world, rank, local = setup_dist(args)

    if args.workload == "fft1024":
        fn, info, keep = make_fft(args, rank)
    elif args.workload == "fft_len":
        fn, info, keep = make_fft_len(args.fft_n, args.fft_dtype)
        args.no_secondary = True
    elif args.workload == "chain":
        fn, info, keep = make_chain(args, rank)
        args.no_secondary = True
    elif args.workload in ("fir8191", "ols8191"):
        fn, info, keep = make_long(args, rank, args.workload == "ols8191")
        args.no_secondary = True
    elif args.workload == "ols8191_split":
        fn, info, keep = make_long_split(args, rank, world)
        args.no_secondary = True
    elif args.workload == "c1":
        # BASELINE config 1: the 16-tap table over one 1M-sample signal (the reference's CPU-runnable case)
        args.channels, args.samples, args.no_secondary = 1, 1 << 20, True
        fn, info, keep = make_fir(args, rank, 16)
    else:
        fn, info, keep = make_fir(args, rank, 16 if args.workload == "fir16" else 127, direct=args.workload == "fir127_direct")

    if args.workload != "fft1024" and args.steps > 50:
        args.steps = 50      # the default step count is sized for the 3 ms headline launch
    if args.workload == "ols8191" and args.warmup < 200:
        args.warmup = 200    # 0.2 ms launches: the clocks settle over some tens of milliseconds of work
    if args.workload in ("chain", "fir16", "fir127", "fir127_direct") and args.warmup < 15:
        args.warmup = 15
    dt, ms_launch, ms_median = timed(fn, args.steps, args.warmup, world)
    local_ms = None
    if "local_only" in info:
        _, local_ms, _ = timed(info.pop("local_only"), args.steps, 1, world)
    units_total = info["units"] * world * args.steps
    gflops = info["flop_per_unit"] * units_total / dt / 1e9
    gsamples = info["samples_per_unit"] * units_total / dt / 1e9

    secondary = []
    if rank == 0 and world == 1 and not args.no_secondary:
        del keep, fn
        torch.cuda.empty_cache()
        jobs = [("fir", 16, False), ("fir", 127, False), ("fir", 127, True), ("ols8191",), ("fir8191",), ("chain", 127), ("chain", 16),
                ("fft", 16384, "f32"), ("fft", 32768, "f32"), ("fft", 1 << 20, "f32"), ("fft", 1 << 24, "f32"),
                ("fft", 1024, "f64"), ("fft", 8192, "f64"), ("fft", 1 << 20, "f64")]
        for job in jobs:
            if job[0] == "fir":
                _, nh, direct = job
                if args.workload == ("fir%d%s" % (nh, "_direct" if direct else "")):
                    continue
                f2, i2, k2 = make_fir(args, rank, nh, direct)
                nsec, nwarm = 20, 40
            elif job[0] == "fft":
                f2, i2, k2 = make_fft_len(job[1], job[2])
                nsec, nwarm = 20, 10
            elif job[0] == "chain":
                args.chain_taps = job[1]
                f2, i2, k2 = make_chain(args, rank)
                nsec, nwarm = 10, 15
            else:
                f2, i2, k2 = make_long(args, rank, job[0] == "ols8191")
                nsec, nwarm = (100, 200) if job[0] == "ols8191" else (3, 2)
            # Enough launches ahead of the timed ones that the clocks have settled (some tens of
            # milliseconds of work: the first launches after a pause run 10-15 % slower), as the
            # headline's 200 launches do for themselves.
            d2, ms2, med2 = timed(f2, nsec, nwarm, 1)
            ent = {"workload": i2["workload"],
                   "gsamples_per_s": round(i2.get("samples_per_unit", 1) * i2["units"] * nsec / d2 / 1e9, 2),
                   flops_key(i2): round(i2["flop_per_unit"] * i2["units"] * nsec / d2 / 1e9, 1),
                   "ms_per_step": round(d2 / nsec * 1e3, 4),
                   "roofline": roofline(i2, ms2, med2)}
            if "fir_fft_ms" in i2:
                ent["fir_fft_stages_only_ms"] = round(i2["fir_fft_ms"], 3)
            secondary.append(ent)
            del f2, k2
            torch.cuda.empty_cache()

    cpu = None
    extra = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_fft() if args.workload == "fft1024" else None if args.workload == "fft_len" else cpu_baseline_fir(info["nh"])
        try:
            extra["cpu_fir_as_shipped"] = cpu_as_shipped()
        except (OSError, AttributeError) as e:
            print("cpu_as_shipped skipped (%s)" % e, file=sys.stderr)
        if args.workload == "fft1024":
            extra["pcie_inclusive"] = pcie_inclusive_fft()

    if rank == 0:
        is_fft = args.workload in ("fft1024", "fft_len")
        if is_fft:
            metric = "FFT GFLOP/s (N=%d, c2c %s, 5*N*log2(N) flop per frame)" % (info["n"], args.fft_dtype if args.workload == "fft_len" else "f32")
        elif args.workload == "chain":
            metric = "Gsamples/s through the tonegen -> FIR -> FFT(4096) -> spectrum chain (f32)"
        else:
            metric = "Gsamples/s batched FIR (%d taps, f32)" % info["nh"]
        line = {
            "metric": metric,
            "value": round(gflops if is_fft else gsamples, 2),
            "unit": "GFLOP/s" if is_fft else "Gsamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.workload == "ols8191_split" else "weak", "vs_baseline": None,
            "dtype": args.fft_dtype if args.workload == "fft_len" else "f32", "data": "synthetic",
            "config": {"workload": info["workload"], "kernel": info["kernel"], "per_gpu_units": info["units"],
                       "unit_name": info["unit_name"],
                       "sharding": ("one signal cut into halo'd output segments, one all_gather of the outputs"
                                    if args.workload == "ols8191_split" else "frames/channels split across ranks, no collective")},
            "gsamples_per_s": round(gsamples, 2), flops_key(info): round(gflops, 1),
            "roofline": roofline(info, ms_launch, ms_median),
            "cpu_baseline": cpu,
            "secondary": secondary,
        }
        line.update(extra)
        if local_ms is not None:
            line["without_gather"] = {"ms_per_step": round(local_ms, 4),
                                      "gsamples_per_s": round(info["units"] * world / local_ms / 1e6, 2)}
        if args.workload == "c1":
            # the same signal through the drop-in host routine: double precision, host buffers in
            # and out over PCIe, plan and device buffers created per call as the reference does
            xh = np.random.default_rng(1).uniform(-1, 1, 1 << 20)
            hh = np.asarray(TAPS16, dtype=np.float64)
            dsp_amd.conv0(xh, hh)
            t0 = time.perf_counter()
            for _ in range(5):
                dsp_amd.conv0(xh, hh)
            wms = (time.perf_counter() - t0) / 5 * 1e3
            line["host_wrapper"] = {"routine": "dsp_conv0 (double, host buffers, PCIe-inclusive)", "ms_per_call": round(wms, 3),
                                    "gsamples_per_s": round((1 << 20) / wms / 1e6, 4)}
        if "fir_fft_ms" in info:
            line["fir_fft_stages_only"] = {"ms_per_step": round(info["fir_fft_ms"], 3),
                                           "gsamples_per_s": round(info["units"] / info["fir_fft_ms"] / 1e6, 2)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
