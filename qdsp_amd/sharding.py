"""Time-axis sharding of one long IQ stream over the ranks of a node.

The FIR / decimator / fused-VFO path has no recurrence: y[n] needs x[n-H .. n] only.  So a
stream is cut into contiguous chunks, one per rank (one process per GPU), and the ONLY
exchange is each rank's last H input samples (H = ntaps-1 for the FIR, taps-per-phase for
the resampler) handed to the next rank as that rank's filter history: a point-to-point
send/recv of ~2 KB to the ring neighbour over xGMI (torch.distributed `nccl` == RCCL on
ROCm).  There is no all-reduce / all-gather anywhere on the data path.  The NCO needs no
communication at all: rank r starts its phase accumulator at r*chunk (`advance`).

The reference has no distributed layer; this module is new, not a translation.

Chunk starts are aligned so the sharded result equals the single-call result:
  * multiples of `decim`, so the resampler's per-call phase restart (src/dsp/resampling.h:
    114,121) lands where the one-call loop would be anyway;
  * multiples of 512 when the VOLK rotator's renormalisation cadence is emulated
    (qdsp_hip_xlate_*_set_volk_gain), so the magnitude sawtooth lines up.
"""
from __future__ import annotations

import math
from dataclasses import dataclass


@dataclass(frozen=True)
class Chunk:
    rank: int
    world: int
    start: int   # first input sample owned by this rank
    count: int   # input samples owned
    halo: int    # history samples needed from the previous rank (0 for rank 0 -> zeros)


def chunk_alignment(decim: int = 1, interp: int = 1, rotator_cadence: int = 0) -> int:
    """Smallest chunk-start granularity for which sharded == single-call output."""
    a = max(1, decim // math.gcd(decim, interp))
    if rotator_cadence:
        a = a * rotator_cadence // math.gcd(a, rotator_cadence)
    return a


def partition(total: int, world: int, hist: int, align: int = 1) -> list[Chunk]:
    """Contiguous, aligned, near-equal chunks covering [0, total).  Every chunk except the
    last is a multiple of `align`; every chunk except possibly the last holds at least
    `hist` samples so one neighbour hop suffices for the halo."""
    if world <= 0 or total < 0 or align <= 0:
        raise ValueError("bad partition arguments")
    per = -(-total // world)            # ceil
    per = -(-per // align) * align      # round up to the alignment
    if world > 1 and per < hist:
        raise ValueError(f"chunk of {per} samples is shorter than the {hist}-sample halo; use fewer ranks")
    chunks = []
    for r in range(world):
        s = min(r * per, total)
        e = min(s + per, total)
        chunks.append(Chunk(r, world, s, e - s, 0 if r == 0 else hist))
    return chunks


def exchange_halo(tail, hist_out, rank: int, world: int, group=None):
    """Ring-neighbour halo: send `tail` (my last H input samples) to rank+1, receive the
    previous rank's tail into `hist_out`.  Rank 0 receives nothing (its history is the
    stream's zero initial state: the caller zeroes / resets it); the last rank sends nothing.
    Works on any torch.distributed backend (nccl/RCCL on GPUs, gloo on CPU tensors).
    Returns after the transfers have completed from the caller's point of view (for nccl:
    enqueued on the current stream, ordered before subsequent kernels on it)."""
    import torch.distributed as dist

    if world == 1:
        return
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, tail, rank + 1, group))
    if rank > 0:
        ops.append(dist.P2POp(dist.irecv, hist_out, rank - 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
