"""Time-axis sharding of one long IQ stream over the ranks of a node.

The FIR / decimator / fused-VFO path has no recurrence: y[n] needs x[n-H .. n] only.  So a
stream is cut into chunks, one per rank (one process per GPU), and the ONLY exchange is each
rank's last H input samples (H = ntaps-1 for the FIR, taps-per-phase for the resampler)
handed to the next rank as that rank's filter history: a point-to-point send/recv of ~2 KB
to the ring neighbour over xGMI (torch.distributed `nccl` == RCCL on ROCm).  There is no
all-reduce / all-gather anywhere on the data path.  The NCO needs no communication at all:
every rank puts its phase accumulator at its chunk's first sample (`*_advance`).

The reference has no distributed layer; this module is new, not a translation.

Two cuts of the stream are supported:
  * `partition` + `exchange_halo`: ONE contiguous chunk per rank (a stream that exists up front);
  * `RingStream`: a stream that keeps coming -- block-cyclic, step s / rank r owns samples
    [(s*world + r)*n, +n) -- with the halo hand-off of every step prefetched under the previous
    step's kernel.  bench.py and the multi-rank tests run through this class.

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
    last is a multiple of `align`.  Every chunk that has a successor holds at least `hist`
    samples, so one neighbour hop suffices for the halo: when the stream is too short for
    that on `world` ranks a ValueError says so (use fewer ranks) -- trailing ranks are never
    silently left with a short or empty chunk that they would have to forward a halo from."""
    if world <= 0 or total < 0 or align <= 0 or hist < 0:
        raise ValueError("bad partition arguments")
    per = -(-total // world)            # ceil
    per = -(-per // align) * align      # round up to the alignment
    chunks = []
    for r in range(world):
        s = min(r * per, total)
        e = min(s + per, total)
        chunks.append(Chunk(r, world, s, e - s, 0 if r == 0 else hist))
    if world > 1:
        for c in chunks[:-1]:
            if c.count < hist:
                raise ValueError(
                    f"rank {c.rank} would own {c.count} samples, fewer than the {hist}-sample halo its successor "
                    f"needs ({total} samples over {world} ranks, alignment {align}); use fewer ranks")
    return chunks


def exchange_halo(tail, hist_out, rank: int, world: int, group=None):
    """Ring-neighbour halo of a contiguous partition: send `tail` (my last H input samples) to
    rank+1, receive the previous rank's tail into `hist_out`.  Rank 0 receives nothing (its
    history is the stream's zero initial state: the caller zeroes / resets it); the last rank
    sends nothing.  Works on any torch.distributed backend (nccl/RCCL on GPUs, gloo on CPU
    tensors).  Returns after the transfers have completed from the caller's point of view
    (for nccl: enqueued on the current stream, ordered before subsequent kernels on it)."""
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


def _ctrl_device(group, fallback):
    """Where the tensors of a control-plane collective live: the CPU when the group can carry CPU tensors (gloo, or a
    'cpu:gloo,cuda:nccl' group), else `fallback` (an nccl-only group)."""
    import torch
    import torch.distributed as dist

    try:
        names = str(dist.get_backend(group)).lower()
    except Exception:  # noqa: BLE001
        names = ""
    return torch.device("cpu") if "gloo" in names else torch.device(fallback)


def negotiate_c_ring(rank: int, world: int, group, ctrl_device, *, load=None):
    """Phase 1 of bringing up the C ring on `world` > 1 ranks: decide TOGETHER whether RCCL can be bound on every rank, and carry
    rank 0's RCCL id to all of them.  Every rank issues exactly the same two collectives whatever fails locally --
        broadcast(id, 0)  (a zeroed id when rank 0 could not produce one),  all_reduce(ok, MIN)
    -- so a rank whose library or RCCL is missing can never leave the others waiting in a different collective.
    Returns (library, id bytes) when every rank is able, else (None, reason)."""
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    L, ok, why = None, 1, ""
    try:
        if load is None:
            from . import capi

            load = capi.load
        L = load()
        if int(L.qdsp_hip_ring_available()) != 1:
            ok, why = 0, "RCCL cannot be bound in this process (qdsp_hip_ring_available() == 0)"
    except Exception as e:  # noqa: BLE001
        ok, why = 0, f"libqdsp_hip.so cannot be loaded: {e!r}"
    idbuf = (C.c_char * 128)()
    if rank == 0 and ok:
        rc = int(L.qdsp_hip_ring_unique_id(idbuf))
        if rc != 0:
            ok, why = 0, f"qdsp_hip_ring_unique_id failed ({rc})"
            idbuf = (C.c_char * 128)()
    t = torch.from_numpy(np.frombuffer(idbuf.raw, dtype=np.uint8).copy()).to(ctrl_device)
    dist.broadcast(t, 0, group=group)
    flag = torch.tensor([ok], dtype=torch.int32, device=ctrl_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) != 1:
        return None, (why or "another rank cannot bind RCCL")
    return L, t.cpu().numpy().tobytes()


class _CRing:
    """qdsp_hip_ring_* (include/qdsp_hip.h, qdsp_amd/csrc/ring.cpp): the RCCL send/recv pair of one step, posted on the ring's own HIP
    stream.  `idbytes`: the 128-byte RCCL id (rank 0's, carried by `negotiate_c_ring`); None = a one-rank ring (the rank is its
    own neighbour), which needs no process group at all."""

    def __init__(self, device: int, rank: int, world: int, halo_bytes: int, idbytes=None):
        import ctypes as C

        from . import capi

        self._L = capi.load()
        self._check = capi.check
        idbuf = (C.c_char * 128)()
        if idbytes is None:
            if world != 1:
                raise ValueError("a ring of several ranks needs rank 0's id")
            capi.check(self._L.qdsp_hip_ring_unique_id(idbuf), "qdsp_hip_ring_unique_id")
        else:
            idbuf.raw = bytes(idbytes)
        self._h = C.c_void_p()
        capi.check(self._L.qdsp_hip_ring_create(C.byref(self._h), device, rank, world, idbuf, halo_bytes), "qdsp_hip_ring_create")
        self._C = C

    def post(self, tail_ptr: int, stream: int):
        self._check(self._L.qdsp_hip_ring_post(self._h, tail_ptr, stream), "qdsp_hip_ring_post")

    def complete(self, stream: int):
        halo, prev = self._C.c_void_p(), self._C.c_void_p()
        self._check(self._L.qdsp_hip_ring_complete(self._h, stream, self._C.byref(halo), self._C.byref(prev)), "qdsp_hip_ring_complete")
        return halo.value, prev.value

    def drain(self):
        self._check(self._L.qdsp_hip_ring_drain(self._h), "qdsp_hip_ring_drain")

    def info(self) -> dict:
        """What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice, RCCL version code)."""
        C = self._C
        v = [C.c_int(-1) for _ in range(4)]
        self._check(self._L.qdsp_hip_ring_info(self._h, *[C.byref(x) for x in v]), "qdsp_hip_ring_info")
        return {"comm_ranks": v[0].value, "comm_rank": v[1].value, "comm_device": v[2].value, "rccl_version": v[3].value}

    def set_timing(self, on: bool = True):
        self._check(self._L.qdsp_hip_ring_set_timing(self._h, 1 if on else 0), "qdsp_hip_ring_set_timing")

    def exchange_us(self) -> dict:
        C = self._C
        mean, mx, n = C.c_double(0), C.c_double(0), C.c_longlong(0)
        self._check(self._L.qdsp_hip_ring_exchange_us(self._h, C.byref(mean), C.byref(mx), C.byref(n)), "qdsp_hip_ring_exchange_us")
        return {"mean_us": mean.value, "max_us": mx.value, "exchanges": n.value}

    def close(self):
        if self._h:
            self._L.qdsp_hip_ring_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class RingStream:
    """One rank's end of a continuous stream processed block-cyclically on `world` ranks.

    Step s of rank r processes samples [(s*world + r)*n, +n).  The halo that step needs is the
    tail (last H input samples) of the chunk just before it in the stream: chunk (s, r-1) for
    r > 0, chunk (s-1, world-1) for rank 0 (zeros at s = 0).  So every step all ranks send the
    tail of their chunk to their ring successor; ranks > 0 use what arrives in the same step,
    rank 0 what arrived one step earlier.  That tail is INPUT, known before the step is
    computed: `step(x, out, next_x=...)` posts the exchange for the following step right after
    this step's history is installed and BEFORE this step's kernel is launched, so the RCCL
    send/recv (on RCCL's own stream, ordered behind what is already queued on the compute
    stream) runs under the kernel instead of in front of it.

    op         a qdsp_amd.ops operator with history (`Fir`, `Resampler`, `Vfo`, `Channelizer`), or an
               `Xlator` (no halo; only the NCO bookkeeping applies)
    transport  "device": halos travel by RCCL send / recv through the library's C ring (qdsp_hip_ring_*, qdsp_amd/csrc/
               ring.cpp -- the entry points a C++ graph uses as well); the process group only carries the 128-byte RCCL id
               once (world 1 = the rank is its own ring neighbour: exercises the real send/recv on a one-GPU box, no
               process group needed);
               "host": halos are staged through CPU tensors (gloo) -- several ranks sharing one
               GPU, where RCCL refuses to run; used by the one-GPU rehearsal tests.
    The NCO (if the operator has one) is put at this rank's first sample on construction and
    stepped over the other ranks' chunks after every step -- no communication.
    """

    NBUF = 3   # receive buffers in rotation: rank 0 reads the one filled a step earlier while the next one is in flight

    def __init__(self, op, n: int, rank: int, world: int, *, group=None, transport: str = "device",
                 prefetch: bool = True, align: int = 1, exchange: bool | None = None, ctrl_group=None):
        import torch

        if transport not in ("device", "host"):
            raise ValueError("transport must be 'device' or 'host'")
        if n <= 0 or n % max(1, align):
            raise ValueError(f"chunk of {n} samples is not a positive multiple of the alignment {align}")
        self.op, self.n, self.rank, self.world = op, int(n), int(rank), int(world)
        self.group, self.transport = group, transport
        # control plane (the id broadcast and the all-ranks vote while the C ring comes up): `ctrl_group` if given -- bench.py hands
        # over a gloo group so that no torch RCCL collective ever shares the process with the ring's communicator -- else `group`
        self.ctrl_group = ctrl_group if ctrl_group is not None else group
        self.H = int(getattr(op, "history_len", 0) or 0) if hasattr(op, "set_history_dev") else 0
        if self.world > 1 and self.n < self.H:
            raise ValueError(f"chunk of {n} samples is shorter than the {self.H}-sample halo")
        self.has_nco = hasattr(op, "advance")
        # world 1: exchange only when asked to (the self-ring of the one-GPU RCCL test / bench mode)
        self.exchange = (self.world > 1) if exchange is None else bool(exchange)
        self.exchange = self.exchange and self.H > 0
        self.prefetch = bool(prefetch) and self.exchange and transport == "device"
        # (an operator may name the device its tensors live on -- the CPU tests drive this class with a host-side
        # stand-in; the HIP operators of qdsp_amd.ops live on cuda:<op.device>)
        self.device = torch.device(getattr(op, "torch_device", None) or f"cuda:{getattr(op, 'device', 0)}")
        self.step_index = 0
        self._pending = None          # (reqs, buffer index) of the exchange in flight
        self._posted = 0              # exchanges posted so far
        self._ring = None
        # (QDSP_RING_TRANSPORT=torch: the same exchange through torch.distributed's batch_isend_irecv on device tensors -- round 2's
        # transport, kept as a switch for hosts whose RCCL the library cannot bind; every rank must set it alike)
        import os

        if self.exchange and transport == "device" and self.device.type == "cuda" and os.environ.get("QDSP_RING_TRANSPORT", "c") != "torch":
            self._ring = self._make_c_ring(group)
        if self.exchange and self._ring is None:      # (host transport, or device tensors of a CPU stand-in operator over gloo: the CPU tests)
            z = lambda: torch.zeros(max(self.H, 1), dtype=torch.complex64, device=self.device)  # noqa: E731
            self._recv = [z() for _ in range(self.NBUF)]
            self._zeros = z()
        if self.has_nco and self.world > 1:
            op.advance(self.rank * self.n)

    def _make_c_ring(self, group):
        """The C ring, or None when it cannot be had on EVERY rank (then all ranks take torch.distributed's p2p path together: a ring
        with one end on another transport would hang).  One rank alone has nobody to agree with: the error is the caller's.
        world > 1: `negotiate_c_ring` (identical collectives on every rank whatever fails locally), then the collective create,
        then a second vote on its outcome."""
        import sys

        import torch

        dev_index = self.device.index or 0
        if self.world == 1:
            return _CRing(dev_index, self.rank, 1, self.H * 8)
        import torch.distributed as dist

        cg = self.ctrl_group
        cdev = _ctrl_device(cg, self.device)
        L, got = negotiate_c_ring(self.rank, self.world, cg, cdev)
        ring, err = None, None
        if L is None:
            err = got
        else:
            try:
                ring = _CRing(dev_index, self.rank, self.world, self.H * 8, got)
            except Exception as e:  # noqa: BLE001
                err = repr(e)
            ok = torch.tensor([0 if ring is None else 1], dtype=torch.int32, device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=cg)
            if int(ok.item()) == 0 and ring is not None:
                ring.close()
                ring = None
        if ring is None:
            print(f"qdsp_amd.sharding: rank {self.rank}: the C ring is not available on every rank ({err or 'another rank failed'}); "
                  "halos travel through torch.distributed send / recv instead", file=sys.stderr, flush=True)
        return ring

    def transport_name(self) -> str:
        """Which transport the halo actually travels on (bench.py's `rccl` object)."""
        if not self.exchange:
            return "none"
        if self._ring is not None:
            return "qdsp_hip_ring (RCCL ncclSend/ncclRecv on the ring's own stream)"
        return "gloo, CPU-staged (one-GPU rehearsal)" if self.transport == "host" else "torch.distributed p2p (fallback)"

    # -- the exchange -------------------------------------------------------------------------
    def _post(self, x):
        """Start the hand-off for the step that will process `x`: my tail -> successor, predecessor's -> me."""
        import torch
        import torch.distributed as dist

        k = self._posted % self.NBUF
        self._posted += 1
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        if self.transport == "host":
            tail_h = x[x.numel() - self.H:].cpu()
            halo_h = torch.empty(self.H, dtype=torch.complex64)
            reqs = dist.batch_isend_irecv([
                dist.P2POp(dist.isend, torch.view_as_real(tail_h), nxt, self.group),
                dist.P2POp(dist.irecv, torch.view_as_real(halo_h), prv, self.group),
            ])
            return (reqs, k, halo_h, tail_h)
        # sent straight from the chunk: `x` is the input of the step this exchange belongs to, so it stays untouched until that
        # step has waited for the exchange
        if self._ring is not None:
            self._ring.post(x.data_ptr() + (x.numel() - self.H) * 8, torch.cuda.current_stream(x.device).cuda_stream)
            return ("ring", k, None, None)
        reqs = dist.batch_isend_irecv([
            dist.P2POp(dist.isend, torch.view_as_real(x[x.numel() - self.H:]), nxt, self.group),
            dist.P2POp(dist.irecv, torch.view_as_real(self._recv[k]), prv, self.group),
        ])
        return (reqs, k, None, None)

    def _complete(self, pending):
        reqs, k, halo_h, _ = pending
        for r in reqs:
            r.wait()
        if halo_h is not None:
            self._recv[k].copy_(halo_h)
        return k

    def step(self, x, out=None, next_x=None):
        """Process this rank's chunk `x` (n samples, device tensor) of the current step.  `next_x`: the
        chunk of the following step if it is already resident (its halo is then prefetched under this
        step's kernel).  Returns the operator's output."""
        if x.numel() != self.n:
            raise ValueError(f"chunk of {x.numel()} samples, expected {self.n}")
        if self.exchange and self._ring is not None:
            import torch

            stream = torch.cuda.current_stream(x.device).cuda_stream
            if self._pending is None:
                self._pending = self._post(x)
            halo, prev = self._ring.complete(stream)
            self._pending = None
            # rank 0: what just arrived is the last rank's tail of THIS step = my halo of the NEXT step; this step reads what
            # arrived one step earlier (qdsp_hip_ring_complete's d_prev_halo: zeros before the first exchange)
            self.op.set_history_ptr(prev if self.rank == 0 else halo, stream)
            if self.prefetch and next_x is not None:
                self._pending = self._post(next_x)
        elif self.exchange:
            if self._pending is None:
                self._pending = self._post(x)
            k = self._complete(self._pending)
            self._pending = None
            if self.rank == 0:
                # what just arrived is the last rank's tail of THIS step = my halo of the NEXT step;
                # this step reads what arrived one step earlier (the stream's zero state at step 0)
                src = self._recv[(k - 1) % self.NBUF] if self.step_index > 0 else self._zeros
            else:
                src = self._recv[k]
            self.op.set_history_dev(src)     # *_set_history_dev: asynchronous, on the current stream
            if self.prefetch and next_x is not None:
                self._pending = self._post(next_x)
        y = self.op.process(x, out)
        self.step_index += 1
        if self.has_nco and self.world > 1:
            self.op.advance((self.world - 1) * self.n)      # the call itself advanced by n
        return y

    def drain(self):
        """Complete an exchange posted for a step that will not run (every rank has one in flight)."""
        if self._pending is not None:
            if self._ring is not None:
                self._ring.drain()
            else:
                for r in self._pending[0]:
                    r.wait()
            self._pending = None

    def stream_position(self) -> int:
        """First sample of the chunk the next step() of this rank processes."""
        return (self.step_index * self.world + self.rank) * self.n


def process_stream(op, chunks, n: int, rank: int, world: int, **kw):
    """Run `chunks` (an iterable of this rank's device tensors, one per step, in stream order)
    through `op` on a `RingStream`; yields the outputs.  The iterable is read one chunk ahead so
    each step's successor halo is prefetched."""
    rs = RingStream(op, n, rank, world, **kw)
    it = iter(chunks)
    try:
        cur = next(it)
    except StopIteration:
        return
    while cur is not None:
        nxt = next(it, None)
        yield rs.step(cur, next_x=nxt)
        cur = nxt
    rs.drain()
