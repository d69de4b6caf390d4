"""The N > 1 path on CPU: world_size-2 (and 3) gloo runs of qdsp_amd.sharding -- chunk
partition + ring-neighbour halo hand-off -- with the oracle standing in for the per-rank
filter, checked against the unsharded oracle run.  (On GPUs the same exchange runs over
RCCL with the HIP filter; bench.py --gpus N checks that leg against an unsharded device
filter.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
from qdsp_amd import sharding


def test_partition_properties():
    for total, world, hist, align in ((1000, 2, 255, 1), (1 << 20, 8, 255, 512), (12345, 3, 63, 8), (4096, 4, 100, 16)):
        ch = sharding.partition(total, world, hist, align)
        assert len(ch) == world and ch[0].start == 0 and ch[0].halo == 0
        assert sum(c.count for c in ch) == total
        for a, b in zip(ch, ch[1:]):
            assert a.start + a.count == b.start and b.start % align == 0 and b.halo == hist
    with pytest.raises(ValueError):
        sharding.partition(100, 4, 255)
    assert sharding.chunk_alignment(8) == 8 and sharding.chunk_alignment(8, 1, 512) == 512
    assert sharding.chunk_alignment(10, 1, 512) == 2560 and sharding.chunk_alignment(3, 2) == 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = 40_000
        x = O.synth_iq(0, total, seed=77)
        taps = O.lowpass_taps_f64(256, 1 / 16)
        if case == "fir":
            op, hist, align = O.Fir(taps), 255, 1
        elif case == "decim8":
            op, hist, align = O.Resampler(taps, 1, 8), 256, sharding.chunk_alignment(8)
        else:  # fused VFO: NCO phase needs no communication, only the (rotated-input) halo does
            op, hist, align = O.Resampler(taps, 1, 8), 256, sharding.chunk_alignment(8, 1, 512)
        c = sharding.partition(total, world, hist, align)[rank]
        mine = x[c.start:c.start + c.count]
        if case == "vfo":
            xl = O.Xlator(1.0, 0.1234, exact=True, volk_gain=True)
            xl.turns.value = (c.start * np.arctan2(float(xl.delta[1]), float(xl.delta[0])) / (2 * np.pi)) % 1.0  # advance(start)
            mine = xl.process(mine)
        tail = torch.from_numpy(np.ascontiguousarray(mine[-hist:]).view(np.float32).copy())
        halo = torch.zeros(2 * hist, dtype=torch.float32)
        sharding.exchange_halo(tail, halo, rank, world)
        # the oracle keeps one more (unused) history slot for the FIR: history = last ntaps samples
        h = np.zeros(len(op.hist), np.float32)
        h[len(h) - 2 * hist:] = halo.numpy()
        op.hist[:] = h
        y = op.process(mine)
        np.save(os.path.join(outdir, f"{case}_{world}_{rank}.npy"), y)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["fir", "decim8", "vfo"])
def test_sharded_equals_unsharded(tmp_path, world, case):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    y = np.concatenate([np.load(tmp_path / f"{case}_{world}_{r}.npy") for r in range(world)])
    x = O.synth_iq(0, 40_000, seed=77)
    taps = O.lowpass_taps_f64(256, 1 / 16)
    if case == "fir":
        want = O.Fir(taps).process(x)
    elif case == "decim8":
        want = O.Resampler(taps, 1, 8).process(x)
    else:
        want = O.Resampler(taps, 1, 8).process(O.Xlator(1.0, 0.1234, exact=True, volk_gain=True).process(x))
    assert len(y) == len(want)
    if case == "vfo":
        assert np.abs(y - want).max() < 1e-6      # phase restart per rank: float rounding of the seed only
    else:
        assert np.array_equal(y, want)            # same arithmetic, same order: bit-identical


# ---------------------------------------------------------------------------------------------
# RingStream (the block-cyclic runner bench.py and the GPU multi-rank tests use), driven on CPU
# tensors over gloo with the oracle standing in for the per-rank operator.
# ---------------------------------------------------------------------------------------------
class _OracleOp:
    """The operator surface RingStream needs (history_len, set_history_dev, process, advance, torch_device),
    backed by the oracle.  `vfo`: history arrives as RAW input samples and is rotated here with the NCO phases
    those samples had, as qdsp_hip_xlate_fir_decim_cf32_set_history_dev does."""

    torch_device = "cpu"

    def __init__(self, case):
        taps = O.lowpass_taps_f64(256, 1 / 16)
        self.case = case
        self.op = O.Fir(taps) if case == "fir" else O.Resampler(taps, 1, 8)
        self.history_len = 255 if case == "fir" else 256
        self.pos = 0
        if case == "vfo":
            self.xl = O.Xlator(1.0, 0.1234, exact=True, volk_gain=True)
            self.dt = np.arctan2(float(self.xl.delta[1]), float(self.xl.delta[0])) / (2 * np.pi)
            self.advance = self._advance            # (only NCO-bearing operators have advance())

    def _advance(self, n):
        self.pos += int(n)

    def _rot(self, x, first):
        self.xl.turns.value = (first * self.dt) % 1.0
        return self.xl.process(x)

    def set_history_dev(self, t):
        h = t.numpy().copy()
        if self.case == "vfo":
            # (the magnitude sawtooth restarts per call: history samples sit at the END of the previous chunk,
            # whose length is a multiple of 512, so they carry gains |inc|^(512-H..511))
            pad = np.zeros(512 - self.history_len, np.complex64)
            h = self._rot(np.concatenate([pad, h]), self.pos - 512)[len(pad):]
        buf = np.zeros(len(self.op.hist), np.float32)
        buf[len(buf) - 2 * self.history_len:] = h.view(np.float32)
        self.op.hist[:] = buf

    def process(self, x, out=None):
        a = x.numpy()
        if self.case == "vfo":
            a = self._rot(a, self.pos)
            self.pos += len(a)
        return torch.from_numpy(self.op.process(a))


def _ring_worker(rank, world, port, case, outdir, steps, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = O.synth_iq(0, steps * world * n, seed=78)
        align = sharding.chunk_alignment(1 if case == "fir" else 8, 1, 512 if case == "vfo" else 0)
        rs = sharding.RingStream(_OracleOp(case), n, rank, world, align=align)
        chunks = [torch.from_numpy(x[(s * world + rank) * n:(s * world + rank + 1) * n].copy()) for s in range(steps)]
        ys = list(sharding.process_stream(rs.op, chunks, n, rank, world, align=align)) if case == "decim8" else \
            [rs.step(c, next_x=chunks[i + 1] if i + 1 < steps else None) for i, c in enumerate(chunks)]
        rs.drain()
        assert rs.stream_position() == (steps * world + rank) * n or case == "decim8"
        for s, y in enumerate(ys):
            np.save(os.path.join(outdir, f"{case}_{s}_{rank}.npy"), y.numpy())
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["fir", "decim8", "vfo"])
def test_ring_stream_block_cyclic_equals_unsharded(tmp_path, world, case):
    """Step s / rank r owns samples [(s*world + r)*n, +n): rank 0 reads the halo that arrived a step earlier,
    the others the one of the same step; the NCO is stepped over the other ranks' chunks without communication."""
    steps, n = 3, 2048
    mp.spawn(_ring_worker, args=(world, _free_port(), case, str(tmp_path), steps, n), nprocs=world, join=True)
    y = np.concatenate([np.load(tmp_path / f"{case}_{s}_{r}.npy") for s in range(steps) for r in range(world)])
    x = O.synth_iq(0, steps * world * n, seed=78)
    taps = O.lowpass_taps_f64(256, 1 / 16)
    if case == "fir":
        want = O.Fir(taps).process(x)
    elif case == "decim8":
        want = O.Resampler(taps, 1, 8).process(x)
    else:
        want = O.Resampler(taps, 1, 8).process(O.Xlator(1.0, 0.1234, exact=True, volk_gain=True).process(x))
    assert len(y) == len(want)
    if case == "vfo":
        assert np.abs(y - want).max() < 1e-6
    else:
        assert np.array_equal(y, want)


def test_partition_refuses_short_trailing_chunks():
    with pytest.raises(ValueError):
        sharding.partition(1000, 8, 255, 512)        # ranks 2..7 would own nothing yet be asked for a halo
    ch = sharding.partition(1000, 2, 255, 512)
    assert [c.count for c in ch] == [512, 488]
    with pytest.raises(ValueError):
        sharding.RingStream(_OracleOp("fir"), 1000, 0, 2, align=512)   # chunk not a multiple of the alignment


# ---------------------------------------------------------------------------------------------
# Bringing up the C ring: every rank must issue the same collectives whatever fails locally
# (rank 0 failing before the id broadcast left the other ranks inside it).
# ---------------------------------------------------------------------------------------------
def _vote_worker(rank, world, port, disabled_rank, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if rank == disabled_rank:
        os.environ["QDSP_RING_DISABLE_RCCL"] = "1"      # read once, when this process first asks for RCCL
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, got = sharding.negotiate_c_ring(rank, world, None, sharding._ctrl_device(None, "cpu"))
        # a collective AFTER the negotiation: if the ranks had taken different paths this would pair with a left-over one
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        assert t.item() == world * (world + 1) / 2
        with open(os.path.join(outdir, f"vote_{rank}.txt"), "w") as f:
            f.write("none" if L is None else got.hex())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("disabled_rank", [0, 1, -1])
def test_c_ring_vote_is_the_same_collective_sequence_on_every_rank(tmp_path, disabled_rank):
    """RCCL forced off in the library of ONE rank (rank 0: the id's owner; rank 1: a receiver), or on none: all ranks come
    back with the same verdict -- nobody hangs in a collective the failing rank skipped -- and, when all are able, with the
    same 128-byte id."""
    world = 2
    ctx = mp.spawn(_vote_worker, args=(world, _free_port(), disabled_rank, str(tmp_path)), nprocs=world, join=False)
    import time

    t_end = time.time() + 120
    while not ctx.join(timeout=1.0):
        assert time.time() < t_end, "the ranks did not finish: mismatched collectives"
    got = [open(tmp_path / f"vote_{r}.txt").read() for r in range(world)]
    assert got[0] == got[1]
    if disabled_rank >= 0:
        assert got[0] == "none"
    else:
        assert len(got[0]) == 256 and set(got[0]) != {"0"}
