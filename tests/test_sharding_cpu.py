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
