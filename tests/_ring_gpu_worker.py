"""One rank of the multi-rank GPU tests (tests/test_gpu_ring.py): the HIP operator of `case` behind
qdsp_amd.sharding.RingStream, on cuda:0.

    python _ring_gpu_worker.py <case> <backend> <world> <rank> <port> <outdir> <steps> <n>

backend "gloo": several ranks share the one GPU of the test box; halos are staged through CPU tensors
(RingStream transport "host").  backend "nccl": world 1, the rank is its own ring neighbour over real RCCL
(transport "device").  Every rank regenerates its chunks of the seeded stream on the device and saves its
outputs; the parent concatenates them in stream order and compares with the unsharded CPU oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_op(ops, case):
    from bench import lowpass_taps

    inc = ops.phase_delta(1.0, 0.1234)
    if case == "fir256":
        return ops.Fir(lowpass_taps(256, 1 / 16), max_block=0)
    if case == "decim8":
        return ops.Resampler(lowpass_taps(256, 1 / 16), 1, 8, max_block=0)
    if case == "xlate_fir_decim8":
        return ops.Vfo(lowpass_taps(256, 1 / 16), 1, 8, inc, max_block=0)
    if case == "vfo50":
        return ops.Vfo(lowpass_taps(401, 0.4 / 50), 1, 50, inc, max_block=0)
    if case == "chan64":
        incs = [ops.phase_delta(1.0, -(c - 31.5) / 64.0) for c in range(64)]
        return ops.Channelizer(lowpass_taps(256, 1 / 128), 1, 64, incs, max_block=0)
    raise SystemExit(f"unknown case {case}")


def main():
    case, backend, world, rank, port, outdir, steps, n = sys.argv[1:9]
    world, rank, steps, n = int(world), int(rank), int(steps), int(n)
    import torch
    import torch.distributed as dist

    from qdsp_amd import ops
    from qdsp_amd.sharding import RingStream, chunk_alignment

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        op = make_op(ops, case)
        decim = {"fir256": 1, "decim8": 8, "xlate_fir_decim8": 8, "vfo50": 50, "chan64": 64}[case]
        align = chunk_alignment(decim, 1, 512 if hasattr(op, "advance") else 0)
        rs = RingStream(op, n, rank, world, transport="device" if backend == "nccl" else "host", align=align,
                        exchange=True)
        chunks = [ops.synth_iq(n, first_sample=(s * world + rank) * n, seed=4321, device=0) for s in range(steps)]
        for s in range(steps):
            y = rs.step(chunks[s], next_x=chunks[s + 1] if s + 1 < steps else None)
            torch.cuda.synchronize()
            np.save(os.path.join(outdir, f"{case}_{s}_{rank}.npy"), y.cpu().numpy())
        rs.drain()
        name = op.last_kernel()["name"]
        with open(os.path.join(outdir, f"{case}_kernel_{rank}.txt"), "w") as f:
            f.write(name)
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
