#!/usr/bin/env python3
"""Per-call time of the device path against the call size (the reference's blocks hand over at most 1e6 samples
per call, stream.h:7): where launch overhead takes over from the kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from qdsp_amd import ops

def timeit(op, x, out, iters=200):
    op.process(x, out); torch.cuda.synchronize()
    return min(op.time_dev(x, out, iters) for _ in range(3))

taps = bench.lowpass_taps(256, 1.0 / 16.0)
taps63 = bench.lowpass_taps(63, 0.1)
t401 = bench.lowpass_taps(401, 0.4 / 50)
mk = {
    "fir256": lambda: ops.Fir(taps, max_block=0),
    "fir63": lambda: ops.Fir(taps63, max_block=0),
    "decim8/256": lambda: ops.Resampler(taps, 1, 8, max_block=0),
    "vfo8/256": lambda: ops.Vfo(taps, 1, 8, ops.phase_delta(1.0, 0.1234), max_block=0),
    "vfo50/401": lambda: ops.Vfo(t401, 1, 50, ops.phase_delta(1.0, 0.1234), max_block=0),
    "xlate": lambda: ops.Xlator(phase_inc=ops.phase_delta(1.0, 0.1234), max_block=0),
}
for name, f in mk.items():
    row = []
    for lg in (16, 18, 20, 22, 23, 24, 25):
        n = 1 << lg
        if lg == 20:
            n = 1_000_000
        x = ops.synth_iq(n, seed=1)
        out = torch.empty(n + 8, dtype=torch.complex64, device="cuda")
        op = f()
        ms = timeit(op, x, out)
        row.append(f"{n:>8d}: {ms*1e3:7.1f} us {n/ms/1e6:6.1f} Gs/s {op.last_kernel()['name'][:10]}")
    print(f"{name:11s} | " + " | ".join(row), flush=True)
