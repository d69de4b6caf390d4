#!/usr/bin/env python3
"""Fused VFO (NCO + FIR + decimate) vs the plain decimator on the same taps, 2^26 complex samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qdsp_amd import ops
n = 1 << 26
x = ops.synth_iq(n, seed=1)
inc = ops.phase_delta(2.4e6, -300e3)
for M, ntaps in ((10, 97), (8, 63), (4, 63), (16, 127), (5, 31), (8, 256), (10, 400)):
    taps = bench.lowpass_taps(ntaps, 0.4 / M)
    out = torch.empty(n // M + 8, dtype=torch.complex64, device="cuda")
    row = []
    for name, op in (("decim", ops.Resampler(taps, 1, M, max_block=0)), ("vfo", ops.Vfo(taps, 1, M, inc, max_block=0))):
        op.process(x, out); torch.cuda.synchronize()
        ms = min(op.time_dev(x, out, 10) for _ in range(3))
        row.append(f"{name} {op.last_kernel()['name'][:13]:13s} {ms:6.3f} ms {n/ms/1e6:6.1f} Gs/s")
    print(f"M={M:2d} taps={ntaps:4d} | " + " | ".join(row), flush=True)
