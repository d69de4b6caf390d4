#!/usr/bin/env python3
"""Large integer decimations (the VFO's usual job: 2.4 Msps -> 48 kHz is M = 50), complex data, 2^26 samples:
direct forms vs overlap-save with the strided store, plain decimator and fused VFO."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from qdsp_amd import ops

def timeit(op, x, out, iters=10):
    op.process(x, out); torch.cuda.synchronize()
    return min(op.time_dev(x, out, iters) for _ in range(3))

n = 1 << 26
x = ops.synth_iq(n, seed=1)
for M, ntaps in ((9, 63), (14, 127), (17, 69), (20, 255), (25, 201), (32, 255), (40, 321), (50, 201), (50, 401), (64, 513), (100, 401), (100, 801), (125, 501), (128, 1025), (192, 1537), (250, 1001), (250, 2001)):
    taps = bench.lowpass_taps(ntaps, 0.4 / M)
    for vfo in (False, True):
        row = []
        for mode in (1, 2, 0):
            op = ops.Vfo(taps, 1, M, ops.phase_delta(1.0, 0.1234), max_block=0) if vfo else ops.Resampler(taps, 1, M, max_block=0)
            op.set_mode(mode)
            out = torch.empty(n // M + 8, dtype=torch.complex64, device="cuda")
            try:
                ms = timeit(op, x, out)
                row.append(f"{['auto','direct','fft'][mode]} {op.last_kernel()['name'][:14]:14s} {ms:6.3f} ms {n/ms/1e6:6.1f} Gs/s")
            except Exception as e:
                row.append(f"{['auto','direct','fft'][mode]} failed {e}")
        print(f"M={M:3d} taps={ntaps:4d} {'vfo' if vfo else 'dec'} | " + " | ".join(row), flush=True)
