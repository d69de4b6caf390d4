#!/usr/bin/env python3
"""Throughput map of short filters: FIR and integer decimators at 7..127 taps (the sizes an SDR chain is
made of), direct form vs overlap-save, complex data, 2^26 samples."""
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
for M in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
    for ntaps in (7, 15, 31, 63, 127):
        taps = bench.lowpass_taps(ntaps, 0.4 / max(M, 2))
        row = []
        for mode in (1, 2, 0):
            op = ops.Fir(taps, max_block=0) if M == 1 else ops.Resampler(taps, 1, M, max_block=0)
            op.set_mode(mode)
            out = torch.empty(n // M + 8, dtype=torch.complex64, device="cuda")
            try:
                ms = timeit(op, x, out)
                row.append(f"{['auto','direct','fft'][mode]} {op.last_kernel()['name'][:12]:12s} {ms:6.3f} ms {n/ms/1e6:6.1f} Gs/s")
            except Exception as e:
                row.append(f"{['auto','direct','fft'][mode]} failed")
        print(f"M={M:2d} taps={ntaps:4d} | " + " | ".join(row), flush=True)
