#!/usr/bin/env python3
"""decim_win_kernel: outputs per lane (QDSP_HIP_WIN_R) per decimation, 2^26 complex samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from qdsp_amd import capi, ops
n = 1 << 26
x = ops.synth_iq(n, seed=1)
for M, Rs in ((2, (2, 4, 8)), (3, (2, 4)), (4, (1, 2, 4)), (5, (1, 2, 4)), (6, (1, 2)), (8, (1, 2)), (16, (1,))):
    for ntaps in (15, 63, 127, 191):
        row = []
        for R in Rs:
            capi.setenv("QDSP_HIP_WIN_R", str(R))
            op = ops.Resampler(bench.lowpass_taps(ntaps, 0.4 / M), 1, M, max_block=0)
            out = torch.empty(n // M + 8, dtype=torch.complex64, device="cuda")
            op.process(x, out); torch.cuda.synchronize()
            ms = min(op.time_dev(x, out, 10) for _ in range(3))
            row.append(f"R={R} {op.last_kernel()['name'][:9]} {ms:6.3f}")
        print(f"M={M} taps={ntaps:4d} | " + " | ".join(row), flush=True)
