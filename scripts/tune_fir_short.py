#!/usr/bin/env python3
"""FIR<complex_t> with short filters: strided-window direct kernel (M = 1) vs overlap-save, 2^26 samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from qdsp_amd import capi, ops
n = 1 << 26
x = ops.synth_iq(n, seed=1)
out = torch.empty(n, dtype=torch.complex64, device="cuda")
for ntaps in (3, 7, 15, 23, 31, 47, 63):
    row = []
    for R in (4, 8):
        capi.setenv("QDSP_HIP_WIN_R", str(R))
        capi.setenv("QDSP_HIP_WIN_MAX_TAPS", "200")
        op = ops.Fir(bench.lowpass_taps(ntaps, 0.2), max_block=0)
        op.process(x, out); torch.cuda.synchronize()
        row.append(f"R={R} {op.last_kernel()['name'][:9]} {min(op.time_dev(x, out, 10) for _ in range(3)):6.3f}")
    op = ops.Fir(bench.lowpass_taps(ntaps, 0.2), max_block=0)
    op.set_mode(op.FFT)
    op.process(x, out); torch.cuda.synchronize()
    row.append(f"fft {min(op.time_dev(x, out, 10) for _ in range(3)):6.3f}")
    print(f"taps={ntaps:3d} | " + " | ".join(row), flush=True)
