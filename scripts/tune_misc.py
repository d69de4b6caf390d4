#!/usr/bin/env python3
"""Throughput of the paths that have no bench workload: rational resampling, real-valued data."""
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
xr = torch.view_as_real(x)[:, 0].contiguous()
for (L, M, ntaps) in ((2, 3, 126), (3, 2, 189), (147, 160, 147 * 24), (160, 147, 160 * 24), (1, 20, 400), (10, 1, 240)):
    taps = (bench.lowpass_taps(ntaps, 0.4 / max(L, M)) * L).astype(np.float32)
    op = ops.Resampler(taps, L, M, max_block=0)
    out = torch.empty(n * L // M + 8, dtype=torch.complex64, device="cuda")
    ms = timeit(op, x, out)
    print(f"resamp cf32 L={L:4d} M={M:4d} ntaps={ntaps:5d} {op.last_kernel()['name']:18s} {ms:8.3f} ms  in {n/ms/1e6:7.1f} Gs/s  out {n*L/M/ms/1e6:7.1f} Gs/s", flush=True)
for ntaps in (63, 256):
    taps = bench.lowpass_taps(ntaps, 0.05)
    f = ops.Fir(taps, complex_data=False, max_block=0)
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    ms = timeit(f, xr, out)
    print(f"fir f32 ntaps={ntaps:4d} {f.last_kernel()['name']:18s} {ms:8.3f} ms {n/ms/1e6:7.1f} Gs/s", flush=True)
    r = ops.Resampler(taps, 1, 8, complex_data=False, max_block=0)
    out = torch.empty(n // 8 + 8, dtype=torch.float32, device="cuda")
    ms = timeit(r, xr, out)
    print(f"decim8 f32 ntaps={ntaps:4d} {r.last_kernel()['name']:18s} {ms:8.3f} ms {n/ms/1e6:7.1f} Gs/s", flush=True)
