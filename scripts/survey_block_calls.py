#!/usr/bin/env python3
"""Survey: per-call time of reference-sized calls over a broad set of plans (complex and real), to find kernels whose
time does not come down with the call size (a throughput arrangement on a latency-bound call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qdsp_amd import ops

def run(name, mk, L, M, cplx, sizes):
    row = [f"{name:26s}"]
    for n in sizes:
        nn = n - n % M
        x = ops.synth_iq(nn, seed=1) if cplx else torch.randn(nn, dtype=torch.float32, device="cuda")
        out = torch.empty(nn * L // M + 8, dtype=torch.complex64 if cplx else torch.float32, device="cuda")
        op = mk(); op.process(x, out); torch.cuda.synchronize()
        us = min(op.time_dev(x, out, 100) for _ in range(3)) * 1e3
        row.append(f"{us:6.1f} {op.last_kernel()['name'][:13]:13s}")
        op.close()
    print(" | ".join(row), flush=True)

sizes = (4096, 65536, 1_000_000)
print(f"{'':26s} | " + " | ".join(f"{n:>20d}" for n in sizes))
lp = bench.lowpass_taps
inc = ops.phase_delta(1.0, 0.1234)
for cplx in (True, False):
    tag = "c" if cplx else "r"
    for nt in (15, 63, 127, 256, 600, 1500):
        run(f"fir {tag} {nt}", lambda: ops.Fir(lp(nt, 0.1), complex_data=cplx, max_block=0), 1, 1, cplx, sizes)
    for M, nt in ((2, 31), (2, 127), (3, 63), (4, 255), (5, 127), (8, 63), (8, 256), (10, 81), (16, 129), (16, 600), (25, 201), (50, 401), (64, 513), (100, 801), (200, 1601), (7, 1000)):
        run(f"dec {tag} /{M} {nt}", lambda: ops.Resampler(lp(nt, 0.4 / M), 1, M, complex_data=cplx, max_block=0), 1, M, cplx, sizes)
        if cplx:
            run(f"vfo /{M} {nt}", lambda: ops.Vfo(lp(nt, 0.4 / M), 1, M, inc, max_block=0), 1, M, True, sizes)
    for L, M, nt in ((2, 1, 31), (3, 2, 100), (4, 1, 64), (5, 3, 160), (7, 5, 141), (12, 5, 240), (16, 1, 256), (33, 32, 1000), (48, 125, 2000), (160, 147, 2048), (441, 480, 4000)):
        run(f"rs {tag} {L}/{M} {nt}", lambda: ops.Resampler(lp(nt, 0.4 / max(L, M)) * L, L, M, complex_data=cplx, max_block=0), L, M, cplx, sizes)
