#!/usr/bin/env python3
"""Reference-sized calls on REAL streams (FIR<float>, PolyphaseResampler<float>: the demodulators' audio paths)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qdsp_amd import ops

mk = {}
for nt in (31, 63, 127, 256):
    mk[f"fir_f32/{nt}"] = (lambda nt=nt: ops.Fir(bench.lowpass_taps(nt, 1 / 16), complex_data=False, max_block=0), 1, 1)
mk["dec5_f32/127"] = (lambda: ops.Resampler(bench.lowpass_taps(127, 0.08), 1, 5, complex_data=False, max_block=0), 1, 5)
mk["rs_24_125_f32/1001"] = (lambda: ops.Resampler(bench.lowpass_taps(1001, 0.4 / 125) * 24, 24, 125, complex_data=False, max_block=0), 24, 125)
sizes = [int(s) for s in sys.argv[1:]] or [16384, 65536, 262144, 1_000_000]
for n in sizes:
    for name, (f, L, M) in mk.items():
        nn = n - n % M
        x = torch.randn(nn, dtype=torch.float32, device="cuda")
        out = torch.empty(nn * L // M + 8, dtype=torch.float32, device="cuda")
        op = f()
        op.process(x, out)
        torch.cuda.synchronize()
        ms = min(op.time_dev(x, out, 200) for _ in range(3))
        print(f"{name:20s} {nn:8d}: {ms * 1e3:6.2f} us per call  {op.last_kernel()['name']}", flush=True)
# the same ratios on complex streams
for name, (L, M, nt) in {"rs_24_125_cf32/1001": (24, 125, 1001), "rs_147_160_cf32/2048": (147, 160, 2048), "rs_2_3_cf32/64": (2, 3, 64), "rs_1_3_cf32/400": (1, 3, 400)}.items():
    for n in sizes:
        nn = n - n % M
        x = ops.synth_iq(nn, seed=1)
        out = torch.empty(nn * L // M + 8, dtype=torch.complex64, device="cuda")
        op = ops.Resampler(bench.lowpass_taps(nt, 0.4 / max(L, M)) * L, L, M, max_block=0)
        op.process(x, out)
        torch.cuda.synchronize()
        ms = min(op.time_dev(x, out, 200) for _ in range(3))
        print(f"{name:20s} {nn:8d}: {ms * 1e3:6.2f} us per call  {op.last_kernel()['name']}", flush=True)
