#!/usr/bin/env python3
"""fir_fft1k_kernel (1024-point segments, one wave each) against the kernels AUTO picked before it, per tap count
and call size: where the small-call form starts (tap count) and where the 4096-point kernels take over (call size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qdsp_amd import capi, ops

def t(mk, x, out, env):
    for k, v in env.items(): capi.setenv(k, v)
    op = mk(); op.process(x, out); torch.cuda.synchronize()
    us = min(op.time_dev(x, out, 100) for _ in range(3)) * 1e3
    name = op.last_kernel()["name"]; op.close()
    for k in env: capi.setenv(k, None)
    return us, name

sizes = (131072, 262144, 524288, 1_000_000, 2 << 20, 3 << 20, 4 << 20, 6 << 20)
for kind, M in (("fir", 1), ("dec", 2), ("dec", 4), ("dec", 8), ("dec", 5), ("vfo", 8), ("vfo", 5)):
    for ntaps in (15, 31, 63, 95, 127, 256, 401, 513):
        taps = bench.lowpass_taps(ntaps, 0.4 / max(M, 2))
        if kind == "fir": mk = lambda: ops.Fir(taps, max_block=0)
        elif kind == "dec": mk = lambda: ops.Resampler(taps, 1, M, max_block=0)
        else: mk = lambda: ops.Vfo(taps, 1, M, ops.phase_delta(1.0, 0.1234), max_block=0)
        row = [f"{kind} M {M} taps {ntaps:4d}"]
        for n in sizes:
            x = ops.synth_iq(n, seed=1); out = torch.empty(n + 8, dtype=torch.complex64, device="cuda")
            a, an = t(mk, x, out, {"QDSP_HIP_NO_FFT1K": "1"})
            b, bn = t(mk, x, out, {"QDSP_HIP_FFT1K_MAX_COUNT": str(1 << 30), "QDSP_HIP_FFT_MIN_TAPS_SMALL": "2", "QDSP_HIP_FFT_MIN_TAPS_DECIM": "2", "QDSP_HIP_FIR_LAT_MAX_WORK": "0", "QDSP_HIP_NO_WIN": "1"})
            row.append(f"{n:>8d}: {a:5.1f} {an[:9]:9s} | {b:5.1f} {bn[:9]:9s}")
        print("  ".join(row), flush=True)
