#!/usr/bin/env python3
"""Where does the rational MFMA resampler beat what AUTO picks?  Grid of small interpolation / decimation pairs and taps per phase, 2^26 input
samples (output-limited plans: 2^26 outputs): default dispatch against QDSP_HIP_RM_MIN_INTERP=2 (resamp_mfma_kernel wherever it has a plan).

    python scripts/sweep_rm_grid.py > profiles/r03_sweep_rm_grid.txt
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26      # (a reference-sized count, e.g. 1000000: per-call times, the size rule lifted for the forced run)
SMALL = N < (1 << 25)      # a subset of the grid, the size rule lifted for the forced run
x = ops.synth_iq(N, seed=5)


def timed(taps, L, M, xin, forced):
    capi.setenv("QDSP_HIP_RM_MIN_INTERP", "2" if forced else None)
    capi.setenv("QDSP_HIP_RM_MIN_COUNT", "0" if forced and SMALL else None)
    capi.setenv("QDSP_HIP_NO_RM_EXT", None if forced else "1")      # baseline: round 2's rule
    op = ops.Resampler(taps, L, M, max_block=0)
    out = torch.empty(xin.numel() // M * L + 64, dtype=torch.complex64, device="cuda")
    op.process(xin, out)
    op.time_dev(xin, out, 5)
    t = min(op.time_dev(xin, out, max(8, min(200, (1 << 27) // N))) for _ in range(3))
    name = op.last_kernel()["name"]
    op.close()
    capi.setenv("QDSP_HIP_RM_MIN_INTERP", None)
    capi.setenv("QDSP_HIP_RM_MIN_COUNT", None)
    capi.setenv("QDSP_HIP_NO_RM_EXT", None)
    return t, name


print("# L/M taps-per-phase: default kernel ms | forced resamp_mfma ms | forced / default   (2^26 input samples, or 2^26 outputs for interpolators)")
for L in ((3, 4, 5, 7, 8, 12, 16, 25) if SMALL else (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 20, 24, 25, 32)):
    for M in ((3, 5, 7, 9, 24) if SMALL else (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 15, 24, 25, 49)):
        if math.gcd(L, M) != 1:
            continue
        for tpp in (tuple(int(v) for v in os.environ["QDSP_SWEEP_TPP"].split(",")) if os.environ.get("QDSP_SWEEP_TPP") else (16, 32) if SMALL else (8, 16, 24, 32)):
            taps = (bench.lowpass_taps(L * tpp - 3, 0.4 / max(L, M)) * L).astype(np.float32)
            nin = N if L <= M else int(N * M / L)
            nin -= nin % M
            xin = x[:nin]
            t0, n0 = timed(taps, L, M, xin, False)
            t1, n1 = timed(taps, L, M, xin, True)
            if n1 != "resamp_mfma_kernel":
                print(f"{L}/{M} {tpp}: {n0[:18]} {t0:.4f} | no plan", flush=True)
            elif n0 == "resamp_mfma_kernel":
                print(f"{L}/{M} {tpp}: {n0[:18]} {t0:.4f} | (default)", flush=True)
            else:
                print(f"{L}/{M} {tpp}: {n0[:18]} {t0:.4f} | {t1:.4f} | {t1 / t0:.2f}" + ("  <--" if t1 < 0.95 * t0 else ""), flush=True)
