#!/usr/bin/env python3
"""resamp_mfma_kernel: waves queued per SIMD (QDSP_HIP_RM_WAVES_PER_SIMD; 3 are resident), kernel ms per 2^26 input samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from qdsp_amd import capi, ops

n = 1 << 26
x = ops.synth_iq(n, seed=3)
for L, M, tpp in ((147, 160, 16), (160, 147, 16), (48, 50, 20), (441, 480, 8), (10, 7, 8), (6, 1, 10), (24, 125, 12), (3, 8, 20), (5, 6, 20)):
    taps = (bench.lowpass_taps(L * tpp - 3, 0.4 / max(L, M)) * L).astype(np.float32)
    nin = n if L <= M else int(n * M / L)
    nin -= nin % M
    xin = x[:nin]
    row = []
    for nco in (False, True):
        for w in (3, 6, 8, 12, 16):
            capi.setenv("QDSP_HIP_RM_WAVES_PER_SIMD", str(w))
            op = ops.Vfo(taps, L, M, ops.phase_delta(1.0, 0.1234), max_block=0) if nco else ops.Resampler(taps, L, M, max_block=0)
            out = torch.empty(nin // M * L + 8, dtype=torch.complex64, device="cuda")
            op.process(xin, out)
            for _ in range(2):
                op.time_dev(xin, out, 10)
            t = min(op.time_dev(xin, out, 10) for _ in range(5))
            row.append(f"{'nco ' if nco else ''}{w}: {t:.4f}")
            name = op.last_kernel()["name"]
            op.close()
    capi.setenv("QDSP_HIP_RM_WAVES_PER_SIMD", None)
    print(f"{L}/{M} {tpp} taps per phase ({name}): " + "  ".join(row), flush=True)
