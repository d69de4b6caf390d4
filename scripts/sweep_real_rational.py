#!/usr/bin/env python3
"""PolyphaseResampler<float> with interp > 1: resamp_mfma_real_kernel wherever it has a plan (size and interpolation rules lifted) against what AUTO
runs without it (QDSP_HIP_NO_RM=1), per ratio x taps per phase x call size.  The real-data rules of the rm plan (qdsp_hip.hip: 14 taps per phase
except 3/8-like ratios; rm_min_count for ch == 1) are read off this table.

    python scripts/sweep_real_rational.py > profiles/r04_real_rational.txt        (on the GPU box, ~1 minute)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402

FORCE = len(sys.argv) > 1 and sys.argv[1] == "forced"      # "forced": the kernel wherever it has a plan; default: under the shipped rules
print(f"# scripts/sweep_real_rational.py {'forced' if FORCE else '(shipped rules)'}: us per call, ratio = with the MFMA kernel / without")
for N in (1 << 26, 1 << 23, 1 << 20):
    xr = torch.view_as_real(ops.synth_iq(N, seed=5))[:, 0].contiguous()
    for L, M in [(147, 160), (160, 147), (33, 32), (10, 7), (3, 8), (6, 1), (100, 99), (5, 8), (4, 3), (48, 5), (2, 5)]:
        for tpp in (8, 12, 14, 20, 32):
            taps = bench.lowpass_taps(tpp * L, 0.45 / max(L, M))
            n = N if L <= M else (N // L * M) // M * M
            n = n // M * M
            x = xr[:n]
            out = torch.empty(n // M * L + 64, dtype=torch.float32, device="cuda")
            op = ops.Resampler(taps, L, M, complex_data=False, max_block=0)
            res = {}
            for norm in (0, 1):
                capi.setenv("QDSP_HIP_NO_RM", "1" if norm else None)
                if FORCE:
                    capi.setenv("QDSP_HIP_RM_MIN_COUNT", None if norm else "0")
                    capi.setenv("QDSP_HIP_RM_MIN_INTERP", None if norm else "2")
                op.process(x, out)
                nm = op.last_kernel()["name"]
                reps = 8 if N > (1 << 24) else 60
                op.time_dev(x, out, 3)
                res[norm] = (min(op.time_dev(x, out, reps) for _ in range(3)), nm)
            for k in ("QDSP_HIP_NO_RM", "QDSP_HIP_RM_MIN_COUNT", "QDSP_HIP_RM_MIN_INTERP"):
                capi.setenv(k, None)
            print(f"N {N} real {L}/{M} tpp {tpp}: with {res[0][1]} {res[0][0]*1e3:.1f} | without {res[1][1]} {res[1][0]*1e3:.1f}  ratio {res[0][0]/res[1][0]:.2f}", flush=True)
            op.close()
