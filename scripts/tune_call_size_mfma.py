"""MFMA kernels against the kernels AUTO picks without them, by call size (the thresholds mf_min_count / rm_min_count in
qdsp_hip.hip come from this table: profiles/r02_tune_call_size_mfma.txt). Run on the GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from qdsp_amd import capi, ops
capi.setenv("QDSP_HIP_MF_MIN_COUNT", "0")
capi.setenv("QDSP_HIP_RM_MIN_COUNT", "0")
shapes = [(1, 50, 401), (1, 16, 129), (1, 25, 201), (1, 100, 801), (1, 200, 1601), (1, 50, 1201), (147, 160, 2349), (160, 147, 2557), (10, 7, 77), (3, 8, 57)]
for L, M, ntaps in shapes:
    taps = (O.lowpass_taps_f64(ntaps, 0.4 / max(L, M)) * L).astype(np.float32)
    for vfo in (True, False):
        row = []
        for n in (65536, 262144, 1_000_000, 2_000_000, 4_000_000):
            n = n // M * M
            x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
            out = torch.empty(n // M * L + 8, dtype=torch.complex64, device="cuda")
            t = []
            for off in ("0", "1"):
                capi.setenv("QDSP_HIP_NO_MF", off)
                capi.setenv("QDSP_HIP_NO_RM", off)
                op = ops.Vfo(taps, L, M, ops.phase_delta(1.0, 0.1234), max_block=0) if vfo else ops.Resampler(taps, L, M, max_block=0)
                for _ in range(10): op.process(x, out=out)
                torch.cuda.synchronize()
                t.append(min(op.time_dev(x, out, 40) for _ in range(3)) * 1e3)
                name = op.last_kernel()["name"][:10]
            row.append(f"{n}: {t[0]:.1f}/{t[1]:.1f}({name})")
        print(f"{L}/{M}/{ntaps} {'vfo' if vfo else 'dec'} " + " | ".join(row), flush=True)
