#!/usr/bin/env python3
"""Reference-sized calls of rational resamplers: resamp_lm_kernel (R x L accumulators per lane over all taps: the throughput
form) against resamp_any_kernel with call-sized tiles, and what AUTO picks (lm_yields_to_any, any_plan)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qdsp_amd import capi, ops

def t(mk, x, out, env, iters):
    for k, v in env.items(): capi.setenv(k, v)
    op = mk(); op.process(x, out); torch.cuda.synchronize()
    us = min(op.time_dev(x, out, iters) for _ in range(3)) * 1e3
    nm = op.last_kernel()["name"]; op.close()
    for k in env: capi.setenv(k, None)
    return us, nm

shapes = ((3, 7, 200), (2, 3, 64), (5, 2, 81), (10, 1, 160), (2, 1, 63), (3, 2, 100), (4, 5, 127), (10, 7, 400), (2, 1, 15), (5, 8, 640), (3, 1, 31), (24, 125, 1001), (147, 160, 2048))
for (L, M, nt) in shapes:
    for vfo in (False, True):
        for n in (16384, 65536, 262144, 1_000_000, 4 << 20):
            nn = n - n % M
            x = ops.synth_iq(nn, seed=1); out = torch.empty(nn * L // M + 8, dtype=torch.complex64, device="cuda")
            taps = bench.lowpass_taps(nt, 0.4 / max(L, M)) * L
            mk = (lambda: ops.Vfo(taps, L, M, ops.phase_delta(1.0, 0.1234), max_block=0)) if vfo else (lambda: ops.Resampler(taps, L, M, max_block=0))
            it = 100 if n <= 1_000_000 else 20
            a = t(mk, x, out, {"QDSP_HIP_NO_RM": "1", "QDSP_HIP_NO_LM_SMALL_CALL_RULE": "1", "QDSP_HIP_ANY_SMALL_CALL_TILES": "0"}, it)
            b = t(mk, x, out, {"QDSP_HIP_NO_LM": "1", "QDSP_HIP_NO_RM": "1"}, it)
            d = t(mk, x, out, {}, it)
            print(f"{L}/{M} {nt} taps vfo={int(vfo)} n={nn:8d}: round-1 choice {a[0]:6.1f} us {a[1][:11]:11s} | general kernel, call-sized tiles {b[0]:6.1f} | AUTO {d[0]:6.1f} {d[1][:12]}", flush=True)
