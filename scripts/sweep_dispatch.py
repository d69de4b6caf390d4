#!/usr/bin/env python3
"""Is the default dispatch the fastest one?  Chip-filling calls (2^26 input samples) of FIRs and rational resamplers under the library's
own rules and with each kernel family switched off in turn; rows where an alternative beats the default by more than 4 % are marked.

    python scripts/sweep_dispatch.py [fir] [rational] > profiles/r03_sweep_dispatch.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402

N = 1 << 26


def timed(make, x, nout_of, setting):
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        capi.setenv(k, v)
    op = make()
    out = torch.empty(nout_of(x.numel()) + 64, dtype=x.dtype, device="cuda")
    op.process(x, out)
    for _ in range(2):
        op.time_dev(x, out, 10)
    t = min(op.time_dev(x, out, 10) for _ in range(4))
    name = op.last_kernel()["name"]
    op.close()
    for kv in filter(None, setting.split(",")):
        capi.setenv(kv.split("=")[0], None)
    return t, name


def report(label, cells):
    t0 = cells[0][1]
    best = min(c[1] for c in cells)
    mark = "  <-- an alternative is %.0f %% faster" % (100 * (1 - best / t0)) if best < 0.96 * t0 else ""
    print(f"{label:34s} " + "   ".join(f"{s or 'default'}: {nm[:22]} {t:.4f}" for s, t, nm in cells) + mark, flush=True)


def main():
    what = sys.argv[1:] or ["fir", "rational"]
    x = ops.synth_iq(N, seed=5)
    if "fir" in what:
        for ntaps in (8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 512, 1024):
            taps = bench.lowpass_taps(ntaps, 0.2)
            cells = []
            for s in ("", "QDSP_HIP_FIR_MODE=1", "QDSP_HIP_FIR_MODE=2", "QDSP_HIP_FFT_DMA=0"):
                t, nm = timed(lambda: ops.Fir(taps, max_block=0), x, lambda n: n, s)
                if s and nm == cells[0][2] and s != "QDSP_HIP_FFT_DMA=0":
                    continue
                cells.append((s, t, nm))
            report(f"FIR<complex_t> {ntaps} taps", cells)
        xr = torch.randn(N, device="cuda")
        for ntaps in (16, 32, 64, 96, 128, 256, 512):
            taps = bench.lowpass_taps(ntaps, 0.2)
            cells = []
            for s in ("", "QDSP_HIP_FIR_MODE=1", "QDSP_HIP_FIR_MODE=2", "QDSP_HIP_NO_FFT1K_REAL=1"):
                t, nm = timed(lambda: ops.Fir(taps, complex_data=False, max_block=0), xr, lambda n: n, s)
                if s and nm == cells[0][2] and "FFT1K" not in s:
                    continue
                cells.append((s, t, nm))
            report(f"FIR<float> {ntaps} taps", cells)
    if "rational" in what:
        for L, M, tpp in ((2, 1, 16), (3, 1, 16), (4, 1, 12), (6, 1, 10), (3, 2, 20), (4, 3, 20), (5, 3, 20), (5, 4, 20), (7, 5, 24), (10, 7, 16), (12, 5, 12), (16, 15, 16),
                          (25, 24, 8), (2, 3, 20), (3, 4, 20), (4, 5, 32), (3, 8, 20), (5, 6, 20), (5, 8, 20), (10, 3, 16), (24, 125, 12), (48, 50, 20),
                          (147, 160, 16), (160, 147, 16), (441, 480, 8)):
            taps = (bench.lowpass_taps(L * tpp - 3, 0.4 / max(L, M)) * L).astype(np.float32)
            nin = N if L <= M else int(N * M / L)
            nin -= nin % M
            xin = x[:nin]
            for nco in (False, True):
                mk = (lambda: ops.Vfo(taps, L, M, ops.phase_delta(1.0, 0.1234), max_block=0)) if nco else (lambda: ops.Resampler(taps, L, M, max_block=0))
                cells = []
                for s in ("", "QDSP_HIP_NO_RM=1", "QDSP_HIP_NO_LM=1", "QDSP_HIP_FORCE_ANY=1", "QDSP_HIP_RM_MIN_INTERP=2"):
                    t, nm = timed(mk, xin, lambda n: n // M * L, s)
                    if s and any(nm == c[2] for c in cells):
                        continue
                    cells.append((s, t, nm))
                report(f"{L}/{M} {tpp} taps per phase{' NCO' if nco else ''}", cells)


if __name__ == "__main__":
    main()
