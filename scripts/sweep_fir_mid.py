#!/usr/bin/env python3
"""FIR<complex_t> on mid-sized calls: the library's dispatch against fir_lat / fir_fft1k / overlap-save / direct forced (profiles/r03_sweep_fir_mid.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from qdsp_amd import capi, ops
def timed(taps, x, setting):
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("="); capi.setenv(k, v)
    op = ops.Fir(taps, max_block=0); out = torch.empty(x.numel()+8, dtype=torch.complex64, device="cuda")
    op.process(x, out); op.time_dev(x, out, 20)
    t = min(op.time_dev(x, out, max(10, min(200, (1<<26)//x.numel()))) for _ in range(4)); nm = op.last_kernel()["name"]; op.close()
    for kv in filter(None, setting.split(",")): capi.setenv(kv.split("=")[0], None)
    return t, nm
for log2n in (18, 20, 22, 23, 24):
    x = ops.synth_iq(1 << log2n, seed=3)
    for ntaps in (16, 32, 64, 128, 256, 512):
        taps = bench.lowpass_taps(ntaps, 0.2)
        cells = []
        for s in ("", "QDSP_HIP_NO_FFT1K=1", "QDSP_HIP_NO_FIR_LAT=1", "QDSP_HIP_FIR_MODE=2", "QDSP_HIP_FIR_MODE=1"):
            t, nm = timed(taps, x, s)
            if s and any(nm == c[2] for c in cells): continue
            cells.append((s, t, nm))
        best = min(c[1] for c in cells)
        print(f"2^{log2n} {ntaps:4d} taps: " + "  ".join(f"{(s or 'default')[9:] or 'default'}:{nm[:11]} {t*1000:.1f}us" for s, t, nm in cells) + ("   <-- %.0f %%" % (100*(1-best/cells[0][1])) if best < 0.93*cells[0][1] else ""), flush=True)
