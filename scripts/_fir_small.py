import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import lowpass_taps
from qdsp_amd import ops
for ntaps in (63, 127, 256):
    taps = lowpass_taps(ntaps, 1 / 16)
    for n in (16384, 65536, 262144, 1_000_000):
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
        out = torch.empty(n + 8, dtype=torch.complex64, device="cuda")
        row = []
        for mode in ("AUTO", "DIRECT", "FFT"):
            op = ops.Fir(taps, max_block=0)
            if mode != "AUTO": op.set_mode(getattr(op, mode))
            for _ in range(10): op.process(x, out=out)
            torch.cuda.synchronize()
            us = min(op.time_dev(x, out, 50) for _ in range(3)) * 1e3
            row.append(f"{mode} {op.last_kernel()['name'][:12]} {us:.1f}")
        print(ntaps, n, " | ".join(row), flush=True)
