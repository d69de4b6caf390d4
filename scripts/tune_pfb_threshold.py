"""Where the one-wave-per-segment polyphase kernel (pfb_dec8_kernel) overtakes the per-segment / grouped 4096-point
kernels, by call size (decimate by 8, 256 taps): sets QDSP_HIP_PFB_MIN_COUNT."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import lowpass_taps
from qdsp_amd import capi, ops

taps = lowpass_taps(256, 1 / 16)
inc = ops.phase_delta(1.0, 0.1234)
for log2n in (16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27):
    n = 1 << log2n
    x = ops.synth_iq(n, seed=1, device=0)
    out = torch.empty(n // 8, dtype=torch.complex64, device="cuda")
    row = [f"2^{log2n}"]
    for rot in (False, True):
        for pfb in (0, 1):
            capi.setenv("QDSP_HIP_NO_PFB", "0" if pfb else "1")
            capi.setenv("QDSP_HIP_PFB_MIN_COUNT", "0")
            op = ops.Vfo(taps, 1, 8, inc, max_block=0) if rot else ops.Resampler(taps, 1, 8, max_block=0)
            op.set_mode(op.FFT)
            for _ in range(20):
                op.process(x, out)
            torch.cuda.synchronize()
            us = min(op.time_dev(x, out, 50) for _ in range(3)) * 1e3
            row.append(f"{'vfo' if rot else 'dec'} {op.last_kernel()['name'][:8]} {us:8.1f} us")
            op.close()
    print("  ".join(row), flush=True)
