"""Non-uniform channel plans (Splitter -> N x VFO at arbitrary offsets) per reference-sized block: one batched launch
(decim_mfma_batch_kernel where the design allows, resamp_any_batch_kernel otherwise and with QDSP_HIP_NO_MF_BATCH)
against one fused kernel per channel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import lowpass_taps
from qdsp_amd import capi, ops

for (M, ntaps, nch) in ((50, 401, 4), (50, 401, 16), (50, 401, 64), (10, 97, 16), (8, 256, 16), (64, 256, 64)):
    taps = lowpass_taps(ntaps, 0.4 / M)
    incs = [ops.phase_delta(1.0, -0.45 + 0.9 * (i + 0.37) / nch) for i in range(nch)]
    for n in (65_536, 1_000_000):
        n = n // M * M
        x = ops.synth_iq(n, seed=1, device=0)
        out = torch.empty((nch, n // M), dtype=torch.complex64, device="cuda")
        row = [f"M {M:3d} taps {ntaps:4d} ch {nch:3d} n {n:8d}"]
        for batch in (2, 1, 0):
            capi.setenv("QDSP_HIP_NO_CHAN_BATCH", "0" if batch else "1")
            capi.setenv("QDSP_HIP_NO_MF_BATCH", "0" if batch == 2 else "1")
            ch = ops.Channelizer(taps, 1, M, incs, max_block=0)
            for _ in range(10):
                ch.process(x, out)
            torch.cuda.synchronize()
            us = min(ch.time_dev(x, out, 30) for _ in range(3)) * 1e3
            row.append(f"{ch.last_kernel()['name'][:23]:23s} {us:8.1f} us")
            ch.close()
        print("  ".join(row), flush=True)
