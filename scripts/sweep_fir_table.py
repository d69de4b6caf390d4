#!/usr/bin/env python3
"""FIR<complex_t> per call: the four kernel families the AUTO choice picks from, each forced with QDSP_HIP_FIR_PICK, on the grid the
dispatch table is built on (scripts/gen_dispatch_table.py -> qdsp_amd/csrc/dispatch_table.inc).

    python scripts/sweep_fir_table.py > profiles/r04_sweep_fir_table.txt         (on the GPU box, ~1 minute)

Line format:  <log2 count> <ntaps> <us lat> <us core> <us fft1k> <us fft4k>     ("-" = the family cannot serve the shape)
Families: 1 = fir_lat_kernel (direct form arranged for latency), 2 = fir_core_kernel (direct form), 3 = fir_fft1k_kernel (overlap-save,
one wave per 1024-point segment), 4 = fir_fft_dma_kernel / fir_fft_kernel (4096-point overlap-save)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402

LOG2 = list(range(10, 28))
TAPS = [8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024]
EXPECT = {1: ("fir_lat_kernel",), 2: ("fir_core_kernel",), 3: ("fir_fft1k_kernel",), 4: ("fir_fft_dma_kernel", "fir_fft_kernel", "fir_fft_dmapk_kernel")}


def timed(taps, x, out, pick):
    capi.setenv("QDSP_HIP_FIR_PICK", str(pick))
    op = ops.Fir(taps, max_block=0)
    try:
        op.process(x, out)
        if op.last_kernel()["name"] not in EXPECT[pick]:
            return None                      # the family declined the shape: the call went elsewhere
        n = x.numel()
        # the direct forms cost n * taps: keep a timing batch under ~20 ms
        reps = max(3, min(100, int(2e-2 / max(2e-6, n * max(1, len(taps) if pick <= 2 else 8) * 2.5e-13))))
        op.time_dev(x, out, max(3, reps // 4))
        return min(op.time_dev(x, out, reps) for _ in range(4))
    finally:
        op.close()
        capi.setenv("QDSP_HIP_FIR_PICK", None)


def main():
    print("# scripts/sweep_fir_table.py: FIR<complex_t>, microseconds per call (min of 4 batches), families forced with QDSP_HIP_FIR_PICK = 1 lat, 2 core, 3 fft1k, 4 fft4k")
    print("# log2n taps lat core fft1k fft4k")
    for lg in LOG2:
        x = ops.synth_iq(1 << lg, seed=3)
        out = torch.empty(x.numel() + 8, dtype=torch.complex64, device="cuda")
        for nt in TAPS:
            taps = bench.lowpass_taps(nt, 0.2)
            cells = []
            for pick in (1, 2, 3, 4):
                # (skip what cannot win and would take seconds: direct forms on 2^24+ samples x 512+ taps)
                if pick <= 2 and (1 << lg) * nt > (1 << 32):
                    cells.append(None)
                    continue
                cells.append(timed(taps, x, out, pick))
            print(f"{lg} {nt} " + " ".join("-" if c is None else f"{c * 1000:.2f}" for c in cells), flush=True)


if __name__ == "__main__":
    main()
