#!/usr/bin/env python3
"""Where a wave of fir_fft_dmapk_kernel spends its time: the diagnostic build of the kernel (template parameter STAMPS) sums
s_memtime ticks per phase of its segment loop; this script runs it on the bench workload and prints the share of each phase.

    python scripts/stamp_fir_fft.py [--log2n 27]

(Needs the diagnostic build of the library: make -C qdsp_amd/csrc -B DIAG=1.)
The device buffer is handed over through QDSP_HIP_FFT_STAMPS=<device pointer> (read by launch_fft only for this purpose)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402

PHASES = ["wait for DMA (vmcnt)", "read raw + pass A + write (in place)", "barrier 1", "read + pass B", "barrier 2", "write layout 2", "barrier 3",
          "read + pass C, x Hf, pass C'", "barrier 4", "write layout 2", "barrier 5", "read + pass B'", "barrier 6", "write layout 1", "barrier 7",
          "read, request next DMA, pass A', stores, loop"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=27)
    a = ap.parse_args()
    n = 1 << a.log2n
    os.environ["QDSP_HIP_FFT_DMA"] = "2"
    x = ops.synth_iq(n, seed=1234)
    out = torch.empty(n, dtype=torch.complex64, device="cuda")
    op = bench.make_op(ops, "fir256", 0)
    if hasattr(capi, "reload_env"):
        capi.reload_env()
    for _ in range(3):
        op.time_dev(x, out, 20)
    grid = op.last_kernel()["grid"]
    stamps = torch.zeros(grid * 16, dtype=torch.int32, device="cuda")
    os.environ["QDSP_HIP_FFT_STAMPS"] = str(stamps.data_ptr())
    if hasattr(capi, "reload_env"):
        capi.reload_env()
    ms = op.time_dev(x, out, 1)
    torch.cuda.synchronize()
    os.environ.pop("QDSP_HIP_FFT_STAMPS")
    s = stamps.view(grid, 16)[: grid - 1].to(torch.float64).cpu()      # (the last workgroup hands over the history)
    tot = s.sum(dim=1)
    print(f"diagnostic launch {ms:.4f} ms, grid {grid}; mean ticks per workgroup-wave {tot.mean():.0f} (s_memtime ticks = shader cycles)")
    order = [15] + list(range(15))
    names = {15: PHASES[15]}
    names.update({i: PHASES[i] for i in range(15)})
    for i in [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]:
        print(f"  {names[i]:48s} {100.0 * s[:, i].sum() / tot.sum():5.1f} %")
    op.close()


if __name__ == "__main__":
    main()
