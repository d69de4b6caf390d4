#!/usr/bin/env python3
"""Does the distance between the input and the output buffer matter?  Both are large power-of-two-aligned allocations by default, so a
kernel that reads position p and writes position ~p at the same time sends both to the same memory channel / bank.  Kernel ms per
2^27 samples with the output placed `off` samples (8 bytes each) further into its allocation, interleaved rounds in one process.

    python scripts/tune_out_offset.py fir256 [decim8 xlate ...]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import ops  # noqa: E402

OFFS = (0, 32, 96, 160, 544, 2080, 8224, 32800, 131104, 524320, 2097184)   # (samples; odd multiples of 32 = 256 bytes)


def main():
    n = 1 << 27
    x = ops.synth_iq(n, seed=1234)
    for name in sys.argv[1:] or ["fir256"]:
        w = bench.WORKLOADS[name]
        nout = n // w["decim"] * w.get("interp", 1)
        big = torch.empty(nout + max(OFFS) + 64, dtype=torch.complex64, device="cuda")
        op = bench.make_op(ops, name, 0)
        outs = {o: big[o:o + nout] for o in OFFS}
        for _ in range(5):
            op.time_dev(x, outs[0], 20)
        t = {o: [] for o in OFFS}
        for _ in range(5):
            for o in OFFS:
                op.process(x, outs[o])
                t[o].append(op.time_dev(x, outs[o], 10))
        print(f"{name} ({op.last_kernel()['name']}): " + "  ".join(f"+{o}: {statistics.median(t[o]):.4f}" for o in OFFS), flush=True)
        op.close()


if __name__ == "__main__":
    main()
