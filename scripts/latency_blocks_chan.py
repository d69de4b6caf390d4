#!/usr/bin/env python3
"""Reference-sized calls of the channelizer (chan_cf32): the uniform 64-channel plans and non-uniform banks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qdsp_amd import ops

sizes = [int(s) for s in sys.argv[1:]] or [16384, 65536, 262144, 1_000_000]
def run(name, mk, nch, M):
    row = [f"{name:28s}"]
    for n in sizes:
        nn = n - n % M
        x = ops.synth_iq(nn, seed=1)
        out = torch.empty((nch, nn // M), dtype=torch.complex64, device="cuda")
        ch = mk()
        for _ in range(3): ch.process(x, out)
        torch.cuda.synchronize()
        us = min(ch.time_dev(x, out, 100) for _ in range(3)) * 1e3
        row.append(f"{nn:8d}: {us:6.1f} us {ch.last_kernel()['name'][:22]:22s}")
        ch.close()
    print(" | ".join(row), flush=True)

for M in (64, 8):
    incs = [ops.phase_delta(1.0, -(c - 31.5) / 64.0) for c in range(64)]
    run(f"uniform 64 ch, decim {M}", lambda: ops.Channelizer(bench.lowpass_taps(256, 1 / 128), 1, M, incs, max_block=0), 64, M)
for (M, ntaps, nch) in ((50, 401, 4), (50, 401, 16), (10, 97, 16), (8, 256, 16), (64, 256, 64)):
    incs = [ops.phase_delta(1.0, -0.45 + 0.9 * (i + 0.37) / nch) for i in range(nch)]
    run(f"bank {nch} ch, {ntaps} taps / {M}", lambda: ops.Channelizer(bench.lowpass_taps(ntaps, 0.4 / M), 1, M, incs, max_block=0), nch, M)
