#!/usr/bin/env python3
"""Reference-sized calls (<= 1e6 samples, stream.h:7): per-call time of back-to-back launches beside the kernel's own
duration (run under `rocprofv3 --kernel-trace --stats` for the second figure) and a plain 8 MB device copy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from qdsp_amd import ops

taps = bench.lowpass_taps(256, 1.0 / 16.0)
taps63 = bench.lowpass_taps(63, 0.1)
t401 = bench.lowpass_taps(401, 0.4 / 50)
def _novg(v):
    v.set_volk_gain(False)
    return v


mk = {
    "fir256": lambda: ops.Fir(taps, max_block=0),
    "fir63": lambda: ops.Fir(taps63, max_block=0),
    "decim8/256": lambda: ops.Resampler(taps, 1, 8, max_block=0),
    "vfo8/256": lambda: ops.Vfo(taps, 1, 8, ops.phase_delta(1.0, 0.1234), max_block=0),
    "vfo8 ideal": lambda: _novg(ops.Vfo(taps, 1, 8, ops.phase_delta(1.0, 0.1234), max_block=0)),
    "vfo50/401": lambda: ops.Vfo(t401, 1, 50, ops.phase_delta(1.0, 0.1234), max_block=0),
    "xlate": lambda: ops.Xlator(phase_inc=ops.phase_delta(1.0, 0.1234), max_block=0),
}
sizes = [int(s) for s in sys.argv[1:]] or [65536, 1_000_000]
for n in sizes:
    x = ops.synth_iq(n, seed=1)
    out = torch.empty(n + 8, dtype=torch.complex64, device="cuda")
    for name, f in mk.items():
        op = f()
        op.process(x, out)
        torch.cuda.synchronize()
        ms = min(op.time_dev(x, out, 200) for _ in range(3))
        print(f"{name:11s} {n:8d}: {ms * 1e3:6.2f} us per call  {op.last_kernel()['name']}", flush=True)
    y = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        for _ in range(200):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
    print(f"{'torch copy':11s} {n:8d}: {e0.elapsed_time(e1) / 200 * 1e3:6.2f} us per call", flush=True)
