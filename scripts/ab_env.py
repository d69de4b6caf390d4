#!/usr/bin/env python3
"""A/B/A of library knobs in ONE process (cdna_hip_programming.md rule 24): N settings x M interleaved rounds on one
bench workload, median and minimum of the kernel time per setting.

    python scripts/ab_env.py fir256 "" QDSP_HIP_FFT_NT=0 "QDSP_HIP_FFT_NT=3,QDSP_HIP_FFT_WG_PER_CU=4" [--rounds 7] [--log2n 27]

Each setting is a comma-separated list of NAME=VALUE pairs ("" = the defaults)."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402


def apply(setting, names):
    for n in names:
        os.environ.pop(n, None)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    if hasattr(capi, "reload_env"):
        capi.reload_env()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--log2n", type=int, default=27)
    ap.add_argument("--zeros", action="store_true", help="all-zero input (DVFS check: the chip holds a higher clock on zeros)")
    a = ap.parse_args()
    names = sorted({kv.split("=")[0] for s in a.settings for kv in filter(None, s.split(","))})
    n = 1 << a.log2n
    w = bench.WORKLOADS[a.workload]
    if w.get("interp", 1) > 1:
        n -= n % w["decim"]
    x = ops.synth_iq(n, seed=1234)
    if w.get("real", False):
        x = torch.view_as_real(x)[:, 0].contiguous()
    if a.zeros:
        x.zero_()
    nout = n // w["decim"] * w.get("interp", 1)
    out = torch.empty((w["nchan"], nout + bench.CHAN_ROW_PAD) if "nchan" in w else nout + 8, dtype=torch.float32 if w.get("real", False) else torch.complex64, device="cuda")
    op = bench.make_op(ops, a.workload, 0)
    apply(a.settings[0], names)
    for _ in range(5):   # settle the clocks
        op.time_dev(x, out, 20)
    times = {s: [] for s in a.settings}
    kern = {}
    for _ in range(a.rounds):
        for s in a.settings:
            apply(s, names)
            op.process(x, out)
            times[s].append(op.time_dev(x, out, a.iters))
            kern[s] = op.last_kernel()["name"]
    apply("", names)
    for s in a.settings:
        t = times[s]
        med = statistics.median(t)
        print(f"{a.workload:18s} {s or '(defaults)':60s} {kern[s]:20s} median {med:.4f} ms  min {min(t):.4f}  max {max(t):.4f}  "
              f"{w['bytes'] * n / med / 1e6:7.1f} GB/s algorithmic = {w['bytes'] * n / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    op.close()


if __name__ == "__main__":
    main()
