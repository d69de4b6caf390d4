#!/usr/bin/env python3
"""Kernel-geometry sweep on the GPU box: prints kernel ms / Gsamples/s for (R, NT) choices.
Usage: python tools/tune.py [workload ...]   (env QDSP_HIP_R / QDSP_HIP_NT are set per trial)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402

GEOMS = {
    "fir256": [(8, 256), (16, 256), (4, 256)],
    "fir63": [(8, 256), (16, 256), (4, 256)],
    "decim8": [(4, 128), (8, 128), (4, 256), (8, 256), (4, 64)],
    "xlate_fir_decim8": [(4, 128), (8, 128), (4, 256), (8, 256), (4, 64)],
}


def main():
    names = [a for a in sys.argv[1:] if a in GEOMS]
    if not sys.argv[1:]:
        names = list(GEOMS)
    n = 1 << 27
    x = ops.synth_iq(n, seed=1234)
    for name in names:
        w = bench.WORKLOADS[name]
        out = torch.empty(n // w["decim"], dtype=torch.complex64, device="cuda")
        for (r, nt) in GEOMS[name]:
            capi.setenv("QDSP_HIP_R", r); capi.setenv("QDSP_HIP_NT", nt)
            try:
                op = bench.make_op(ops, name, 0)
                op.process(x, out)
                torch.cuda.synchronize()
                ms = min(op.time_dev(x, out, 10) for _ in range(3))
                k = op.last_kernel()
                print(f"{name:18s} R={r:2d} NT={nt:3d} lds={k['lds_bytes']:6d} grid={k['grid']:6d}  {ms:8.4f} ms  "
                      f"{n / ms / 1e6:8.1f} Gs/s  {w['bytes'] * n / ms / 1e6:7.1f} GB/s  {w['flops'] * n / ms / 1e9:6.1f} TF", flush=True)
                op.close()
            except Exception as e:  # noqa: BLE001
                print(f"{name} R={r} NT={nt}: {e}", flush=True)
    capi.setenv("QDSP_HIP_R", None)
    capi.setenv("QDSP_HIP_NT", None)
    # any-decimation VFO (the reference's typical 2.4 MHz -> 240 kHz): direct vs overlap-save + strided store
    for (dec, ntaps) in ((10, 97), (10, 256), (3, 63), (5, 128), (2, 256), (64, 256)):
        taps = bench.lowpass_taps(ntaps, 0.4 / dec)
        out = torch.empty(n // dec + 1, dtype=torch.complex64, device="cuda")
        for mode in (1, 2):
            op = ops.Vfo(taps, 1, dec, ops.phase_delta(1.0, 0.1234), max_block=0)
            op.set_mode(mode)
            op.process(x, out)
            torch.cuda.synchronize()
            ms = min(op.time_dev(x, out, 10) for _ in range(3))
            print(f"vfo dec={dec:3d} ntaps={ntaps:4d} mode={'direct' if mode == 1 else 'fft':6s} {op.last_kernel()['name']:18s} {ms:8.4f} ms {n / ms / 1e6:8.1f} Gs/s", flush=True)
            op.close()
    # decimators: direct form vs overlap-save with pruned inverse
    for name in ("decim8", "xlate_fir_decim8"):
        w = bench.WORKLOADS[name]
        out = torch.empty(n // w["decim"], dtype=torch.complex64, device="cuda")
        for mode in (1, 2):
            op = bench.make_op(ops, name, 0)
            op.set_mode(mode)
            op.process(x, out)
            torch.cuda.synchronize()
            ms = min(op.time_dev(x, out, 10) for _ in range(3))
            print(f"{name:18s} mode={'direct' if mode == 1 else 'fft':6s} {ms:8.4f} ms {n / ms / 1e6:8.1f} Gs/s  {w['bytes'] * n / ms / 1e6:7.1f} GB/s", flush=True)
            op.close()
    # FIR algorithm: direct form vs overlap-save FFT, per tap count
    for ntaps in [int(a) for a in os.environ.get("TUNE_TAPS", "32,63,128,256,512,1024").split(",")]:
        taps = bench.lowpass_taps(ntaps, 1.0 / 16.0)
        out = torch.empty(n, dtype=torch.complex64, device="cuda")
        for mode, wg, nt in ((1, 0, 0), (2, 8, 0), (2, 16, 0)):
            capi.setenv("QDSP_HIP_FFT_NT", str(nt))
            if wg:
                capi.setenv("QDSP_HIP_FFT_WG_PER_CU", str(wg))
            op = ops.Fir(taps, max_block=0)
            op.set_mode(mode)
            op.process(x, out)
            torch.cuda.synchronize()
            ms = min(op.time_dev(x, out, 10) for _ in range(3))
            print(f"fir ntaps={ntaps:5d} mode={'direct' if mode == 1 else 'fft wg/cu=' + str(wg) + ' nt=' + str(nt):18s} {ms:8.4f} ms "
                  f"{n / ms / 1e6:8.1f} Gs/s  {16 * n / ms / 1e6:7.1f} GB/s", flush=True)
            op.close()


if __name__ == "__main__":
    main()
