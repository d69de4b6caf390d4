#!/usr/bin/env python3
"""What the chip-filling decimators next to decimate-by-8 cost today ("pfb_dec8_kernel's form for
decimations 4 and 16"): decimations 2 / 4 / 16 / 32 at 128 / 256 / 512 taps, plain and fused with the NCO, 2^27 samples per call,
under the library's own dispatch and with the MFMA decimator switched off (QDSP_HIP_NO_MF=1) where it is the default.

    python scripts/tune_dec_small.py [--log2n 27] > profiles/r03_tune_dec_small.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from qdsp_amd import capi, ops  # noqa: E402


def lowpass(ntaps, fc):
    n = np.arange(ntaps) - (ntaps - 1) / 2.0
    return (2 * fc * np.sinc(2 * fc * n) * np.blackman(ntaps)).astype(np.float32)


def run(x, M, ntaps, nco, setting, real=False):
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        capi.setenv(k, v)
    taps = lowpass(ntaps, 0.45 / M)
    if real:
        op = ops.Resampler(taps, 1, M, complex_data=False, max_block=0) if M > 1 else ops.Fir(taps, complex_data=False, max_block=0)
    else:
        op = ops.Vfo(taps, 1, M, ops.phase_delta(1.0, 0.1234), max_block=0) if nco else ops.Resampler(taps, 1, M, max_block=0)
    out = torch.empty(x.numel() // M + 8, dtype=x.dtype, device="cuda")
    op.process(x, out)
    for _ in range(3):
        op.time_dev(x, out, 20)
    t = min(op.time_dev(x, out, 20) for _ in range(5))
    name = op.last_kernel()["name"]
    op.close()
    for kv in filter(None, setting.split(",")):
        capi.setenv(kv.split("=")[0], None)
    return t, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=27)
    ap.add_argument("--decims", default="2,4,16,32")
    ap.add_argument("--taps", default="128,256,512")
    ap.add_argument("--alts", default="QDSP_HIP_NO_MF=1", help="semicolon-separated alternative settings next to the defaults")
    ap.add_argument("--no-nco", action="store_true")
    ap.add_argument("--real", action="store_true", help="PolyphaseResampler<float> (4 + 4/M bytes per sample)")
    a = ap.parse_args()
    n = 1 << a.log2n
    x = torch.randn(n, device="cuda") if a.real else ops.synth_iq(n, seed=7)
    print(f"# kernel ms per 2^{a.log2n} input samples (min of 5 x 20 launches), algorithmic bytes (8 + 8/M) per sample, fraction of 8 TB/s")
    for M in [int(v) for v in a.decims.split(",")]:
        for ntaps in [int(v) for v in a.taps.split(",")]:
            for nco in ((False,) if a.no_nco or a.real else (False, True)):
                row = []
                for setting in [""] + a.alts.split(";"):
                    t, name = run(x, M, ntaps, nco, setting, a.real)
                    row.append((t, name, setting))
                cells = "   ".join(f"{s or 'default':18s} {nm:20s} {t:.4f} ms = {(0.5 if a.real else 1.0) * (8 + 8 / M) * n / t / 1e6 / 8000:.3f}" for t, nm, s in row)
                print(f"decimate by {M:2d}, {ntaps:3d} taps{', NCO' if nco else '     '}: {cells}", flush=True)


if __name__ == "__main__":
    main()
