#!/usr/bin/env python3
"""Regenerates tests/golden/vectors.npz from the CPU oracle (oracle/qdsp_oracle.c).

What these fixtures pin -- and what they do not: the reference (AlexandreRouma/qdsp) cannot
be built in this image (its arithmetic is in VOLK, which is absent; no stand-in allowed) and
ships no vectors of its own, so these files are produced by the repo's own restatement of
the reference algorithm with VOLK-generic accumulation order.  They freeze that restatement
(any later change to the oracle or the kernels shows up as a diff against them); the
hand-checkable known answers in kat.json are what ties the restatement to the reference's
equations.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402


def blocks_of(n, sizes):
    out, i, k = [], 0, 0
    while i < n:
        b = min(sizes[k % len(sizes)], n - i)
        out.append(b)
        i += b
        k += 1
    return out


def run_blocks(op, x, sizes):
    ys, i = [], 0
    for b in blocks_of(len(x), sizes):
        ys.append(op.process(x[i:i + b]))
        i += b
    return np.concatenate(ys) if ys else np.zeros(0, np.complex64)


def main():
    d = {}
    manifest = []
    N = 6000
    x = O.synth_iq(0, N, seed=1234)
    d["x"] = x
    xr = np.ascontiguousarray(x.real)

    # a5: tap tables
    n63 = O.blackman_tap_count(0.1, 4.0 / 63.0, 1.0)
    assert n63 == 63
    t63 = O.blackman_taps(0.1, 1.0, 63)
    d["taps63"] = t63
    d["taps256"] = O.lowpass_taps_f64(256, 1.0 / 16.0)
    d["taps4"] = np.array([0.1, 0.2, 0.3, 0.4], np.float32)
    d["taps_bp63"] = O.blackman_bandpass_taps(0.05, 0.2, 1.0, 63)
    d["taps_rrc31"] = O.rrc_taps(31, 4.0, 1.0, 0.35)

    # a1: FIR, N in {4, 63, 256}, ragged block sizes incl. blocks shorter than the history
    for name in ("taps4", "taps63", "taps256"):
        sizes = [1000, 37, 1, 2048, 5]
        d[f"fir_{name}"] = run_blocks(O.Fir(d[name]), x, sizes)
        d[f"firf32_{name}"] = run_blocks(O.Fir(d[name], complex_data=False), xr, sizes)
        manifest.append((f"fir_{name}", "FIR<complex_t>", sizes))

    # a2: resampler, (L, M) in {(1,2),(1,8),(2,1),(2,3)} + (3,7) and an odd block size
    for (L, M) in ((1, 2), (1, 8), (2, 1), (2, 3), (3, 7)):
        taps = (d["taps63"] * L).astype(np.float32)
        sizes = [1001, 64, 7, 2000]
        d[f"rs_{L}_{M}"] = run_blocks(O.Resampler(taps, L, M), x, sizes)
        manifest.append((f"rs_{L}_{M}", "PolyphaseResampler<complex_t>", sizes))
    d["rs_1_8_t256"] = run_blocks(O.Resampler(d["taps256"], 1, 8), x, [1001, 64, 7, 2000])
    d["rsf32_1_8"] = run_blocks(O.Resampler(d["taps63"], 1, 8, complex_data=False), xr, [1001, 64, 7, 2000])

    # a3: xlator, two frequencies, multi-block (VOLK-generic recursive phasor AND exact NCO)
    for i, (fs, f) in enumerate(((2.4e6, 123456.0), (48000.0, -7000.0))):
        sizes = [700, 512, 513, 1]
        d[f"xl{i}_generic"] = run_blocks(O.Xlator(fs, f), x, sizes)
        d[f"xl{i}_exact"] = run_blocks(O.Xlator(fs, f, exact=True), x, sizes)
        d[f"xl{i}_exact_vg"] = run_blocks(O.Xlator(fs, f, exact=True, volk_gain=True), x, sizes)
        d[f"xl{i}_delta"] = O.Xlator(fs, f).delta.copy()

    # a4: one VFO config (2.4 Msps -> 240 ksps, offset 300 kHz, bw 200 kHz)
    v = O.Vfo(300e3, 2.4e6, 240e3, 200e3, exact_nco=True)
    d["vfo_taps"] = v.taps
    d["vfo_ratio"] = np.array([v.interp, v.decim], np.int32)
    d["vfo_delta"] = v.xl.delta.copy()
    d["vfo_exact"] = run_blocks(v, x, [1000, 2000, 10, 2990])
    d["vfo_generic"] = run_blocks(O.Vfo(300e3, 2.4e6, 240e3, 200e3), x, [1000, 2000, 10, 2990])
    d["vfo_exact_vg"] = run_blocks(O.Vfo(300e3, 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True), x, [1000, 2000, 10, 2990])

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes;", len(d), "arrays")


if __name__ == "__main__":
    main()
