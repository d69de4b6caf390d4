"""CPU tests of the oracle (oracle/qdsp_oracle.c): known answers of SURVEY section 8a, the
committed golden vectors, and internal consistency between accumulation variants."""
import json
import os

import numpy as np
import pytest

import oracle as O
from conftest import rel_rms

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "vectors.npz"))


def run_blocks(op, x, sizes):
    ys, i, k = [], 0, 0
    while i < len(x):
        b = min(sizes[k % len(sizes)], len(x) - i)
        ys.append(op.process(x[i:i + b]))
        i += b
        k += 1
    return np.concatenate(ys)


def test_kat_fir(kat):
    for c in kat["fir"]:
        x = np.array(c["x_re"], np.float32)
        x = (x - 1j * x).astype(np.complex64)
        for acc in (O.ACC_F32, O.ACC_FMA, O.ACC_F64):
            y = run_blocks(O.Fir(c["taps"], acc=acc), x, c["blocks"])
            assert np.allclose(y.real, c["y_re"], atol=c["tol"] * 16)
            assert np.array_equal(y.imag, -y.real)


def test_kat_resampler(kat):
    for c in kat["resamp"]:
        x = np.array(c["x_re"], np.float32).astype(np.complex64)
        for acc in (O.ACC_F32, O.ACC_FMA, O.ACC_F64):
            y = run_blocks(O.Resampler(np.array(c["taps"], np.float32), c["interp"], c["decim"], acc=acc), x, c["blocks"])
            assert np.array_equal(y.real, np.array(c["y_re"], np.float32)), (c, y.real)  # small integers: exact


def test_kat_xlator(kat):
    for c in kat["xlator"]:
        for exact in (False, True):
            y = O.Xlator(c["sample_rate"], c["freq"], exact=exact).process(np.ones(c["n"], np.complex64))
            want = np.array([complex(a, b) for a, b in c["y"]])
            assert np.abs(y - want).max() < c["tol"]


def test_kat_blackman(kat):
    for c in kat["blackman"]:
        n = O.blackman_tap_count(c["cutoff"], c["trans_width"], c["sample_rate"])
        assert n == c["tap_count"]
        t = O.blackman_taps(c["cutoff"], c["sample_rate"], n)
        assert abs(t[0] - c["t0"]) < c["tol"] and abs(t[31] - c["t31"]) < c["tol"] and abs(t[62] - c["t62"]) < c["tol"]
        assert abs(t.sum() - 1.0) < 1e-5
        # plain truncated sinc, NOT symmetric (centre at tc/2 = 31.5)
        assert abs(t[0] - t[62]) > 1e-3


def test_blackman_even_count_is_nan():
    # i - tc/2 == 0 at i = 128 for 256 taps -> 0/0
    t = O.blackman_taps(0.1, 1.0, 256)
    assert np.isnan(t).any()
    with pytest.raises(ValueError):
        O.rrc_taps(256, 4.0, 1.0, 0.35)


def test_golden_vectors_reproduce(gold):
    """The committed fixtures are what the oracle produces today (bit for bit)."""
    x = gold["x"]
    xr = np.ascontiguousarray(x.real)
    assert np.array_equal(O.synth_iq(0, len(x), 1234), x)
    assert np.array_equal(O.blackman_taps(0.1, 1.0, 63), gold["taps63"])
    assert np.array_equal(O.lowpass_taps_f64(256, 1 / 16), gold["taps256"])
    assert np.array_equal(O.blackman_bandpass_taps(0.05, 0.2, 1.0, 63), gold["taps_bp63"])
    assert np.array_equal(O.rrc_taps(31, 4.0, 1.0, 0.35), gold["taps_rrc31"])
    for name in ("taps4", "taps63", "taps256"):
        assert np.array_equal(run_blocks(O.Fir(gold[name]), x, [1000, 37, 1, 2048, 5]), gold[f"fir_{name}"])
        assert np.array_equal(run_blocks(O.Fir(gold[name], complex_data=False), xr, [1000, 37, 1, 2048, 5]), gold[f"firf32_{name}"])
    for (L, M) in ((1, 2), (1, 8), (2, 1), (2, 3), (3, 7)):
        taps = (gold["taps63"] * L).astype(np.float32)
        assert np.array_equal(run_blocks(O.Resampler(taps, L, M), x, [1001, 64, 7, 2000]), gold[f"rs_{L}_{M}"])
    v = O.Vfo(300e3, 2.4e6, 240e3, 200e3, exact_nco=True)
    assert np.array_equal(v.taps, gold["vfo_taps"]) and [v.interp, v.decim] == list(gold["vfo_ratio"])
    assert np.array_equal(run_blocks(v, x, [1000, 2000, 10, 2990]), gold["vfo_exact"])


def test_fir_block_split_invariance(gold):
    """History carry-over is exact (filter.h:71): any block split gives identical output."""
    x = gold["x"]
    for name in ("taps4", "taps63", "taps256"):
        whole = O.Fir(gold[name]).process(x)
        assert np.array_equal(whole, gold[f"fir_{name}"])
        assert np.array_equal(run_blocks(O.Fir(gold[name]), x, [1]), whole) if name == "taps4" else True


def test_fir_is_convolution(gold):
    """y = lfilter(taps[::-1], x): newest sample pairs with taps[N-1]."""
    x = gold["x"].astype(np.complex128)
    for name in ("taps63", "taps256"):
        h = gold[name].astype(np.float64)
        want = np.convolve(x, h[::-1])[: len(x)]
        assert rel_rms(gold[f"fir_{name}"], want) < 1e-6
        got64 = O.Fir(gold[name], acc=O.ACC_F64).process(gold["x"])
        assert rel_rms(got64, want) < 1e-7


def test_accumulation_variants_agree(gold):
    x = gold["x"]
    h = gold["taps256"]
    y32, yfma, y64 = (O.Fir(h, acc=a).process(x) for a in (O.ACC_F32, O.ACC_FMA, O.ACC_F64))
    assert rel_rms(y32, y64) < 1e-6 and rel_rms(yfma, y64) < 1e-6
    r32, r64 = (O.Resampler(h, 1, 8, acc=a).process(x) for a in (O.ACC_F32, O.ACC_F64))
    assert rel_rms(r32, r64) < 1e-6


def test_resampler_phase_restart_skips_samples():
    """i restarts at 0 each block, so count*L % M != 0 drops input samples."""
    x = np.arange(1, 16, dtype=np.float32).astype(np.complex64)
    a = run_blocks(O.Resampler(np.array([1, 2, 3, 4], np.float32), 1, 2), x, [5])
    b = O.Resampler(np.array([1, 2, 3, 4], np.float32), 1, 2).process(x)
    assert len(a) == 6 and len(b) == 7 and not np.array_equal(a, b[:6])


def test_resampler_build_phases():
    ph = O.build_phases(np.arange(1, 8, dtype=np.float32), 3)  # 7 taps, L=3 -> P=3
    # tapPhases[(L-1)-p][t] = taps[t*L + p], zero padded (resampling.h:155-165)
    assert ph.shape == (3, 3)
    assert np.array_equal(ph[2], [1, 4, 7]) and np.array_equal(ph[1], [2, 5, 0]) and np.array_equal(ph[0], [3, 6, 0])
    assert O.resamp_ratio(2.4e6, 240e3) == (1, 10)
    assert O.resamp_ratio(48000.0, 44100.0) == (147, 160)


def test_rotator_drift_vs_exact(gold):
    """SURVEY H2.  The generic rotator differs from the FP64-phase NCO in two ways:
    (1) a deterministic magnitude sawtooth |inc|^(n mod 512) (~1e-5), reproduced by the
        volk_gain yardstick to float rounding on short streams;
    (2) a phase drift of the float recursion that grows with stream length and that no
        parallel NCO can (or should) reproduce -- it bounds how long a parity stream may be."""
    for fs, f in ((2.4e6, 123456.0), (48000.0, -7000.0)):
        x = np.ones(2000, np.complex64)
        g = O.Xlator(fs, f).process(x)
        ideal = O.Xlator(fs, f, exact=True).process(x)
        vg = O.Xlator(fs, f, exact=True, volk_gain=True).process(x)
        assert np.abs(g - ideal).max() < 3e-5           # sawtooth visible ...
        assert np.abs(g - vg).max() < 3e-6              # ... and explained
        assert np.abs(np.abs(g) - 1).max() < 4e-5       # |phase| held by the renormalisation
    x = np.ones(1_000_000, np.complex64)
    g = O.Xlator(48000.0, -7000.0).process(x)
    vg = O.Xlator(48000.0, -7000.0, exact=True, volk_gain=True).process(x)
    drift = np.abs(g - vg)
    assert drift[:4096].max() < 1e-5 and 1e-3 < drift.max() < 5e-2  # the reference's own drift


def test_vfo_design_matches_reference_formulas():
    L, M, taps = O.vfo_design(2.4e6, 240e3, 200e3)
    assert (L, M) == (1, 10)
    # realCutoff = 100k, design rate = 2.4M * 1 -> N = int(4 / (100e3/2.4e6)) = 96 -> odd -> 97
    assert len(taps) == 97
    assert abs(taps.sum() - 1.0) < 1e-5


def test_empty_and_tiny_blocks(gold):
    f = O.Fir(gold["taps63"])
    assert len(f.process(np.zeros(0, np.complex64))) == 0
    y = np.concatenate([f.process(gold["x"][i:i + 1]) for i in range(100)])
    assert np.array_equal(y, gold["fir_taps63"][:100]) or rel_rms(y, gold["fir_taps63"][:100]) == 0.0
    r = O.Resampler(gold["taps63"], 1, 8)
    assert len(r.process(gold["x"][:7])) == 0  # 7*1/8 = 0 outputs, history still advances


def test_math_blocks_known_answers():
    """src/dsp/math.h: Add / Substract per float, Multiply<complex_t> = complex product, every product
    and sum rounded on its own (VOLK generic)."""
    a = np.array([1 + 2j, 3 - 1j, 0.5 + 0.25j], dtype=np.complex64)
    b = np.array([2 - 1j, 1 + 1j, -4 + 8j], dtype=np.complex64)
    assert np.array_equal(O.math_op(0, a, b), np.array([3 + 1j, 4 + 0j, -3.5 + 8.25j], dtype=np.complex64))
    assert np.array_equal(O.math_op(1, a, b), np.array([-1 + 3j, 2 - 2j, 4.5 - 7.75j], dtype=np.complex64))
    assert np.array_equal(O.math_op(2, a, b), np.array([4 + 3j, 4 + 2j, -4 + 3j], dtype=np.complex64))
    assert np.array_equal(O.math_op(2, np.float32([1, 2, 3]), np.float32([4, 5, 6])), np.float32([4, 10, 18]))
    # separately rounded: differs from a fused evaluation on inputs chosen to show it
    x = np.array([1.0000001 + 1.0000002j], dtype=np.complex64)
    y = np.array([1.0000003 - 0.9999999j], dtype=np.complex64)
    xr, xi, yr, yi = (np.float32(v) for v in (x.real[0], x.imag[0], y.real[0], y.imag[0]))
    want = np.complex64(complex(np.float32(np.float32(xr * yr) - np.float32(xi * yi)), np.float32(np.float32(xr * yi) + np.float32(xi * yr))))
    assert O.math_op(2, x, y)[0] == want



def test_simd_lane_order_variant_is_the_same_filter():
    """ACC_SIMD (lane-partial sums + FMA, the shape of VOLK's SIMD kernels; bench.py's `value_simd`) computes the same
    FIR / resampler as the generic order: both sit within FP32 summation noise of the FP64 accumulation."""
    x = O.synth_iq(0, 30_000, seed=3)
    for ntaps in (4, 63, 64, 255, 256, 257):
        taps = O.lowpass_taps_f64(ntaps, 1 / 16)
        want = O.Fir(taps, acc=O.ACC_F64).process(x)
        a, b = O.Fir(taps, acc=O.ACC_SIMD), O.Fir(taps, acc=O.ACC_SIMD)
        y = a.process(x)
        assert rel_rms(y, want) < 5e-7, ntaps
        yb = np.concatenate([b.process(x[:7777]), b.process(x[7777:])])     # history carried as for every other order
        assert np.array_equal(y, yb)
        xr = np.ascontiguousarray(x.real)
        assert rel_rms(O.Fir(taps, complex_data=False, acc=O.ACC_SIMD).process(xr), O.Fir(taps, complex_data=False, acc=O.ACC_F64).process(xr)) < 5e-7
    taps = O.lowpass_taps_f64(256, 1 / 16)
    for L, M in ((1, 8), (3, 2), (2, 3), (1, 50)):
        assert rel_rms(O.Resampler(taps, L, M, acc=O.ACC_SIMD).process(x), O.Resampler(taps, L, M, acc=O.ACC_F64).process(x)) < 5e-7, (L, M)
