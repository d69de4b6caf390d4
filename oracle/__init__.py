"""ctypes/numpy front-end of the CPU oracle (oracle/qdsp_oracle.c).

TEST INFRASTRUCTURE ONLY -- parity unpinned (see the header of qdsp_oracle.c).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under qdsp_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libqdsp_oracle.so")

ACC_F32, ACC_FMA, ACC_F64, ACC_SIMD = 0, 1, 2, 3   # ACC_SIMD: lane-partial sums + FMA, the shape of VOLK's SIMD kernels

_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle with plain gcc (seconds)."""
    src = os.path.join(_HERE, "qdsp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libqdsp_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.oracle_fir_cf32.restype = C.c_long
        L.oracle_fir_cf32.argtypes = [fp, C.c_int, fp, fp, C.c_long, fp, C.c_int]
        L.oracle_fir_f32.restype = C.c_long
        L.oracle_fir_f32.argtypes = [fp, C.c_int, fp, fp, C.c_long, fp, C.c_int]
        for f in (L.oracle_resamp_cf32, L.oracle_resamp_f32):
            f.restype = C.c_long
            f.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_long, fp, C.c_int]
        L.oracle_resamp_ratio.restype = None
        L.oracle_resamp_ratio.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_resamp_taps_per_phase.restype = C.c_int
        L.oracle_resamp_taps_per_phase.argtypes = [C.c_int, C.c_int]
        L.oracle_resamp_out_size.restype = C.c_long
        L.oracle_resamp_out_size.argtypes = [C.c_long, C.c_int, C.c_int]
        L.oracle_resamp_build_phases.restype = None
        L.oracle_resamp_build_phases.argtypes = [fp, C.c_int, C.c_int, fp]
        L.oracle_xlator_phase_delta.restype = None
        L.oracle_xlator_phase_delta.argtypes = [C.c_float, C.c_float, fp]
        L.oracle_math_f32.restype = None
        L.oracle_math_f32.argtypes = [C.c_int, fp, fp, fp, C.c_long]
        L.oracle_mul_cf32.restype = None
        L.oracle_mul_cf32.argtypes = [fp, fp, fp, C.c_long]
        L.oracle_rotator_cf32.restype = None
        L.oracle_rotator_cf32.argtypes = [fp, fp, fp, fp, C.c_long]
        L.oracle_rotator_cf32_f64.restype = None
        L.oracle_rotator_cf32_f64.argtypes = [fp, fp, fp, C.POINTER(C.c_double), C.c_long, C.c_int]
        L.oracle_blackman_tap_count.restype = C.c_int
        L.oracle_blackman_tap_count.argtypes = [C.c_float, C.c_float, C.c_float]
        L.oracle_blackman_taps.restype = None
        L.oracle_blackman_taps.argtypes = [C.c_float, C.c_float, fp, C.c_int, C.c_float]
        L.oracle_blackman_bandpass_taps.restype = None
        L.oracle_blackman_bandpass_taps.argtypes = [C.c_float, C.c_float, C.c_float, fp, C.c_int, C.c_float]
        L.oracle_rrc_taps.restype = C.c_int
        L.oracle_rrc_taps.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, fp]
        L.oracle_vfo_design.restype = C.c_int
        L.oracle_vfo_design.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int), fp, C.c_int]
        L.oracle_synth_iq.restype = None
        L.oracle_synth_iq.argtypes = [fp, C.c_long, C.c_long, C.c_uint32]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _as_f32(x) -> np.ndarray:
    """complex64 / float32 array -> flat float32 view (interleaved re,im for complex)."""
    a = np.ascontiguousarray(x)
    if a.dtype == np.complex64:
        return a.view(np.float32)
    return np.ascontiguousarray(a, dtype=np.float32)


class Fir:
    """dsp::FIR<T> (src/dsp/filter.h:9-88): stateful, one process() == one run()."""

    def __init__(self, taps, complex_data: bool = True, acc: int = ACC_F32):
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.ch = 2 if complex_data else 1
        self.acc = acc
        self.hist = np.zeros(len(self.taps) * self.ch, dtype=np.float32)

    def reset(self):
        self.hist[:] = 0

    def process(self, x) -> np.ndarray:
        xin = _as_f32(x)
        n = xin.size // self.ch
        out = np.empty(n * self.ch, dtype=np.float32)
        fn = lib().oracle_fir_cf32 if self.ch == 2 else lib().oracle_fir_f32
        r = fn(_fp(self.taps), len(self.taps), _fp(self.hist), _fp(xin), n, _fp(out), self.acc)
        assert r == n
        return out.view(np.complex64) if self.ch == 2 else out


class Resampler:
    """dsp::PolyphaseResampler<T> (src/dsp/resampling.h:9-189); `taps` is the prototype as
    the window produced it (i.e. already scaled by interp)."""

    def __init__(self, taps, interp: int, decim: int, complex_data: bool = True, acc: int = ACC_F32):
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.interp, self.decim = int(interp), int(decim)
        self.ch = 2 if complex_data else 1
        self.acc = acc
        self.tpp = lib().oracle_resamp_taps_per_phase(len(self.taps), self.interp)
        self.hist = np.zeros(self.tpp * self.ch, dtype=np.float32)

    def reset(self):
        self.hist[:] = 0

    def out_size(self, n: int) -> int:
        return lib().oracle_resamp_out_size(n, self.interp, self.decim)

    def process(self, x) -> np.ndarray:
        xin = _as_f32(x)
        n = xin.size // self.ch
        out = np.empty(max(self.out_size(n), 1) * self.ch, dtype=np.float32)
        fn = lib().oracle_resamp_cf32 if self.ch == 2 else lib().oracle_resamp_f32
        r = fn(_fp(self.taps), len(self.taps), self.interp, self.decim, _fp(self.hist), _fp(xin), n,
               _fp(out), self.acc)
        assert r == self.out_size(n)
        out = out[: r * self.ch]
        return out.view(np.complex64) if self.ch == 2 else out


class Xlator:
    """dsp::FrequencyXlator<complex_t> (src/dsp/processing.h:10-81) on VOLK's generic
    rotator.  exact=True swaps in the FP64-phase NCO (the drift-free yardstick);
    volk_gain=True keeps the generic rotator's deterministic magnitude sawtooth in it."""

    def __init__(self, sample_rate: float, freq: float, exact: bool = False, volk_gain: bool = False):
        self.delta = np.zeros(2, dtype=np.float32)
        lib().oracle_xlator_phase_delta(sample_rate, freq, _fp(self.delta))
        self.phase = np.array([1.0, 0.0], dtype=np.float32)
        self.turns = C.c_double(0.0)
        self.exact = exact
        self.volk_gain = volk_gain

    def process(self, x) -> np.ndarray:
        xin = _as_f32(x)
        n = xin.size // 2
        out = np.empty(2 * n, dtype=np.float32)
        if self.exact:
            lib().oracle_rotator_cf32_f64(_fp(xin), _fp(out), _fp(self.delta), C.byref(self.turns), n, int(self.volk_gain))
        else:
            lib().oracle_rotator_cf32(_fp(xin), _fp(out), _fp(self.delta), _fp(self.phase), n)
        return out.view(np.complex64)


def resamp_ratio(in_rate: float, out_rate: float):
    i, d = C.c_int(), C.c_int()
    lib().oracle_resamp_ratio(in_rate, out_rate, C.byref(i), C.byref(d))
    return i.value, d.value


def build_phases(taps, interp: int) -> np.ndarray:
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    tpp = lib().oracle_resamp_taps_per_phase(len(taps), interp)
    out = np.zeros((interp, tpp), dtype=np.float32)
    lib().oracle_resamp_build_phases(_fp(taps), len(taps), interp, _fp(out.reshape(-1)))
    return out


def blackman_tap_count(cutoff, trans_width, sample_rate) -> int:
    return lib().oracle_blackman_tap_count(cutoff, trans_width, sample_rate)


def blackman_taps(cutoff, sample_rate, ntaps: int, factor: float = 1.0) -> np.ndarray:
    t = np.zeros(ntaps, dtype=np.float32)
    lib().oracle_blackman_taps(cutoff, sample_rate, _fp(t), ntaps, factor)
    return t


def blackman_bandpass_taps(cutoff, offset, sample_rate, ntaps: int, factor: float = 1.0) -> np.ndarray:
    t = np.zeros(ntaps, dtype=np.float32)
    lib().oracle_blackman_bandpass_taps(cutoff, offset, sample_rate, _fp(t), ntaps, factor)
    return t


def rrc_taps(ntaps: int, sample_rate, baud_rate, alpha) -> np.ndarray:
    t = np.zeros(ntaps, dtype=np.float32)
    r = lib().oracle_rrc_taps(ntaps, sample_rate, baud_rate, alpha, _fp(t))
    if r < 0:
        raise ValueError("RRCTaps needs an odd tap count (src/dsp/window.h:183)")
    return t


def vfo_design(in_rate, out_rate, bandwidth):
    """-> (interp, decim, taps) exactly as dsp::VFO::init derives them (src/dsp/vfo.h:19-36)."""
    i, d = C.c_int(), C.c_int()
    n = lib().oracle_vfo_design(in_rate, out_rate, bandwidth, C.byref(i), C.byref(d), None, 0)
    t = np.zeros(n, dtype=np.float32)
    lib().oracle_vfo_design(in_rate, out_rate, bandwidth, C.byref(i), C.byref(d), _fp(t), n)
    return i.value, d.value, t


class Vfo:
    """dsp::VFO (src/dsp/vfo.h): FrequencyXlator(-offset) -> PolyphaseResampler."""

    def __init__(self, offset, in_rate, out_rate, bandwidth, exact_nco: bool = False, acc: int = ACC_F32,
                 volk_gain: bool = False):
        self.interp, self.decim, self.taps = vfo_design(in_rate, out_rate, bandwidth)
        self.xl = Xlator(in_rate, -offset, exact=exact_nco, volk_gain=volk_gain)
        self.rs = Resampler(self.taps, self.interp, self.decim, True, acc)

    def process(self, x) -> np.ndarray:
        return self.rs.process(self.xl.process(x))


def math_op(op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Add (0) / Substract (1) / Multiply (2) of two streams, src/dsp/math.h, VOLK-generic arithmetic.
    complex64 inputs: add / subtract per float, multiply = complex product; float32: per element."""
    cplx = np.iscomplexobj(a)
    dt = np.complex64 if cplx else np.float32
    x, y = np.ascontiguousarray(a, dtype=dt), np.ascontiguousarray(b, dtype=dt)
    assert x.shape == y.shape
    out = np.empty_like(x)
    xf, yf, of = x.view(np.float32), y.view(np.float32), out.view(np.float32)
    if cplx and op == 2:
        lib().oracle_mul_cf32(_fp(xf), _fp(yf), _fp(of), x.size)
    else:
        lib().oracle_math_f32(op, _fp(xf), _fp(yf), _fp(of), xf.size)
    return out


def synth_iq(first_sample: int, count: int, seed: int = 1234) -> np.ndarray:
    out = np.empty(2 * count, dtype=np.float32)
    lib().oracle_synth_iq(_fp(out), first_sample, count, seed)
    return out.view(np.complex64)


def lowpass_taps_f64(ntaps: int, fc: float) -> np.ndarray:
    """Harness-defined taps for the 256-tap configs: true Blackman
    windowed sinc designed in FP64, cutoff fc cycles/sample, unit DC gain, cast to f32.
    Not part of the reference (its BlackmanWindow cannot produce an even count)."""
    n = np.arange(ntaps, dtype=np.float64)
    c = (ntaps - 1) / 2.0
    h = 2 * fc * np.sinc(2 * fc * (n - c))
    w = 0.42 - 0.5 * np.cos(2 * np.pi * n / (ntaps - 1)) + 0.08 * np.cos(4 * np.pi * n / (ntaps - 1))
    h = h * w
    return (h / h.sum()).astype(np.float32)
