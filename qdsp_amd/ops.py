"""Operator-level front-end of the C ABI, one class per reference block on the hot path.

    Fir        <-> dsp::FIR<T>                 (src/dsp/filter.h:9-88)
    Resampler  <-> dsp::PolyphaseResampler<T>  (src/dsp/resampling.h:9-189)
    Xlator     <-> dsp::FrequencyXlator<T>     (src/dsp/processing.h:10-81)
    Vfo        <-> dsp::VFO                    (src/dsp/vfo.h), fused into one kernel

`process(x)`: a numpy array takes the host-pointer entry point (`*_process`, what a block's
run() calls on the stream buffers); a CUDA/HIP torch tensor takes `*_process_dev` on torch's
current stream.  One call == one run() of the reference block.  torch is only used for
device memory and streams.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

FL_M_PI = np.float32(3.1415926535)  # src/dsp/types.h:4


def phase_delta(sample_rate: float, freq: float):
    """phaseDelta exactly as FrequencyXlator::init computes it (processing.h:20): theta in
    float with FL_M_PI, then the float cos/sin overloads."""
    theta = np.float32(np.float32(np.float32(freq) / np.float32(sample_rate)) * np.float32(2.0)) * FL_M_PI
    return float(_libm().cosf(C.c_float(theta))), float(_libm().sinf(C.c_float(theta)))


_LIBM = None


def _libm():
    """glibc's cosf/sinf -- the same functions std::cos(float)/std::sin(float) resolve to in
    the C++ host code (numpy's float32 cos may differ in the last bit)."""
    global _LIBM
    if _LIBM is None:
        _LIBM = C.CDLL("libm.so.6")
        for f in (_LIBM.cosf, _LIBM.sinf):
            f.restype = C.c_float
            f.argtypes = [C.c_float]
    return _LIBM


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _taps_ptr(taps):
    t = np.ascontiguousarray(taps, dtype=np.float32)
    return t, t.ctypes.data_as(C.POINTER(C.c_float))


class _Op:
    _prefix = ""
    _ch = 2

    def __init__(self):
        self._L = capi.load()
        self._h = C.c_void_p()
        self.device = 0

    def _fn(self, name):
        return getattr(self._L, f"{self._prefix}_{name}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- data plumbing ---------------------------------------------------------------------
    def _np_in(self, x):
        a = np.ascontiguousarray(x)
        if self._ch == 2:
            if a.dtype != np.complex64:
                a = a.astype(np.complex64)
            return a, a.size
        return np.ascontiguousarray(a, dtype=np.float32), a.size

    def _out_size(self, n: int) -> int:
        return n

    def process(self, x, out=None):
        if _is_torch(x):
            return self._process_dev(x, out)
        a, n = self._np_in(x)
        no = self._out_size(n)
        y = np.empty(max(no, 1), dtype=np.complex64 if self._ch == 2 else np.float32)
        rc = self._fn("process")(self._h, a.ctypes.data, n, y.ctypes.data)
        capi.check(rc, self._prefix + "_process")
        return y[:no]

    def _process_dev(self, x, out=None):
        import torch

        assert x.is_cuda and x.is_contiguous()
        want = torch.complex64 if self._ch == 2 else torch.float32
        assert x.dtype == want, f"expected {want}"
        n = x.numel()
        no = self._out_size(n)
        if out is None:
            out = torch.empty(max(no, 1), dtype=want, device=x.device)
        assert out.is_cuda and out.is_contiguous() and out.numel() >= no and out.dtype == want
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = self._fn("process_dev")(self._h, x.data_ptr(), n, out.data_ptr(), stream)
        capi.check(int(rc), self._prefix + "_process_dev")
        return out[:no]

    def time_dev(self, x, out, iters: int) -> float:
        """Mean ms per launch of `iters` back-to-back process_dev calls, HIP events on the
        launch stream (qdsp_hip_time_process_dev)."""
        import torch

        stream = torch.cuda.current_stream(x.device).cuda_stream
        ms = C.c_float()
        rc = self._L.qdsp_hip_time_process_dev(self._h, x.data_ptr(), x.numel(), out.data_ptr(), stream, iters, C.byref(ms))
        capi.check(rc, "qdsp_hip_time_process_dev")
        return float(ms.value)

    def last_kernel(self):
        name = C.create_string_buffer(128)
        g, b, l = C.c_int(), C.c_int(), C.c_int()
        capi.check(self._L.qdsp_hip_last_kernel(self._h, name, 128, C.byref(g), C.byref(b), C.byref(l)))
        return {"name": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value}


class _HistMixin:
    AUTO, DIRECT, FFT = 0, 1, 2

    def set_mode(self, mode: int):
        """AUTO / DIRECT (direct form) / FFT (overlap-save fast convolution), *_set_mode."""
        capi.check(self._fn("set_mode")(self._h, int(mode)))

    def reset(self):
        capi.check(self._fn("reset")(self._h))

    @property
    def history_len(self) -> int:
        return capi.check(self._fn("history_len")(self._h))

    def get_history(self) -> np.ndarray:
        n = self.history_len
        a = np.zeros(max(n, 1), dtype=np.complex64 if self._ch == 2 else np.float32)
        capi.check(self._fn("get_history")(self._h, a.ctypes.data))
        return a[:n]

    def set_history(self, hist):
        a, n = self._np_in(hist)
        assert n == self.history_len, (n, self.history_len)
        capi.check(self._fn("set_history")(self._h, a.ctypes.data))

    def history_dev_ptr(self) -> int:
        p = C.c_void_p()
        capi.check(self._fn("history_dev")(self._h, C.byref(p)))
        return p.value or 0

    def set_history_dev(self, t):
        """Device-to-device copy of `history_len` samples from torch tensor `t` into the history
        the next call reads, on torch's current stream (*_set_history_dev)."""
        import torch

        assert t.is_cuda and t.is_contiguous() and t.numel() == self.history_len
        stream = torch.cuda.current_stream(t.device).cuda_stream
        capi.check(self._fn("set_history_dev")(self._h, t.data_ptr(), stream))

    def set_history_ptr(self, ptr: int, stream: int):
        """The same from a raw device pointer (the receive buffer of the C ring, qdsp_hip_ring_complete) on HIP stream `stream`."""
        capi.check(self._fn("set_history_dev")(self._h, int(ptr), int(stream)))

    def history_dev_tensor(self):
        """The device history buffer the NEXT process call reads, as a torch view (no copy):
        an RCCL recv of the neighbour's tail can land here directly (multi-GPU halo)."""
        import torch

        n = self.history_len
        ptr = self.history_dev_ptr()
        nfloat = n * self._ch

        class _Holder:
            pass

        holder = _Holder()
        holder.__cuda_array_interface__ = {
            "shape": (nfloat,), "typestr": "<f4", "data": (ptr, False), "version": 2,
        }
        t = torch.as_tensor(holder, device=f"cuda:{self.device}")
        return torch.view_as_complex(t.view(n, 2)) if self._ch == 2 else t


class Fir(_Op, _HistMixin):
    def __init__(self, taps, complex_data: bool = True, device: int = 0, max_block: int = 1_000_000):
        super().__init__()
        self._ch = 2 if complex_data else 1
        self._prefix = "qdsp_hip_fir_cf32" if complex_data else "qdsp_hip_fir_f32"
        self.device = device
        self._taps, p = _taps_ptr(taps)
        capi.check(self._fn("create")(C.byref(self._h), device, p, len(self._taps), max_block), self._prefix + "_create")

    def set_taps(self, taps):
        self._taps, p = _taps_ptr(taps)
        capi.check(self._fn("set_taps")(self._h, p, len(self._taps)))


class Resampler(_Op, _HistMixin):
    def __init__(self, taps, interp: int, decim: int, complex_data: bool = True, device: int = 0,
                 max_block: int = 1_000_000):
        super().__init__()
        self._ch = 2 if complex_data else 1
        self._prefix = "qdsp_hip_decim_cf32" if complex_data else "qdsp_hip_decim_f32"
        self.device = device
        self._taps, p = _taps_ptr(taps)
        capi.check(self._fn("create")(C.byref(self._h), device, p, len(self._taps), int(interp), int(decim), max_block),
                   self._prefix + "_create")

    def configure(self, taps, interp: int, decim: int):
        self._taps, p = _taps_ptr(taps)
        capi.check(self._fn("configure")(self._h, p, len(self._taps), int(interp), int(decim)))

    def _out_size(self, n: int) -> int:
        return int(capi.check(self._fn("out_size")(self._h, n)))


class _NcoMixin:
    def set_phase_inc(self, re: float, im: float):
        capi.check(self._fn("set_phase_inc")(self._h, re, im))

    def get_phase(self) -> complex:
        re, im = C.c_float(), C.c_float()
        capi.check(self._fn("get_phase")(self._h, C.byref(re), C.byref(im)))
        return complex(re.value, im.value)

    def set_phase(self, re: float, im: float):
        capi.check(self._fn("set_phase")(self._h, re, im))

    def advance(self, nsamples: int):
        capi.check(self._fn("advance")(self._h, int(nsamples)))

    def set_volk_gain(self, on: bool):
        capi.check(self._fn("set_volk_gain")(self._h, int(bool(on))))


class Xlator(_Op, _NcoMixin):
    _prefix = "qdsp_hip_xlate_cf32"

    def __init__(self, sample_rate: float = None, freq: float = None, phase_inc=None, device: int = 0,
                 max_block: int = 1_000_000):
        super().__init__()
        self.device = device
        re, im = phase_inc if phase_inc is not None else phase_delta(sample_rate, freq)
        self.phase_inc = (re, im)
        capi.check(self._fn("create")(C.byref(self._h), device, re, im, max_block), self._prefix + "_create")


class Vfo(_Op, _HistMixin, _NcoMixin):
    """xlator(-offset) -> resampler in one kernel.  `taps`/interp/decim come from the caller
    (the window design stays on the host, SURVEY a5: qdsp_amd/host/dsp/window.h)."""

    _prefix = "qdsp_hip_xlate_fir_decim_cf32"

    def __init__(self, taps, interp: int, decim: int, phase_inc, device: int = 0, max_block: int = 1_000_000):
        super().__init__()
        self.device = device
        self._taps, p = _taps_ptr(taps)
        re, im = phase_inc
        capi.check(self._fn("create")(C.byref(self._h), device, p, len(self._taps), int(interp), int(decim), re, im,
                                      max_block), self._prefix + "_create")

    def configure(self, taps, interp: int, decim: int):
        self._taps, p = _taps_ptr(taps)
        capi.check(self._fn("configure")(self._h, p, len(self._taps), int(interp), int(decim)))

    def _out_size(self, n: int) -> int:
        return int(capi.check(self._fn("out_size")(self._h, n)))


class Channelizer:
    """Splitter -> N x VFO (src/dsp/routing.h:47-57, src/dsp/vfo.h) as one operator:
    qdsp_hip_chan_cf32_*.  process() returns an (nchan, outCount) array / tensor."""

    AUTO, DIRECT, FFT = 0, 1, 2

    def __init__(self, taps, interp: int, decim: int, phase_incs, device: int = 0, max_block: int = 1_000_000):
        self._L = capi.load()
        self._h = C.c_void_p()
        self.device = device
        self._taps, p = _taps_ptr(taps)
        incs = np.asarray(phase_incs, dtype=np.float32).reshape(-1, 2)
        self.nchan = len(incs)
        re, im = np.ascontiguousarray(incs[:, 0]), np.ascontiguousarray(incs[:, 1])
        fp = C.POINTER(C.c_float)
        capi.check(self._L.qdsp_hip_chan_cf32_create(C.byref(self._h), device, p, len(self._taps), int(interp), int(decim),
                                                     self.nchan, re.ctypes.data_as(fp), im.ctypes.data_as(fp), max_block),
                   "qdsp_hip_chan_cf32_create")

    def close(self):
        if self._h:
            self._L.qdsp_hip_chan_cf32_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode: int):
        capi.check(self._L.qdsp_hip_chan_cf32_set_mode(self._h, int(mode)))

    def set_volk_gain(self, on: bool):
        capi.check(self._L.qdsp_hip_chan_cf32_set_volk_gain(self._h, int(bool(on))))

    def reset(self):
        capi.check(self._L.qdsp_hip_chan_cf32_reset(self._h))

    def out_size(self, n: int) -> int:
        return int(capi.check(self._L.qdsp_hip_chan_cf32_out_size(self._h, n)))

    _prefix = "qdsp_hip_chan_cf32"

    @property
    def history_len(self) -> int:
        return int(capi.check(self._L.qdsp_hip_chan_cf32_history_len(self._h)))

    def set_history_dev(self, hist):
        """The history_len raw input samples preceding the next call (device tensor)."""
        import torch

        assert hist.is_cuda and hist.is_contiguous() and hist.dtype == torch.complex64 and hist.numel() == self.history_len
        stream = torch.cuda.current_stream(hist.device).cuda_stream
        capi.check(self._L.qdsp_hip_chan_cf32_set_history_dev(self._h, hist.data_ptr(), stream))

    def set_history_ptr(self, ptr: int, stream: int):
        """The same from a raw device pointer (the receive buffer of the C ring, qdsp_hip_ring_complete) on HIP stream `stream`."""
        capi.check(self._L.qdsp_hip_chan_cf32_set_history_dev(self._h, int(ptr), int(stream)))

    def advance(self, n: int):
        capi.check(self._L.qdsp_hip_chan_cf32_advance(self._h, int(n)))

    def last_kernel(self):
        name = C.create_string_buffer(128)
        g, b, l = C.c_int(), C.c_int(), C.c_int()
        capi.check(self._L.qdsp_hip_last_kernel(self._h, name, 128, C.byref(g), C.byref(b), C.byref(l)))
        return {"name": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value}

    def time_dev(self, x, out, iters: int) -> float:
        """Mean ms per process_dev over `iters` calls (torch events on the current stream)."""
        import torch

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            self.process(x, out)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters

    def process(self, x, out=None):
        if _is_torch(x):
            import torch

            assert x.is_cuda and x.is_contiguous() and x.dtype == torch.complex64
            n, no = x.numel(), self.out_size(x.numel())
            if out is None:
                out = torch.empty((self.nchan, max(no, 1)), dtype=torch.complex64, device=x.device)
            assert out.is_contiguous() and out.shape[0] == self.nchan and out.shape[1] >= no
            stream = torch.cuda.current_stream(x.device).cuda_stream
            rc = self._L.qdsp_hip_chan_cf32_process_dev(self._h, x.data_ptr(), n, out.data_ptr(), out.shape[1], stream)
            capi.check(int(rc), "qdsp_hip_chan_cf32_process_dev")
            return out[:, :no]
        a = np.ascontiguousarray(x, dtype=np.complex64)
        no = self.out_size(a.size)
        y = np.empty((self.nchan, max(no, 1)), dtype=np.complex64)
        rc = self._L.qdsp_hip_chan_cf32_process(self._h, a.ctypes.data, a.size, y.ctypes.data, y.shape[1])
        capi.check(rc, "qdsp_hip_chan_cf32_process")
        return y[:, :no]


def synth_iq(count: int, first_sample: int = 0, seed: int = 1234, device: int = 0, out=None):
    """Counter-based uniform IQ generated on the device (measurement input)."""
    import torch

    if out is None:
        out = torch.empty(count, dtype=torch.complex64, device=f"cuda:{device}")
    stream = torch.cuda.current_stream(out.device).cuda_stream
    capi.check(capi.load().qdsp_hip_synth_iq_dev(device, out.data_ptr(), first_sample, count, seed, stream),
               "qdsp_hip_synth_iq_dev")
    return out


def device_info(device: int = 0) -> dict:
    L = capi.load()
    name, arch, cus = C.create_string_buffer(256), C.create_string_buffer(256), C.c_int()
    capi.check(L.qdsp_hip_device_info(device, name, 256, arch, 256, C.byref(cus)))
    return {"name": name.value.decode(), "arch": arch.value.decode(), "compute_units": cus.value}


__all__ = ["Fir", "Resampler", "Xlator", "Vfo", "Channelizer", "synth_iq", "phase_delta", "device_info"]


class Math:
    """Add / Substract / Multiply of two streams (src/dsp/math.h:7-145): qdsp_hip_math_*."""

    ADD, SUB, MUL = 0, 1, 2

    def __init__(self, op: int, complex_data: bool = True, device: int = 0, max_block: int = 1_000_000):
        self._L = capi.load()
        self._h = C.c_void_p()
        self.complex_data = complex_data
        capi.check(self._L.qdsp_hip_math_create(C.byref(self._h), device, int(op), int(complex_data), max_block), "qdsp_hip_math_create")

    def close(self):
        if self._h:
            self._L.qdsp_hip_math_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def process(self, a, b, out=None):
        if _is_torch(a):
            import torch

            dt = torch.complex64 if self.complex_data else torch.float32
            assert a.is_cuda and b.is_cuda and a.dtype == dt and b.dtype == dt and a.numel() == b.numel()
            assert a.is_contiguous() and b.is_contiguous()
            if out is None:
                out = torch.empty_like(a)
            stream = torch.cuda.current_stream(a.device).cuda_stream
            capi.check(self._L.qdsp_hip_math_process_dev(self._h, a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), stream),
                       "qdsp_hip_math_process_dev")
            return out
        dt = np.complex64 if self.complex_data else np.float32
        x, y = np.ascontiguousarray(a, dtype=dt), np.ascontiguousarray(b, dtype=dt)
        assert x.size == y.size
        o = np.empty_like(x)
        capi.check(self._L.qdsp_hip_math_process(self._h, x.ctypes.data, y.ctypes.data, x.size, o.ctypes.data), "qdsp_hip_math_process")
        return o

