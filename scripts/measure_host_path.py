#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (what a block's run() calls): 1e6-sample blocks in
pinned stream buffers, H2D + kernel + D2H, synchronous per call."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from qdsp_amd import capi, ops

L = capi.load()
n = 1_000_000
def pinned(nbytes):
    p = C.c_void_p()
    capi.check(L.qdsp_hip_host_alloc(C.byref(p), nbytes), "host_alloc")
    return p
pin, pout = pinned(n * 8), pinned(n * 8)
x = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_float)), shape=(2 * n,))
x[:] = np.random.default_rng(0).standard_normal(2 * n).astype(np.float32)
for name, mk in (("fir256", lambda: ops.Fir(bench.lowpass_taps(256, 1 / 16), max_block=n)),
                 ("xlate_fir_decim8", lambda: ops.Vfo(bench.lowpass_taps(256, 1 / 16), 1, 8, ops.phase_delta(1.0, 0.1234), max_block=n))):
    op = mk()
    fn = getattr(L, op._prefix + "_process")
    for _ in range(5):
        capi.check(fn(op._h, pin, n, pout))
    t0 = time.perf_counter()
    k = 50
    for _ in range(k):
        capi.check(fn(op._h, pin, n, pout))
    dt = (time.perf_counter() - t0) / k
    print(f"{name}: {dt*1e3:.3f} ms per 1e6-sample block = {n/dt/1e6:.0f} Msamples/s (PCIe-inclusive, pinned buffers)")
