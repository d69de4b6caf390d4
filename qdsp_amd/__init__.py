"""qdsp_amd -- MI355X-native FIR / polyphase-resampler / NCO-mixer path of qdsp.

Layout
    csrc/        hand-written HIP kernels (gfx950) + the extern "C" boundary (include/qdsp_hip.h)
    host/dsp/    C++ mirror of the reference's block-graph API (stream / generic_block and the
                 HIP-backed FIR, PolyphaseResampler, FrequencyXlator, VFO blocks)
    capi.py      ctypes binding of the C ABI
    ops.py       operator-level front-end used by tests/ and bench.py
    sharding.py  time-axis chunking of one IQ stream over ranks with an (ntaps-1) halo

Nothing in this package imports oracle/ (the CPU checker); there is no CPU fallback.
"""
from . import capi  # noqa: F401

__version__ = "0.1.0"
