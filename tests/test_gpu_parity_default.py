"""The golden-vector, known-answer and property tests of test_gpu_parity.py once more under the SHIPPED dispatch.

tests/conftest.py lifts four size thresholds and the measured exception table for test_gpu_parity.py, so that the kernels the
bench-sized calls run are exercised at oracle-sized inputs.  That leaves the question VERDICT round 3 asked: do the same
golden vectors (tests/golden/vectors.npz, kat.json), block-invariance, history, retune and degenerate-size properties hold on
the kernels a USER's calls of these sizes reach -- rule chain + dispatch_table.inc + decim_table.inc, no QDSP_HIP_* override?
This module re-collects every test of test_gpu_parity.py that does not pin a kernel by name (25 functions) under its own module
name, for which conftest.py sets nothing.  One of them is restated: FIR<float> promises the fmaf chain bit for bit only in DIRECT mode; in AUTO the
measured table may hand a 2 048-sample block of a 256-tap filter to an overlap-save kernel (1e-6 of the golden vector, not bit-identical)."""
import os

import numpy as np
import pytest

import oracle as O
from conftest import rel_rms

import test_gpu_parity as P
from test_gpu_parity import gold, kat, ops  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

_NAMES = [
    "test_kat_fir", "test_kat_resampler", "test_kat_xlator", "test_fir_golden_host_path",
    "test_fir_device_path_and_block_invariance", "test_fir_empty_and_reset_and_history", "test_fir_set_taps_keeps_stream",
    "test_fir_nan_inf_stay_local", "test_fft_fir_chunk_invariance_and_linearity", "test_resampler_golden",
    "test_resampler_256_decim8_and_f32", "test_resampler_ratios_vs_f64", "test_resampler_zero_output_block_and_phase_restart",
    "test_xlator_golden", "test_xlator_long_stream_exact_phase",
    "test_xlator_deviation_from_the_reference_recursion_is_the_references_own_drift", "test_vfo_golden",
    "test_vfo_equals_xlator_then_resampler_on_device", "test_full_size_properties", "test_vfo_set_history_dev_rotates_raw_samples",
    "test_fir_and_resampler_set_history_dev", "test_math_blocks_bit_exact", "test_degenerate_block_sizes_every_kernel",
    "test_vfo_retune_mid_stream",
]
for _n in _NAMES:
    globals()[_n] = getattr(P, _n)


def test_this_module_runs_without_overrides():
    assert not [k for k in os.environ if k.startswith("QDSP_HIP_")], "the point of this module is the shipped dispatch"


@pytest.mark.parametrize("name", ["taps63", "taps256"])
def test_fir_f32_golden(ops, gold, name):
    xr = np.ascontiguousarray(gold["x"].real)
    y = P.run_blocks(ops.Fir(gold[name], complex_data=False), xr, [1000, 37, 1, 2048, 5])
    assert rel_rms(y, gold[f"firf32_{name}"]) < 1e-6
    assert rel_rms(y, O.Fir(gold[name], complex_data=False, acc=O.ACC_F64).process(xr)) < 1e-6
    op = ops.Fir(gold[name], complex_data=False)
    op.set_mode(op.DIRECT)
    assert np.array_equal(P.run_blocks(op, xr, [1000, 37, 1, 2048, 5]), O.Fir(gold[name], complex_data=False, acc=O.ACC_FMA).process(xr))
