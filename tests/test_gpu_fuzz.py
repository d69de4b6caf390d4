"""The SHIPPED dispatch under the driver's eyes: the library's own size thresholds, no overrides.

Most of tests/test_gpu_parity.py lifts or lowers call-size thresholds (tests/conftest.py) so that oracle-sized inputs reach
the kernels the bench-sized calls run.  This module does the opposite: every operator is created and called the way a user
calls it -- no QDSP_HIP_* variable set -- and judged against the FP64-accumulating oracle (oracle/qdsp_oracle.c, restating
src/dsp/filter.h:51-74, src/dsp/resampling.h:99-132, src/dsp/processing.h:55-70):

  * directed plans, one per kernel family `process_dev` can pick, each at a call size on the side of its crossover where that
    family is the default, plus the same plan on the OTHER side of the crossover where the rule is a call-size rule;
  * a fixed-seed, time-boxed slice of scripts/fuzz_dispatch.py in default_only mode.
The union of the kernel families seen must cover all of them."""
import os
import sys

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = [pytest.mark.gpu, pytest.mark.default_dispatch]

TOL = 3e-6
_SEEN = set()          # kernel families reached so far by this module (one process, tests in file order)
_NOT_DISPATCH = ("QDSP_HIP_LIB", "QDSP_HIP_NO_AUTOBUILD", "QDSP_HIP_DEVICE")


def _no_overrides():
    return not any(k.startswith("QDSP_HIP_") and k not in _NOT_DISPATCH for k in os.environ)

FAMILIES = {"fir_core_kernel", "fir_lat_kernel", "fir_fft1k_kernel", "fir_fft_dma_kernel", "pfb_dec8_kernel", "pfb_dec8_real_kernel", "pfb_dec4_kernel", "pfb_dec4_real_kernel", "decim_win_kernel",
            "decim_mfma_kernel", "decim_mfma_real_kernel", "decim_mfma_batch_kernel", "resamp_lm_kernel", "resamp_mfma_kernel", "resamp_mfma_real_kernel", "resamp_any_kernel",
            "resamp_any_batch_kernel", "chan_uniform_kernel"}


@pytest.fixture(scope="module")
def ops():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from qdsp_amd import ops as _ops

    return _ops


def _rel(a, b, floor=0.0):
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / max(np.mean(np.abs(b) ** 2), floor ** 2, 1e-30)))


def _lp(ntaps, fc, gain=1.0):
    return (O.lowpass_taps_f64(ntaps, fc) * gain).astype(np.float32)


def _dev(x):
    import torch

    return torch.from_numpy(x).cuda()


# (what, interp, decim, ntaps, nco, call sizes and the kernel family each call must land on)
DIRECTED = [
    ("fir", 1, 1, 63, False, [(65_536, "fir_lat_kernel")]),
    # (round 4: FIR<complex_t> follows the measured table, qdsp_amd/csrc/dispatch_table.inc -- cells chosen away from its borders)
    ("fir", 1, 1, 15, False, [(1_000_000, "fir_core_kernel"), (2_000_000, "fir_fft1k_kernel")]),
    ("fir", 1, 1, 256, False, [(16_384, "fir_lat_kernel"), (1_000_000, "fir_fft1k_kernel"), (1 << 26, "fir_fft_dma_kernel")]),
    ("res", 1, 8, 256, False, [(1_000_000, "fir_fft1k_kernel"), ((1 << 23) + 8, "pfb_dec8_kernel")]),
    ("res", 1, 8, 256, True, [((1 << 24) + 16, "pfb_dec8_kernel")]),
    # round 3's big-call rule: past the measured crossovers the strided-window decimator hands chip-filling calls to the overlap-save forms
    # (round 4: the decimators' rule chain has measured exceptions, qdsp_amd/csrc/decim_table.inc -- e.g. the strided-window kernel keeps the
    # 2^24-sample calls of this 128-tap fused VFO (34.7 us against 59.3 on the polyphase kernel), which takes over at 2^26)
    ("res", 1, 8, 128, True, [(3_000_000, "decim_win_kernel"), ((1 << 24) + 8, "decim_win_kernel"), ((1 << 26) + 8, "pfb_dec8_kernel")]),
    # (... and decimate-by-2 changes form inside fir_fft_kernel at 2^25 samples: pruned inverse below, full inverse + every other output kept above)
    # fused VFO through the full inverse, big enough (> 4096 segments) that workgroups take a second segment: the segment phasor they
    # carry in LDS from one to the next (round 3's four-workgroup form of fir_fft_kernel<1, ROT>)
    ("res", 1, 3, 256, True, [((1 << 24) + 2 * 4096 + 3, "fir_fft_kernel")]),
    ("res", 1, 2, 128, False, [(200_000, "fir_fft1k_kernel"), (3_000_000, "fir_core_kernel"), ((1 << 24) + 2, "fir_fft_kernel"), ((1 << 25) + 6, "fir_fft_kernel")]),
    ("res", 1, 4, 63, False, [(200_000, "decim_win_kernel")]),
    ("res", 1, 4, 160, True, [(2_000_000, "fir_core_kernel"), ((1 << 26) + 4, "pfb_dec4_kernel")]),
    ("res", 1, 10, 1024, False, [(8192, "resamp_any_kernel")]),      # (a table exception: 5.5 us against 10.2 on the overlap-save kernel the rules pick)
    ("res", 1, 8, 63, False, [(100_000, "decim_win_kernel"), (3_000_000, "decim_win_kernel")]),
    ("res", 1, 16, 129, True, [(300_000, "decim_mfma_kernel")]),
    ("res", 1, 50, 401, True, [(100_000, "resamp_any_kernel"), (3_200_000, "decim_mfma_kernel")]),
    ("res", 3, 2, 36, False, [(100_000, "resamp_lm_kernel")]),
    ("res", 10, 7, 160, False, [(100_000, "resamp_mfma_kernel")]),
    ("res", 7, 5, 140, True, [(100_000, "resamp_any_kernel")]),
    # round 3: from 16 taps per phase on, chip-filling calls of the small ratios go to the block form on the MFMA units as well
    ("res", 7, 5, 7 * 24 - 3, True, [(200_000, "resamp_any_kernel"), (13_000_005, "resamp_mfma_kernel")]),
    ("res", 4, 3, 4 * 20 - 3, False, [(300_000, "resamp_lm_kernel"), (13_000_003, "resamp_mfma_kernel")]),
    ("res", 147, 160, 147 * 16 - 3, False, [(200_000, "resamp_any_kernel"), (6_400_000, "resamp_mfma_kernel")]),
]


@pytest.mark.parametrize("plan", DIRECTED, ids=lambda p: f"{p[0]}_{p[1]}_{p[2]}_{p[3]}{'_nco' if p[4] else ''}")
def test_directed_plans_default_thresholds(ops, plan):
    what, L, M, ntaps, nco, calls = plan
    assert _no_overrides(), "this module runs under the library's own thresholds"
    taps = _lp(ntaps, 0.45 / max(L, M), L)
    f = 0.1234
    if what == "fir":
        op, orc = ops.Fir(taps, max_block=0), O.Fir(taps, acc=O.ACC_F64)
        ref = orc.process
    elif nco:
        op = ops.Vfo(taps, L, M, ops.phase_delta(1.0, f), max_block=0)
        xl, rs = O.Xlator(1.0, f, exact=True, volk_gain=True), O.Resampler(taps, L, M, acc=O.ACC_F64)
        ref = lambda b: rs.process(xl.process(b))   # noqa: E731
    else:
        op, rs = ops.Resampler(taps, L, M, max_block=0), O.Resampler(taps, L, M, acc=O.ACC_F64)
        ref = rs.process
    pos = 0
    for count, family in calls:      # one continuous stream: the handle's history crosses the kernel families
        x = O.synth_iq(pos, count, seed=ntaps + M)
        pos += count
        got = op.process(_dev(x)).cpu().numpy()
        name = op.last_kernel()["name"]
        assert name == family, (plan, count, name)
        want = ref(x)
        assert got.shape == want.shape and _rel(got, want) < TOL, (plan, count, name)
        _SEEN.add(name)


REAL_DIRECTED = [
    # (decimation, taps, call sizes and the kernel family each must land on): PolyphaseResampler<float> / FIR<float> on both sides of the
    # chip-filling rule of round 3 (2^22 samples)
    (10, 256, [(1_000_000, "resamp_any_kernel"), ((1 << 22) + 10, "fir_fft_kernel")]),        # (round 4 table: the 4096-point overlap-save form; the rules alone say the one-wave 1024-point form)
    (2, 64, [(1_000_000, "fir_core_kernel"), ((1 << 22) + 2, "fir_core_kernel")]),             # (table: the general direct kernel ahead of the strided-window one at 10^6)
    (1, 64, [(1_000_000, "fir_core_kernel"), ((1 << 22) + 1, "fir_core_kernel")]),             # (table: direct form ahead of the one-wave overlap-save form at 64 taps)
    (16, 200, [(1_000_000, "decim_mfma_real_kernel"), ((1 << 22) + 16, "decim_win_kernel")]),  # (round 4: the MFMA decimator on float rows; the table hands this 2^22 cell to the window kernel)
    (8, 128, [(1_000_000, "decim_win_kernel"), (1 << 26, "pfb_dec8_real_kernel")]),            # (round 4: two real segments per set of polyphase transforms on chip-filling calls)
    (4, 256, [(1 << 26, "pfb_dec4_real_kernel")]),
]


@pytest.mark.parametrize("plan", REAL_DIRECTED, ids=lambda p: f"real_{p[0]}_{p[1]}")
def test_directed_real_plans_default_thresholds(ops, plan):
    """Real data (src/dsp/filter.h:58-62, src/dsp/resampling.h:113-118: the float branches) under the shipped thresholds."""
    M, ntaps, calls = plan
    assert _no_overrides()
    taps = _lp(ntaps, 0.45 / M)
    if M == 1:
        op, orc = ops.Fir(taps, complex_data=False, max_block=0), O.Fir(taps, complex_data=False, acc=O.ACC_F64)
    else:
        op, orc = ops.Resampler(taps, 1, M, complex_data=False, max_block=0), O.Resampler(taps, 1, M, complex_data=False, acc=O.ACC_F64)
    rng = np.random.default_rng(ntaps + M)
    names = []
    for count, family in calls:
        x = rng.standard_normal(count).astype(np.float32)
        got = op.process(_dev(x)).cpu().numpy()
        name = op.last_kernel()["name"]
        names.append(name)
        want = orc.process(x)
        assert got.shape == want.shape and _rel(got, want) < TOL, (plan, count, name)
        _SEEN.add(name)
    assert names == [f for _, f in calls], (plan, names)


def test_directed_real_rational_plans_default_thresholds(ops):
    """PolyphaseResampler<float> with interp > 1 under the shipped thresholds (round 4: resamp_mfma_real_kernel and its real-data rules): the
    decimating side of a small ratio at a reference-sized call, 48 kHz -> 44.1 kHz on both sides of its size rule, a pure interpolator below its own."""
    assert _no_overrides()
    for L, M, tpp, calls in ((3, 8, 16, [(1_000_000 // 8 * 8, "resamp_mfma_real_kernel")]),
                             (147, 160, 16, [(160 * 6000, "resamp_any_kernel"), (160 * 52500, "resamp_mfma_real_kernel")]),
                             (6, 1, 16, [(1_000_000, "resamp_any_kernel")])):
        taps = (_lp(L * tpp, 0.45 / max(L, M)) * L).astype(np.float32)
        op, orc = ops.Resampler(taps, L, M, complex_data=False, max_block=0), O.Resampler(taps, L, M, complex_data=False, acc=O.ACC_F64)
        rng = np.random.default_rng(L * 100 + M)
        names = []
        for count, _ in calls:
            x = rng.standard_normal(count).astype(np.float32)
            got = op.process(_dev(x)).cpu().numpy()
            names.append(op.last_kernel()["name"])
            want = orc.process(x)
            assert got.shape == want.shape and _rel(got, want) < TOL, (L, M, count, names)
            _SEEN.add(names[-1])
        assert names == [f for _, f in calls], (L, M, names)


def test_directed_channel_banks_default_thresholds(ops):
    """Splitter -> N x VFO (src/dsp/routing.h:47-57 + src/dsp/vfo.h:19-36): the uniform 64-channel plan, and arbitrary offsets on
    both sides of the batch kernels' work threshold."""
    # uniform plan: (c - 31.5) fs / 64, decimation 64
    taps = _lp(256, 1.0 / 16.0)       # (the parity tests' 256 taps: FP32 rounding is relative to what goes IN, so a 1/128-band filter,
                                      # whose output is 18 dB below its input, would need the input-referred floor fuzz_dispatch uses)
    incs = [ops.phase_delta(1.0, -(c - 31.5) / 64.0) for c in range(64)]
    ch = ops.Channelizer(taps, 1, 64, incs, max_block=0)
    x = O.synth_iq(0, 64 * 3000, seed=9)
    y = ch.process(_dev(x)).cpu().numpy()
    assert ch.last_kernel()["name"] == "chan_uniform_kernel"
    _SEEN.add("chan_uniform_kernel")
    for c in (0, 17, 63):
        want = O.Resampler(taps, 1, 64, acc=O.ACC_F64).process(O.Xlator(1.0, -(c - 31.5) / 64.0, exact=True, volk_gain=True).process(x))
        assert _rel(y[c], want) < 4e-6, c
    # arbitrary offsets, 401 taps / 50: 4 channels x 100 000 samples (general batch kernel), 8 x 600 000 (MFMA batch kernel)
    taps = _lp(401, 0.45 / 50)
    for nch, count, family in ((4, 100_000, "resamp_any_batch_kernel"), (8, 600_000, "decim_mfma_batch_kernel")):
        freqs = [(-0.4 + 0.8 * c / nch) for c in range(nch)]
        ch = ops.Channelizer(taps, 1, 50, [ops.phase_delta(1.0, f) for f in freqs], max_block=0)
        x = O.synth_iq(0, count, seed=nch)
        y = ch.process(_dev(x)).cpu().numpy()
        assert ch.last_kernel()["name"] == family, (nch, count, ch.last_kernel()["name"])
        _SEEN.add(family)
        for c in (0, nch - 1):
            want = O.Resampler(taps, 1, 50, acc=O.ACC_F64).process(O.Xlator(1.0, freqs[c], exact=True, volk_gain=True).process(x))
            assert y[c].shape == want.shape and _rel(y[c], want) < TOL, (nch, c)


def test_random_slice_default_thresholds_covers_every_family(ops):
    """A fixed NUMBER of cases of scripts/fuzz_dispatch.py (seed fixed; a count, not a wall-clock budget, so a slower box runs the same
    cases) with no threshold touched, then: every kernel family of the dispatch has been exercised.  Self-contained: the
    directed plans above are run from here when this test is selected on its own (-k, xdist) and their families are not on record yet."""
    import fuzz_dispatch

    assert _no_overrides()
    if not _SEEN:
        for plan in DIRECTED:
            test_directed_plans_default_thresholds(ops, plan)
        for plan in REAL_DIRECTED:
            test_directed_real_plans_default_thresholds(ops, plan)
        test_directed_real_rational_plans_default_thresholds(ops)
        test_directed_channel_banks_default_thresholds(ops)
    n, worst, kernels = fuzz_dispatch.run(600.0, 20260403, default_only=True, verbose=False, max_cases=int(os.environ.get("QDSP_TEST_FUZZ_CASES", "1500")))
    assert n >= 20 and worst < TOL, (n, worst, kernels)
    seen = _SEEN | set(kernels)
    print(f"fuzz slice: {n} cases, worst {worst:.2e}; families seen by this module: {sorted(seen)}")
    missing = FAMILIES - seen
    assert not missing, f"kernel families never reached under the default thresholds: {sorted(missing)}"
