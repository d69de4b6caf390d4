"""The AUTO choice of FIR<complex_t> against its alternatives, re-measured.

qdsp_amd/csrc/dispatch_table.inc names, per (call size, tap count) cell, the fastest of the four kernel families as measured by
scripts/sweep_fir_table.py (profiles/r04_sweep_fir_table.txt).  This test times a FIXED sample of 40 shapes -- grid cells and shapes between
them -- with the default dispatch and with every family forced (QDSP_HIP_FIR_PICK), and fails when the default is more than 10 % slower
than the best alternative (round 3's if-chain was 12-15 % behind on two cells of profiles/r03_sweep_fir_mid.txt).  Results are also checked
to agree between the families (same filter: the direct forms bit for bit, the overlap-save forms to 2e-6)."""
import numpy as np
import pytest

import oracle as O
from conftest import rel_rms

pytestmark = pytest.mark.gpu

EXPECT = {1: ("fir_lat_kernel",), 2: ("fir_core_kernel",), 3: ("fir_fft1k_kernel",), 4: ("fir_fft_dma_kernel", "fir_fft_kernel", "fir_fft_dmapk_kernel")}


def _shapes():
    rng = np.random.default_rng(20260405)
    grid = [(1 << lg, t) for lg in (14, 16, 18, 19, 20, 21, 22, 23, 24) for t in (16, 32, 64, 256)][:24]
    grid += [(1 << 20, 16), (1 << 20, 32)]                      # the two cells round 3 lost
    between = []
    while len(between) < 14:
        n = int(2 ** rng.uniform(14, 24.5))
        t = int(2 ** rng.uniform(3, 9.5))
        between.append((n - n % 2, max(8, t)))
    return (grid + between)[:40]


def test_default_fir_dispatch_is_within_10_percent_of_the_best_family():
    import torch

    from bench import lowpass_taps
    from qdsp_amd import capi, ops

    worst, report = 0.0, []
    warm = ops.Fir(lowpass_taps(256, 0.2), max_block=0)
    xw = ops.synth_iq(1 << 24, seed=1)
    ow = torch.empty((1 << 24) + 8, dtype=torch.complex64, device="cuda")
    warm.time_dev(xw, ow, 200)                                   # clocks up before anything is compared
    warm.close()
    del xw, ow
    for n, nt in _shapes():
        taps = lowpass_taps(nt, 0.2)
        x = ops.synth_iq(n, seed=5)
        out = torch.empty(n + 8, dtype=torch.complex64, device="cuda")
        reps = max(5, min(100, int(1e-2 / max(2e-6, n * nt * 2.5e-13))))
        cands = [0] + [p for p in (1, 2, 3, 4) if not (p <= 2 and n * nt > (1 << 31))]
        op = ops.Fir(taps, max_block=0)                          # (QDSP_HIP_FIR_PICK is read per call: one handle serves every candidate)
        times, names, y0 = {}, {}, None
        try:
            for rnd in range(4):                                 # interleaved rounds, minimum per candidate: calls of 3-10 us are noisy
                for pick in cands:
                    capi.setenv("QDSP_HIP_FIR_PICK", str(pick) if pick else None)
                    op.reset()                                   # (same zero history for every candidate's checked call)
                    op.process(x, out)
                    name = op.last_kernel()["name"]
                    if pick and name not in EXPECT[pick]:
                        continue                                 # the family declined the shape
                    if rnd == 0:
                        y = out[: min(n, 20000)].cpu().numpy().copy()
                        if pick == 0:
                            y0 = y
                        else:
                            assert rel_rms(y, y0) < 2e-6, (n, nt, pick)
                    op.time_dev(x, out, max(3, reps // 4))
                    t = op.time_dev(x, out, reps)
                    times[pick] = min(times.get(pick, 1e9), t)
                    names[pick] = name
        finally:
            capi.setenv("QDSP_HIP_FIR_PICK", None)
            op.close()
        best = min(t for p, t in times.items() if p)
        ratio = times[0] / best
        worst = max(worst, ratio)
        report.append(f"{n:9d} x {nt:4d} taps: default {names[0]} {times[0] * 1e3:7.1f} us, best alternative {best * 1e3:7.1f} us ({ratio:.3f})")
        assert ratio <= 1.10 or times[None if None in times else 0] - best < 0.6e-3, "\n".join(report[-3:])   # (0.6 us: what two timings of ONE 3 us kernel differ by)
    print("\n".join(report))
    print(f"worst default / best = {worst:.3f}")
    torch.cuda.synchronize()


def test_table_rows_are_reachable_and_forced_picks_fall_back_when_they_cannot_serve():
    """A forced family that cannot serve the shape (1024 taps for the one-wave overlap-save form) hands the call back to the rule chain;
    QDSP_HIP_NO_FIR_TABLE=1 restores the round-3 chain; explicit modes (FIR_DIRECT / FIR_FFT) are not touched by the table."""
    import torch

    from bench import lowpass_taps
    from qdsp_amd import capi, ops

    x = ops.synth_iq(1 << 16, seed=6)
    xh = x.cpu().numpy()
    taps = lowpass_taps(1024, 0.2)
    capi.setenv("QDSP_HIP_FIR_PICK", "3")
    try:
        f = ops.Fir(taps, max_block=0)
        y = f.process(x).cpu().numpy()
        assert f.last_kernel()["name"] != "fir_fft1k_kernel"
        assert rel_rms(y, O.Fir(taps, acc=O.ACC_F64).process(xh)) < 2e-6
    finally:
        capi.setenv("QDSP_HIP_FIR_PICK", None)
    t64 = lowpass_taps(64, 0.2)
    d = ops.Fir(t64, max_block=0)
    d.set_mode(d.DIRECT)
    yd = d.process(x).cpu().numpy()
    assert d.last_kernel()["name"] == "fir_core_kernel" and np.array_equal(yd, O.Fir(t64, acc=O.ACC_FMA).process(xh))
    capi.setenv("QDSP_HIP_NO_FIR_TABLE", "1")
    try:
        a = ops.Fir(t64, max_block=0)
        ya = a.process(x).cpu().numpy()
        assert rel_rms(ya, yd) < 2e-6
    finally:
        capi.setenv("QDSP_HIP_NO_FIR_TABLE", None)
    torch.cuda.synchronize()


def _decim_shapes():
    rng = np.random.default_rng(20260406)
    grid = [(rot, M, 1 << lg, t) for rot in (0, 1) for M, lg, t in ((10, 13, 1024), (16, 15, 1024), (10, 14, 256), (8, 24, 64), (3, 20, 128), (2, 22, 256), (4, 18, 64), (8, 20, 256),
                                                                    (5, 16, 32), (8, 26, 256), (4, 26, 256), (16, 22, 128))]
    between = []
    while len(between) < 16:
        M = int(rng.choice([2, 3, 4, 5, 8, 10, 16]))
        n = int(2 ** rng.uniform(12, 25))
        t = int(2 ** rng.uniform(4, 10))
        between.append((int(rng.integers(0, 2)), M, n - n % M, max(8, t)))
    # the VFO's usual decimations (20 ... 100, swept later in round 4): cells where the rules alone lose 1.9-2.0x, the 2.4 Msps -> 48 kHz shape, in-between shapes
    big = [(0, 64, 1 << 15, 1024), (1, 25, 1 << 14, 1024), (0, 50, 1 << 13, 1024), (1, 50, 1000000, 401), (0, 32, 1 << 16, 768), (1, 100, 1 << 22, 801), (0, 20, 1 << 20, 200),
           (1, 64, 1 << 24, 512)]
    # decimations the sweep does NOT hold: from 10 on they read the row of the nearest swept one, below that the rules alone (decim_table_setting, qdsp_hip.hip)
    unswept = [(0, 12, 3 << 12, 1024), (1, 40, 5 << 11, 1024), (0, 6, 3 << 19, 128), (1, 7, 7 << 18, 64), (0, 14, 7 << 11, 896), (0, 24, 3 << 20, 256), (1, 80, 5 << 18, 640), (0, 48, 3 << 13, 768),
               (1, 13, 13 << 16, 208)]
    return (grid + between)[:40] + big + unswept


def test_default_decimator_dispatch_is_within_10_percent_of_the_best_setting():
    """Integer decimators and the fused VFO (complex data): the rule chain + the measured exception table (qdsp_amd/csrc/decim_table.inc <-
    profiles/r04_sweep_decim_table.txt) against the eight switch settings of the sweep, on 57 fixed shapes (cells where round 3's rules lost
    30-45 %, grid cells, shapes in between).  The default must be within 10 % of the best setting; results must agree between settings."""
    import torch

    from bench import lowpass_taps
    from qdsp_amd import capi, ops

    inc = ops.phase_delta(1.0, 0.1234)
    worst, report = 0.0, []
    for rot, M, n, nt in _decim_shapes():
        taps = lowpass_taps(nt, 0.45 / M)
        x = ops.synth_iq(n, seed=7)
        out = torch.empty(n // M + 8, dtype=torch.complex64, device="cuda")
        op = ops.Vfo(taps, 1, M, inc, max_block=0) if rot else ops.Resampler(taps, 1, M, max_block=0)
        times, names, y0 = {}, {}, None
        try:
            for rnd in range(3):
                seen = set()
                for k in [None] + list(range(9)):                # None = the shipped default (rules + table); 0 = rules only; 1..8 settings
                    capi.setenv("QDSP_HIP_DECIM_SETTING", None if k is None else str(k))
                    if k == 4 and n * nt // M > (1 << 30):
                        continue
                    op.reset()
                    op.process(x, out)
                    name = op.last_kernel()["name"]
                    if k is not None and name in seen and k != 0:
                        continue
                    if k is not None:
                        seen.add(name)
                    if rnd == 0:
                        y = out[: min(n // M, 20000)].cpu().numpy().copy()
                        if k is None:
                            y0 = y
                        else:
                            assert rel_rms(y, y0) < 4e-6, (rot, M, n, nt, k, name)
                    work = n * (nt / M if name in ("fir_core_kernel", "decim_win_kernel", "resamp_any_kernel") else 16)
                    reps = max(3, min(100, int(5e-3 / max(3e-6, work * 2.5e-13))))
                    op.time_dev(x, out, max(2, reps // 4))
                    t = op.time_dev(x, out, reps)
                    times[k] = min(times.get(k, 1e9), t)
                    names[k] = name
        finally:
            capi.setenv("QDSP_HIP_DECIM_SETTING", None)
            op.close()
        best = min(t for k, t in times.items() if k is not None)
        ratio = times[None] / best
        worst = max(worst, ratio)
        report.append(f"rot {rot} M {M:2d} {n:9d} x {nt:4d} taps: default {names[None]} {times[None] * 1e3:7.1f} us, rules only {times[0] * 1e3:7.1f}, best {best * 1e3:7.1f} ({ratio:.3f})")
        assert ratio <= 1.10 or times[None if None in times else 0] - best < 0.6e-3, "\n".join(report[-3:])   # (0.6 us: what two timings of ONE 3 us kernel differ by)
    print("\n".join(report))
    print(f"worst default / best = {worst:.3f}")
    torch.cuda.synchronize()


def test_default_real_data_dispatch_is_within_10_percent_of_the_best_setting():
    """FIR<float> and the integer decimators on real data (class 2 of qdsp_amd/csrc/decim_table.inc): same check as above on 24 shapes --
    cells where the rules alone lose up to 2.3x (decimation 10 / 16, 384-1024 taps, short calls), grid cells, shapes in between."""
    import numpy as np
    import torch

    from bench import lowpass_taps
    from qdsp_amd import capi, ops

    shapes = [(10, 1 << 14, 1024), (16, 1 << 14, 1024), (10, 1 << 13, 384), (16, 1 << 17, 1024), (10, 1 << 12, 512), (1, 1 << 20, 64),
              (1, 1 << 24, 256), (2, 1 << 22, 128), (3, 3 << 18, 96), (4, 1 << 24, 64), (5, 5 << 16, 256), (8, 1 << 26, 128),
              (25, 25 << 9, 1024), (32, 1 << 14, 1024), (20, 5 << 15, 1024), (50, 1000000, 401),
              (12, 3 << 12, 1024), (40, 5 << 12, 768), (6, 3 << 20, 128), (7, 7 << 15, 512),      # (these four: unswept decimations, nearest row)
              (50, 25 << 21, 401), (16, 1 << 26, 256), (100, 25 << 20, 801)]                        # (chip-filling calls of decim_mfma_real_kernel's range)
    rng = np.random.default_rng(4)
    for _ in range(12):
        M = int(rng.choice([1, 2, 3, 4, 5, 8, 10, 16]))
        n = int(2 ** rng.uniform(12, 25))
        shapes.append((M, max(M, n - n % M), max(8, int(2 ** rng.uniform(4, 10)))))
    worst, report = 0.0, []
    for M, n, nt in shapes:
        taps = lowpass_taps(nt, 0.45 / M)
        x = torch.view_as_real(ops.synth_iq(n, seed=9))[:, 0].contiguous()
        out = torch.empty(n // M + 8, dtype=torch.float32, device="cuda")
        op = ops.Fir(taps, complex_data=False, max_block=0) if M == 1 else ops.Resampler(taps, 1, M, complex_data=False, max_block=0)
        times, names, y0 = {}, {}, None
        try:
            for rnd in range(3):
                seen = set()
                for k in [None] + list(range(9)):
                    capi.setenv("QDSP_HIP_DECIM_SETTING", None if k is None else str(k))
                    if k == 4 and n * nt // M > (1 << 30):
                        continue
                    op.reset()
                    op.process(x, out)
                    name = op.last_kernel()["name"]
                    if k is not None and name in seen and k != 0:
                        continue
                    if k is not None:
                        seen.add(name)
                    if rnd == 0:
                        y = out[: min(n // M, 20000)].cpu().numpy().copy()
                        if k is None:
                            y0 = y
                        else:
                            assert rel_rms(y, y0) < 4e-6, (M, n, nt, k, name)
                    work = n * (nt / M if name in ("fir_core_kernel", "decim_win_kernel", "resamp_any_kernel") else 16)
                    reps = max(3, min(100, int(5e-3 / max(3e-6, work * 1.3e-13))))
                    op.time_dev(x, out, max(2, reps // 4))
                    times[k] = min(times.get(k, 1e9), op.time_dev(x, out, reps))
                    names[k] = name
        finally:
            capi.setenv("QDSP_HIP_DECIM_SETTING", None)
            op.close()
        best = min(t for k, t in times.items() if k is not None)
        ratio = times[None] / best
        worst = max(worst, ratio)
        report.append(f"real M {M:2d} {n:9d} x {nt:4d} taps: default {names[None]} {times[None] * 1e3:7.1f} us, rules only {times[0] * 1e3:7.1f}, best {best * 1e3:7.1f} ({ratio:.3f})")
        assert ratio <= 1.10 or times[None if None in times else 0] - best < 0.6e-3, "\n".join(report[-3:])   # (0.6 us: what two timings of ONE 3 us kernel differ by)
    print("\n".join(report))
    print(f"worst default / best = {worst:.3f}")
    torch.cuda.synchronize()
