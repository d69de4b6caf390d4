"""Randomised differential run: random plans (interp, decim, taps, NCO) and random block sequences through the default
dispatch and through the lifted-threshold dispatch, against the FP64-accumulating oracle.  Run on the GPU box:
python scripts/fuzz_dispatch.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from qdsp_amd import capi, ops

def rel_rms(a, b, floor=0.0): return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / max(np.mean(np.abs(b) ** 2), floor ** 2, 1e-30)))

def floor_of(x, taps):
    """-40 dB of a full-scale output: a call whose outputs are all start-up transient or stop band (a first block shorter than
    the filter's delay) is judged against this, not against its own 1e-6-sized outputs -- FP32 rounding of any form of the
    filter is relative to the inputs and taps that went in."""
    return 1e-2 * float(np.sum(np.abs(taps))) * float(np.sqrt(np.mean(np.abs(x) ** 2))) if len(x) else 0.0


KNOBS = ("QDSP_HIP_NO_FFT1K_REAL", "QDSP_HIP_NO_FFT1K", "QDSP_HIP_MF_BATCH_MIN_WORK", "QDSP_HIP_MF_MIN_COUNT", "QDSP_HIP_RM_MIN_COUNT",
         "QDSP_HIP_RM_MIN_INTERP", "QDSP_HIP_NO_LM_SMALL_CALL_RULE", "QDSP_HIP_PFB_MIN_COUNT", "QDSP_HIP_DECIM_SETTING", "QDSP_HIP_FIR_PICK",
         "QDSP_HIP_NO_FIR_TABLE", "QDSP_HIP_NO_DECIM_TABLE")


def run(budget, seed, default_only=False, verbose=True, max_cases=None):
    """Random cases for `budget` seconds (max_cases: stop after that many cases instead -- a count does not depend on the box's speed); returns (cases, worst relative RMS error, {kernel family: cases}).  default_only: the
    library's own thresholds throughout (no QDSP_HIP_* variable is touched) -- what tests/test_gpu_fuzz.py runs; otherwise one
    case in two or three has a size rule switched off so that both sides of every crossover keep being exercised."""
    def knob(name, value):
        if not default_only:
            capi.setenv(name, value)

    if default_only:
        assert not any(k in os.environ for k in KNOBS), "default_only: the environment must not override the dispatch"
        capi.reload_env()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases, kernels, worst = 0, {}, 0.0
    t_note = time.time()
    while (time.time() < t_end) if max_cases is None else (n_cases < max_cases):
        if time.time() - t_note > 30:
            t_note = time.time()
            print(f"... {n_cases} cases, worst {worst:.2e}", flush=True) if verbose else None
        kind = rng.choice(["dec", "dec", "rat", "fir", "chan", "big", "real"])
        if kind == "real":
            # FIR<float> / PolyphaseResampler<float>: real samples, any ratio
            L = int(rng.choice([1, 1, 1, 2, 3, 24]))
            M = int(rng.choice([1, 2, 3, 5, 8, 16, 50, 125]))
            if np.gcd(L, M) != 1:
                continue
            fir = L == 1 and M == 1 and bool(rng.integers(0, 2))
            ntaps = int(rng.integers(1, 700)) if L == 1 else int(rng.integers(L, 40 * L))
            taps = (O.lowpass_taps_f64(ntaps, 0.45 / max(L, M)) * L).astype(np.float32) if ntaps > 2 else rng.standard_normal(ntaps).astype(np.float32)
            for k in ("QDSP_HIP_NO_FFT1K_REAL", "QDSP_HIP_NO_FFT1K"):
                knob(k, None)
            if rng.integers(0, 3) == 0:
                knob("QDSP_HIP_NO_FFT1K_REAL", "1")
            sizes = [int(rng.integers(0, 300_000)) for _ in range(int(rng.integers(1, 5)))]
            if rng.integers(0, 3) == 0:
                sizes[int(rng.integers(0, len(sizes)))] = int(rng.integers(0, 3 * M + 2))
            xr = np.ascontiguousarray(O.synth_iq(0, sum(sizes) + 1, seed=int(rng.integers(0, 1 << 30)))[:sum(sizes)].real)
            cuts = np.cumsum([0] + sizes)
            if fir:
                op, orc = ops.Fir(taps, complex_data=False, max_block=0), O.Fir(taps, complex_data=False, acc=O.ACC_F64)
            else:
                op, orc = ops.Resampler(taps, L, M, complex_data=False, max_block=0), O.Resampler(taps, L, M, complex_data=False, acc=O.ACC_F64)
            got, want, names = [], [], set()
            for a, b in zip(cuts, cuts[1:]):
                got.append(op.process(torch.from_numpy(xr[a:b]).cuda()).cpu().numpy())
                want.append(orc.process(xr[a:b]))
                if b > a:
                    names.add(op.last_kernel()["name"])
            got, want = np.concatenate(got), np.concatenate(want)
            err = rel_rms(got, want, floor_of(xr, taps)) if got.shape == want.shape and len(want) else (0.0 if got.shape == want.shape else float("inf"))
            for nm in names: kernels[nm] = kernels.get(nm, 0) + 1
            n_cases += 1
            worst = max(worst, err)
            if not err < 3e-6:
                msg = f"FAIL real fir={fir} L={L} M={M} ntaps={ntaps} sizes={sizes} kernels={names} err={err} shapes={got.shape}/{want.shape}"
                raise AssertionError(msg)
            continue
        if kind == "chan":
            # non-uniform channel bank (Splitter -> N x VFO): every channel against its own xlator -> resampler oracle
            M = int(rng.choice([8, 10, 16, 25, 50, 64, 100]))
            ntaps = int(rng.integers(M, 9 * M))
            nch = int(rng.integers(2, 20))
            taps = O.lowpass_taps_f64(ntaps, 0.45 / M).astype(np.float32)
            freqs = [float(rng.uniform(-0.45, 0.45)) for _ in range(nch)]
            knob("QDSP_HIP_MF_BATCH_MIN_WORK", str(int(rng.choice([0, 1 << 22]))))
            ch = ops.Channelizer(taps, 1, M, [ops.phase_delta(1.0, f) for f in freqs], max_block=0)
            sizes = [int(rng.integers(0, 300_000)) for _ in range(int(rng.integers(1, 4)))]
            x = O.synth_iq(0, sum(sizes) + 1, seed=int(rng.integers(0, 1 << 30)))[:sum(sizes)]
            cuts = np.cumsum([0] + sizes)
            ys = [ch.process(torch.from_numpy(x[a:b]).cuda()).cpu().numpy() for a, b in zip(cuts, cuts[1:])]
            names = {ch.last_kernel()["name"]}
            y = np.concatenate(ys, axis=1)
            for c in rng.choice(nch, size=min(nch, 3), replace=False):
                xl, rs = O.Xlator(1.0, freqs[c], exact=True, volk_gain=True), O.Resampler(taps, 1, M, acc=O.ACC_F64)
                want = np.concatenate([rs.process(xl.process(x[a:b])) for a, b in zip(cuts, cuts[1:])])
                err = rel_rms(y[c], want) if y[c].shape == want.shape and len(want) else (0.0 if y[c].shape == want.shape else float("inf"))
                worst = max(worst, err)
                if not err < 3e-6:
                    msg = f"FAIL chan M={M} ntaps={ntaps} nch={nch} c={c} sizes={sizes} kernels={names} err={err}"
                    raise AssertionError(msg)
            for nm in names: kernels[nm] = kernels.get(nm, 0) + 1
            n_cases += 1
            continue
        big = kind == "big"
        if big:
            kind = "dec"
        if kind == "fir":
            L, M = 1, 1
            ntaps = int(rng.integers(1, 400))
        elif kind == "dec":
            L, M = 1, int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 14, 16, 17, 24, 31, 32, 33, 50, 64, 100, 127, 128, 130, 200, 256, 300]))
            ntaps = int(rng.integers(1, min(34 * M, 5000)))
        else:
            L = int(rng.choice([2, 3, 5, 7, 10, 12, 16, 33, 48, 64, 100, 147, 160, 192, 200]))
            M = int(rng.choice([1, 2, 3, 5, 7, 8, 25, 49, 50, 147, 160, 175]))
            if np.gcd(L, M) != 1:
                continue
            ntaps = int(rng.integers(L, 40 * L))
        vfo = bool(rng.integers(0, 2)) and kind != "fir"
        lift = bool(rng.integers(0, 2))
        for k in ("QDSP_HIP_MF_MIN_COUNT", "QDSP_HIP_RM_MIN_COUNT", "QDSP_HIP_RM_MIN_INTERP", "QDSP_HIP_NO_FFT1K", "QDSP_HIP_NO_LM_SMALL_CALL_RULE", "QDSP_HIP_PFB_MIN_COUNT"):
            knob(k, None)
        if rng.integers(0, 3) == 0:
            knob("QDSP_HIP_NO_LM_SMALL_CALL_RULE", "1")     # (resamp_lm_kernel on the small calls the general kernel takes by default)
        if rng.integers(0, 3) == 0:
            knob("QDSP_HIP_NO_FFT1K", "1")     # (the 4096-point overlap-save kernels on the calls the one-wave form takes by default)
        if lift:
            knob("QDSP_HIP_MF_MIN_COUNT", "0")
            knob("QDSP_HIP_RM_MIN_COUNT", "0")
            knob("QDSP_HIP_PFB_MIN_COUNT", "4096")      # (the wave-per-segment polyphase kernels, decimation 8 and 4, on test-sized calls)
            if rng.integers(0, 2):
                knob("QDSP_HIP_RM_MIN_INTERP", "2")
        taps = (O.lowpass_taps_f64(ntaps, 0.45 / max(L, M)) * L).astype(np.float32) if ntaps > 2 else rng.standard_normal(ntaps).astype(np.float32)
        nblocks = int(rng.integers(1, 5))
        total_budget = int(3e6 / max(1, ntaps / max(L, 1) / 16))
        sizes = [int(rng.integers(0, max(2, min(total_budget, 400_000)))) for _ in range(nblocks)]
        if big and M >= 16 and ntaps <= 20 * M:
            # one call past the size thresholds of the MFMA kernels (3e6 / 1.6e7 samples)
            sizes[int(rng.integers(0, nblocks))] = int(rng.choice([3_200_000, 5_000_003, 17_000_000 if M >= 130 else 4_000_000]))
        if rng.integers(0, 3) == 0:
            sizes[int(rng.integers(0, nblocks))] = int(rng.integers(0, 3 * M + 2))
        x = O.synth_iq(0, sum(sizes) + 1, seed=int(rng.integers(0, 1 << 30)))[:sum(sizes)]
        cuts = np.cumsum([0] + sizes)
        blocks = [x[a:b] for a, b in zip(cuts, cuts[1:])]
        f = float(rng.uniform(-0.45, 0.45))
        if kind == "fir":
            op, orc = ops.Fir(taps, max_block=0), O.Fir(taps, acc=O.ACC_F64)
            want = np.concatenate([orc.process(b) for b in blocks]) if sum(sizes) else np.zeros(0, np.complex64)
        elif vfo:
            op = ops.Vfo(taps, L, M, ops.phase_delta(1.0, f), max_block=0)
            xl, rs = O.Xlator(1.0, f, exact=True, volk_gain=True), O.Resampler(taps, L, M, acc=O.ACC_F64)
            want = np.concatenate([rs.process(xl.process(b)) for b in blocks])
        else:
            op, rs = ops.Resampler(taps, L, M, max_block=0), O.Resampler(taps, L, M, acc=O.ACC_F64)
            want = np.concatenate([rs.process(b) for b in blocks])
        got, names = [], set()
        for b in blocks:
            got.append(op.process(torch.from_numpy(b).cuda()).cpu().numpy())
            if len(b):
                names.add(op.last_kernel()["name"])
        got = np.concatenate(got) if got else np.zeros(0, np.complex64)
        err = rel_rms(got, want, floor_of(x, taps)) if got.shape == want.shape and len(want) else (0.0 if got.shape == want.shape else float("inf"))
        tol = 3e-6
        for nm in names: kernels[nm] = kernels.get(nm, 0) + 1
        n_cases += 1
        worst = max(worst, err)
        if not err < tol:
            msg = f"FAIL kind={kind} L={L} M={M} ntaps={ntaps} vfo={vfo} lift={lift} env={dict((k, os.environ.get(k)) for k in ('QDSP_HIP_RM_MIN_INTERP',))} sizes={sizes} f={f} kernels={names} err={err} shapes={got.shape}/{want.shape}"
            raise AssertionError(msg)
    if not default_only:
        for k in KNOBS:
            capi.setenv(k, None)
    return n_cases, worst, kernels


if __name__ == "__main__":
    n_cases, worst, kernels = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    print(f"{n_cases} cases ok, worst rel rms {worst:.2e}, kernels {kernels}")
