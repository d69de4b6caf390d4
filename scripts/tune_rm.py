"""resamp_mfma_kernel (rational ratios on the MFMA units) against the FP64 oracle and against the kernels AUTO picks
without it (QDSP_HIP_NO_RM=1). Run on the GPU box: python scripts/tune_rm.py [--time-only]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from qdsp_amd import capi, ops

def dev(a): return torch.from_numpy(a).cuda()
def rel_rms(a, b): return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / max(np.mean(np.abs(b) ** 2), 1e-30)))

PLANS = [(10, 7, 8), (10, 7, 32), (10, 3, 16), (5, 6, 20), (5, 7, 20), (5, 8, 20), (4, 7, 20), (4, 5, 32), (3, 8, 20), (3, 5, 20), (2, 5, 20), (2, 7, 20), (5, 2, 20), (3, 4, 20), (3, 2, 32),
         (147, 160, 16), (160, 147, 16), (7, 5, 24), (24, 125, 12), (25, 24, 8), (6, 1, 10), (16, 15, 16), (17, 16, 3), (10, 3, 32), (441, 480, 8), (48, 50, 20), (8, 25, 40)]

def parity():
    bad = 0
    for L, M, tpp in PLANS:
        ntaps = L * tpp - 3
        taps = (O.lowpass_taps_f64(ntaps, 0.4 / max(L, M)) * L).astype(np.float32)
        sizes = [M * 700 + 17, 5, M * 300, M - 1 if M > 1 else 1, 3 * M + 1, 16 * M, 16 * M + 1, M * 515 + 3]
        x = O.synth_iq(0, sum(sizes), seed=L + M)
        cuts = np.cumsum([0] + sizes)
        blocks = [x[a:b] for a, b in zip(cuts, cuts[1:])]
        for vfo in (False, True):
            if vfo:
                op = ops.Vfo(taps, L, M, ops.phase_delta(1.0, 0.2345), max_block=0)
                xl, rs = O.Xlator(1.0, 0.2345, exact=True, volk_gain=True), O.Resampler(taps, L, M, acc=O.ACC_F64)
                want = np.concatenate([rs.process(xl.process(b)) for b in blocks])
            else:
                op = ops.Resampler(taps, L, M, max_block=0)
                rs = O.Resampler(taps, L, M, acc=O.ACC_F64)
                want = np.concatenate([rs.process(b) for b in blocks])
            got = np.concatenate([op.process(dev(b)).cpu().numpy() for b in blocks])
            name = op.last_kernel()["name"]
            err = rel_rms(got, want) if got.shape == want.shape else float("inf")
            ok = err < 2e-6
            bad += not ok
            print(f"L={L} M={M} ntaps={ntaps} vfo={vfo} kernel={name} shape={got.shape}/{want.shape} err={err:.3g} {'ok' if ok else 'FAIL'}", flush=True)
    return bad

def timing():
    n = 1 << 26
    print("| interp | decim | taps | form | resamp_mfma_kernel us (Gs/s out) | without: kernel | us (Gs/s out) |")
    print("|---|---|---|---|---|---|---|")
    for L, M, tpp in PLANS:
        ntaps = L * tpp - 3
        taps = (O.lowpass_taps_f64(ntaps, 0.4 / max(L, M)) * L).astype(np.float32)
        nin = n if L <= M else int(n * M / L)
        x = torch.view_as_complex(torch.randn(nin, 2, device="cuda"))
        for vfo in (False, True):
            row = []
            for norm in ("0", "1"):
                capi.setenv("QDSP_HIP_NO_RM", norm)
                capi.setenv("QDSP_HIP_NO_LM", "0" if norm == "1" else "1")      # (column 1: resamp_mfma_kernel also where resamp_lm_kernel would be picked)
                op = ops.Vfo(taps, L, M, ops.phase_delta(1.0, 0.2345), max_block=0) if vfo else ops.Resampler(taps, L, M, max_block=0)
                nout = nin * L // M
                out = torch.empty(nout + 8, dtype=torch.complex64, device="cuda")
                for _ in range(3): op.process(x, out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): op.process(x, out=out)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                row.append((op.last_kernel()["name"], ms * 1e3, nout / ms / 1e6, (nin + nout) * 8 / ms / 1e6 / 8000))
            print(f"| {L} | {M} | {ntaps} | {'vfo' if vfo else 'resampler'} | {row[0][1]:.0f} ({row[0][2]:.0f}, {row[0][3]:.2f}) | {row[1][0]} | {row[1][1]:.0f} ({row[1][2]:.0f}, {row[1][3]:.2f}) |", flush=True)
        del x

if __name__ == "__main__":
    os.environ.setdefault("QDSP_HIP_RM_MIN_INTERP", "2")     # (the default policy leaves interp < 33 to the general kernel)
    bad = 0
    if "--time-only" not in sys.argv: bad = parity()
    if not bad: timing()
    sys.exit(1 if bad else 0)
