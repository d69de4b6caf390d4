"""decim_mfma_kernel against whatever AUTO picks without it (QDSP_HIP_NO_MF=1), fused VFO and plain decimator, 2^27
samples; and its knobs (tiles in flight, outputs per wave task). Run on the GPU box: python scripts/tune_mf.py [knobs]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from qdsp_amd import capi, ops

def run(M, ntaps, vfo, n, reps=10):
    taps = O.lowpass_taps_f64(ntaps, 0.4 / M).astype(np.float32)
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda"))
    op = ops.Vfo(taps, 1, M, ops.phase_delta(1.0, 0.2345), max_block=0) if vfo else ops.Resampler(taps, 1, M, max_block=0)
    out = torch.empty(n // M + 8, dtype=torch.complex64, device="cuda")
    for _ in range(3): op.process(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): op.process(x, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms * 1e3, (n * 8 * (1 + 1 / M)) / ms / 1e6 / 8000, op.last_kernel()["name"]

if __name__ == "__main__":
    capi.setenv("QDSP_HIP_MF_MIN_DECIM", "9")
    n = 1 << 27
    if "knobs" in sys.argv:
        for M, ntaps in [(50, 401), (64, 513), (40, 321), (16, 129), (100, 801)]:
            for vfo in (True, False):
                for depth in ("1", "2"):
                    for tmax in ("128", "256", "512"):
                        capi.setenv("QDSP_HIP_MF_DEPTH", depth)
                        capi.setenv("QDSP_HIP_MF_TASK_MAX", tmax)
                        us, frac, name = run(M, ntaps, vfo, n)
                        print(f"M={M} ntaps={ntaps} vfo={vfo} depth={depth} T={tmax} {name} {us:.1f} us frac={frac:.3f}", flush=True)
        sys.exit(0)
    if "ext" in sys.argv:
        # two tap sets (17-32 taps per column) and decimations 130-256 (rows of M / 2 samples, every other output kept)
        def rel_rms(a, b): return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / max(np.mean(np.abs(b) ** 2), 1e-30)))
        print("| decim | taps | form | decim_mfma_kernel us | without: kernel | us | err vs FP64 oracle |")
        print("|---|---|---|---|---|---|---|")
        for M, ntaps in [(16, 511), (24, 700), (50, 1201), (50, 1600), (100, 3200), (128, 2500), (130, 1041), (160, 1281), (200, 1601), (256, 2049), (256, 4096), (250, 4500)]:
            taps = O.lowpass_taps_f64(ntaps, 0.4 / M).astype(np.float32)
            xs = O.synth_iq(0, M * 4000 + 17, seed=M)
            for vfo in (True, False):
                capi.setenv("QDSP_HIP_NO_MF", "0")
                op = ops.Vfo(taps, 1, M, ops.phase_delta(1.0, 0.2345), max_block=0) if vfo else ops.Resampler(taps, 1, M, max_block=0)
                got = np.concatenate([op.process(torch.from_numpy(b).cuda()).cpu().numpy() for b in (xs[:M * 1500 + 5], xs[M * 1500 + 5:])])
                if vfo:
                    xl, rs = O.Xlator(1.0, 0.2345, exact=True, volk_gain=True), O.Resampler(taps, 1, M, acc=O.ACC_F64)
                    want = np.concatenate([rs.process(xl.process(b)) for b in (xs[:M * 1500 + 5], xs[M * 1500 + 5:])])
                else:
                    rs = O.Resampler(taps, 1, M, acc=O.ACC_F64)
                    want = np.concatenate([rs.process(b) for b in (xs[:M * 1500 + 5], xs[M * 1500 + 5:])])
                err = rel_rms(got, want) if got.shape == want.shape else float("inf")
                us, frac, name = run(M, ntaps, vfo, n)
                capi.setenv("QDSP_HIP_NO_MF", "1")
                us0, frac0, name0 = run(M, ntaps, vfo, n)
                print(f"| {M} | {ntaps} | {'vfo' if vfo else 'decim'} | {name} {us:.0f} ({frac:.2f}) | {name0} | {us0:.0f} ({frac0:.2f}) | {err:.2g} |", flush=True)
        sys.exit(0)
    print("| decim | taps | form | decim_mfma_kernel us | without: kernel | us |")
    print("|---|---|---|---|---|---|")
    for M in (9, 12, 14, 16, 20, 24, 32, 48, 50, 96, 128):
        for tpm in (1, 2, 4, 8, 16):
            ntaps = M * tpm - (tpm > 1)
            for vfo in (True, False):
                capi.setenv("QDSP_HIP_NO_MF", "0")
                us, frac, name = run(M, ntaps, vfo, n)
                capi.setenv("QDSP_HIP_NO_MF", "1")
                us0, frac0, name0 = run(M, ntaps, vfo, n)
                print(f"| {M} | {ntaps} | {'vfo' if vfo else 'decim'} | {us:.0f} ({frac:.2f}) | {name0} | {us0:.0f} ({frac0:.2f}) |", flush=True)
