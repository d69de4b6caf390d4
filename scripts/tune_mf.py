"""decim_mfma_kernel against whatever AUTO picks without it (QDSP_HIP_NO_MF=1), fused VFO and plain decimator, 2^27
samples; and its knobs (tiles in flight, outputs per wave task). Run on the GPU box: python scripts/tune_mf.py [knobs]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from qdsp_amd import ops

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
    os.environ["QDSP_HIP_MF_MIN_DECIM"] = "9"
    n = 1 << 27
    if "knobs" in sys.argv:
        for M, ntaps in [(50, 401), (64, 513), (40, 321), (16, 129), (100, 801)]:
            for vfo in (True, False):
                for depth in ("1", "2"):
                    for tmax in ("128", "256", "512"):
                        os.environ["QDSP_HIP_MF_DEPTH"] = depth
                        os.environ["QDSP_HIP_MF_TASK_MAX"] = tmax
                        us, frac, name = run(M, ntaps, vfo, n)
                        print(f"M={M} ntaps={ntaps} vfo={vfo} depth={depth} T={tmax} {name} {us:.1f} us frac={frac:.3f}", flush=True)
        sys.exit(0)
    print("| decim | taps | form | decim_mfma_kernel us | without: kernel | us |")
    print("|---|---|---|---|---|---|")
    for M in (9, 12, 14, 16, 20, 24, 32, 48, 50, 96, 128):
        for tpm in (1, 2, 4, 8, 16):
            ntaps = M * tpm - (tpm > 1)
            for vfo in (True, False):
                os.environ["QDSP_HIP_NO_MF"] = "0"
                us, frac, name = run(M, ntaps, vfo, n)
                os.environ["QDSP_HIP_NO_MF"] = "1"
                us0, frac0, name0 = run(M, ntaps, vfo, n)
                print(f"| {M} | {ntaps} | {'vfo' if vfo else 'decim'} | {us:.0f} ({frac:.2f}) | {name0} | {us0:.0f} ({frac0:.2f}) |", flush=True)
