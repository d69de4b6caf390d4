"""pfb_dec8_kernel (polyphase overlap-save, one wave per segment) against the oracle and against fir_fft_dec_kernel.
   python scripts/tune_pfb.py [check] [time]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle as O
from bench import lowpass_taps
from qdsp_amd import capi, ops


def rms(a, b):
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / np.mean(np.abs(b) ** 2)))


def check():
    capi.setenv("QDSP_HIP_PFB_MIN_COUNT", "0")
    x = O.synth_iq(0, 400_000, seed=5)
    xd = torch.from_numpy(x).cuda()
    inc = ops.phase_delta(1.0, 0.1234)
    for ntaps in (2, 17, 64, 255, 256, 257, 700, 1000):
        taps = lowpass_taps(ntaps, 1 / 16) if ntaps > 8 else np.arange(1, ntaps + 1, dtype=np.float32)
        for rot in (False, True):
            op = ops.Vfo(taps, 1, 8, inc, max_block=0) if rot else ops.Resampler(taps, 1, 8, max_block=0)
            op.set_mode(op.FFT)
            cuts = [0, 8 * 13001, 8 * 13001 + 8 * 9, 8 * 30000, 400_000]
            ys = [op.process(xd[a:b].contiguous()).cpu().numpy() for a, b in zip(cuts, cuts[1:])]
            k = op.last_kernel()["name"]
            y = np.concatenate(ys)
            xin = x
            rs = O.Resampler(taps, 1, 8, acc=O.ACC_F64)
            if rot:
                xl = O.Xlator(1.0, 0.1234, exact=True, volk_gain=True)
                want = np.concatenate([rs.process(xl.process(x[a:b])) for a, b in zip(cuts, cuts[1:])])
            else:
                want = np.concatenate([rs.process(x[a:b]) for a, b in zip(cuts, cuts[1:])])
            print(f"ntaps {ntaps:5d} rot {int(rot)} kernel {k:18s} rms {rms(y, want):.2e}  len {len(y)} {len(want)}", flush=True)


def timeit():
    n = 1 << 27
    x = ops.synth_iq(n, seed=1, device=0)
    out = torch.empty(n // 8, dtype=torch.complex64, device="cuda")
    inc = ops.phase_delta(1.0, 0.1234)
    taps = lowpass_taps(256, 1 / 16)
    for rot in (False, True):
        for pfb in (0, 1):
            capi.setenv("QDSP_HIP_NO_PFB", "0" if pfb else "1")
            for wg in ((2, 3, 4, 8) if pfb else (0,)):
                if wg:
                    capi.setenv("QDSP_HIP_PFB_WG_PER_CU", str(wg))
                op = ops.Vfo(taps, 1, 8, inc, max_block=0) if rot else ops.Resampler(taps, 1, 8, max_block=0)
                for _ in range(30):
                    op.process(x, out)
                torch.cuda.synchronize()
                ms = min(op.time_dev(x, out, 20) for _ in range(3))
                print(f"rot {int(rot)} pfb {pfb} wg/cu {wg}: {op.last_kernel()['name']:18s} {ms:.4f} ms  {9 * n / ms / 1e9:.1f} GB/s", flush=True)
                op.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    if "check" in what:
        check()
    if "time" in what:
        timeit()
