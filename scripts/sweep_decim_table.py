#!/usr/bin/env python3
"""Integer decimators (PolyphaseResampler<complex_t>, interp 1; class 0), the fused VFO (class 1) and real data (FIR<float>, PolyphaseResampler<float>
with interp 1; class 2) per call: the rule chain of the AUTO dispatch (setting 0)
against eight switch settings, each forced with QDSP_HIP_DECIM_SETTING, on the grid the exception table is built on
(scripts/gen_dispatch_table.py -> qdsp_amd/csrc/decim_table.inc).

    python scripts/sweep_decim_table.py > profiles/r04_sweep_decim_table.txt        (on the GPU box, a few minutes)

Line format:  <class> <decim> <log2 count> <ntaps> <us setting 0> ... <us setting 8>   ("=j" : the setting ran the kernel setting j of the
row ran, "-" : not measured).  Settings (qdsp_hip.hip kDecimSettingVeto / kDecimSettingMode, same order): 0 rules, 1 no strided-window kernel,
2 no one-wave 1024-point kernel, 3 neither, 4 direct form, 5 overlap-save, 6 no polyphase wave-per-segment kernel, 7 overlap-save without it,
8 no MFMA decimator."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from qdsp_amd import capi, ops  # noqa: E402

MS = [2, 3, 4, 5, 8, 10, 16, 20, 25, 32, 50, 64, 100]      # (20 ... 100: the VFO's usual decimations, 2.4 Msps -> 48 kHz is 50)
TAPS = [16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024]
LOG2 = list(range(12, 28))
NSET = 9


def main():
    print("# scripts/sweep_decim_table.py: microseconds per call (min of 3 batches); settings 0..8 forced with QDSP_HIP_DECIM_SETTING")
    print("# rot decim log2n taps " + " ".join(f"s{k}" for k in range(NSET)))
    xall = ops.synth_iq(1 << LOG2[-1], seed=3)
    oall = torch.empty((1 << (LOG2[-1] - 1)) + 64, dtype=torch.complex64, device="cuda")
    inc = ops.phase_delta(1.0, 0.1234)
    xr = torch.view_as_real(xall)[:, 0].contiguous()          # real samples for class 2 (FIR<float> / PolyphaseResampler<float>)
    orl = torch.empty((1 << LOG2[-1]) + 64, dtype=torch.float32, device="cuda")
    classes = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2"])]
    only_ms = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else None      # (a partial sweep: merge its lines into the profile file)
    for rot in classes:
        for M in ([1] + MS if rot == 2 else MS):
            if only_ms is not None and M not in only_ms:
                continue
            for nt in TAPS:
                taps = bench.lowpass_taps(nt, 0.45 / M)
                if rot == 2:
                    op = ops.Fir(taps, complex_data=False, max_block=0) if M == 1 else ops.Resampler(taps, 1, M, complex_data=False, max_block=0)
                else:
                    op = ops.Vfo(taps, 1, M, inc, max_block=0) if rot else ops.Resampler(taps, 1, M, max_block=0)
                for lg in LOG2:
                    n = (1 << lg) // M * M
                    x, out = (xr[:n], orl[: n // M + 8]) if rot == 2 else (xall[:n], oall[: n // M + 8])
                    seen, cells = {}, []
                    for k in range(NSET):
                        capi.setenv("QDSP_HIP_DECIM_SETTING", str(k))
                        if k == 4 and n * nt // M > (1 << 31):
                            cells.append("-")
                            continue
                        op.process(x, out)
                        name = op.last_kernel()["name"]
                        if name in seen:
                            cells.append(f"={seen[name]}")
                            continue
                        work = n * (nt / M if name in ("fir_core_kernel", "decim_win_kernel", "resamp_any_kernel") else 16)
                        reps = max(3, min(100, int(5e-3 / max(3e-6, work * 2.5e-13))))
                        op.time_dev(x, out, max(2, reps // 4))
                        t = min(op.time_dev(x, out, reps) for _ in range(3))
                        seen[name] = k
                        cells.append(f"{t * 1000:.2f}")
                    capi.setenv("QDSP_HIP_DECIM_SETTING", None)
                    print(f"{rot} {M} {lg} {nt} " + " ".join(cells), flush=True)
                op.close()


if __name__ == "__main__":
    main()
