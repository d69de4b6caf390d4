#!/usr/bin/env python3
"""Register / scratch usage of every gfx950 kernel the library ships, from hipcc's own remarks.

    python scripts/resource_usage.py            # writes profiles/r04_resource_usage.txt

Each translation unit of qdsp_amd/csrc is compiled (device side only, no GPU needed) with
-Rpass-analysis=kernel-resource-usage and the flags the Makefile builds it with; the table lists, per kernel instantiation:
VGPRs, SGPRs, SGPRs spilled (to VGPR lanes), VGPRs spilled, scratch bytes per lane, waves per SIMD.  The header records a hash
of the sources the table was made from: tests/test_capi_cpu.py::test_no_kernel_uses_scratch fails when the table is stale or any
kernel has scratch > 0."""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qdsp_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "r04_resource_usage.txt")
UNITS = {"qdsp_hip": "", "chan_ops": "", "misc_ops": "", "fft_fir": "-fno-slp-vectorize", "fft1k_fir": "-fno-slp-vectorize", "chan": "-fno-slp-vectorize",
         "pfb_dec": "-fno-slp-vectorize", "mf_dec": "", "rm_resamp": "", "fir_lat": ""}


def source_hash() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hip.h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return [re.sub(r"^void ", "", n).split("(")[0] for n in out.strip().split("\n")]
    except Exception:  # noqa: BLE001
        return names


def dma_wait_check(asm: str):
    """Kernels that prefetch by LDS-DMA count their own vector-memory operations: `s_waitcnt vmcnt(N)` (inline asm) waits for the DMA and leaves
    the N stores issued after it in flight (fft_fir.hip, chan.hip).  N is written by hand; what hipcc emits is checked here:
    for every such N the kernel must contain N global stores in one stretch of adjacent basic blocks that hold stores and no vector-memory
    load -- the full tile / segment's store sequence (one block when the stores are unconditional; one block per store when each is
    predicated, as the FIR's overlap test makes them; hipcc also tail-merges the last store of two unrolled copies).  A split 16-byte store
    or a spilled register would change the count.  Returns [(kernel, N, ok)]."""
    out = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "global_load_lds" not in body:
            continue
        waits = sorted({int(n) for n in re.findall(r";;#ASMSTART\n\s*s_waitcnt vmcnt\((\d+)\)", body)} - {0})
        if not waits:
            continue
        runs, stores, loads = [], 0, 0
        for ln in body.splitlines():
            t = ln.strip()
            if not t or t.startswith(";"):
                continue
            if t.endswith(":") or t.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                runs.append((stores, loads))
                stores = loads = 0
            elif t.startswith(("global_store", "buffer_store", "scratch_store")):
                stores += 1
            elif t.startswith(("global_load", "buffer_load", "scratch_load", "global_atomic")):
                loads += 1
        runs.append((stores, loads))
        def found(n):
            for i in range(len(runs)):
                tot = 0
                for st, ld in runs[i:]:
                    if ld or not st:
                        break
                    tot += st
                    if tot == n:
                        return True
                    if tot > n:
                        break
            return False

        for n in waits:
            out.append((name, n, found(n)))
    return out


def main():
    rows, dma = [], []
    for unit, extra in UNITS.items():
        asm_path = f"/tmp/qdsp_ru_{unit}.s"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *extra.split(), "-S", "--cuda-device-only",
               "-Rpass-analysis=kernel-resource-usage", "-o", asm_path, f"{unit}.hip"]
        err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
        if os.path.exists(asm_path):
            dma += dma_wait_check(open(asm_path).read())
            os.remove(asm_path)
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: (?:Function Name: (\S+)|\s+(\S[^:]*): (\d+))", line)
            if not m:
                continue
            if m.group(1):
                cur = {"unit": unit, "name": m.group(1)}
                rows.append(cur)
            elif cur is not None:
                cur[m.group(2).strip()] = int(m.group(3))
    names = demangle([r["name"] for r in rows])
    with open(OUT, "w") as f:
        f.write(f"# sources {source_hash()}  (qdsp_amd/csrc: *.hip, *.hip.h)  -- scripts/resource_usage.py, hipcc -Rpass-analysis=kernel-resource-usage, gfx950\n")
        dnames = demangle([d[0] for d in dma])
        for (_, n, ok), dn in zip(dma, dnames):
            f.write(f"# dma-wait {'ok ' if ok else 'BAD'} s_waitcnt vmcnt({n}): {n} stores in adjacent load-free blocks {'found' if ok else 'NOT FOUND'} in {dn}\n")
        f.write(f"# {len(rows)} kernel instantiations; columns: unit, VGPRs, SGPRs, SGPRs spilled (to VGPR lanes), VGPRs spilled, scratch bytes/lane, waves/SIMD, LDS bytes (static), kernel\n")
        for r, n in zip(rows, names):
            f.write(f"{r['unit']:10s} {r.get('VGPRs', -1):4d} {r.get('TotalSGPRs', -1):4d} {r.get('SGPRs Spill', -1):4d} {r.get('VGPRs Spill', -1):4d} "
                    f"{r.get('ScratchSize [bytes/lane]', -1):5d} {r.get('Occupancy [waves/SIMD]', -1):2d} {r.get('LDS Size [bytes/block]', -1):6d}  {n}\n")
    bad = [n for r, n in zip(rows, names) if r.get("ScratchSize [bytes/lane]", 1) != 0 or r.get("VGPRs Spill", 1) != 0]
    bad += [f"dma-wait vmcnt({n}) of {k}" for k, n, ok in dma if not ok]
    print(f"{len(rows)} kernels -> {OUT}; with scratch or spilled VGPRs: {len(bad)}")
    for n in bad:
        print("  ", n)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
