"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/qdsp_hip.h
declares, and -- with no GPU -- refuses to do anything (no CPU fallback)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from qdsp_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = capi.load()
    names = capi.declared_symbols()
    assert len(names) >= 80
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.qdsp_hip_abi_version() == 1


def test_exported_symbols_are_all_declared():
    """Nothing leaks out of the .so beyond the header (C-ABI is exactly the header)."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and "qdsp_hip_" in l}
    assert exported == set(capi.declared_symbols())


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "qdsp_hip.h"\nint main(void){return QDSP_HIP_ABI_VERSION==1?0:1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "t")])
    subprocess.check_call([str(tmp_path / "t")])


def _has_gpu():
    n = C.c_int(0)
    return capi.load().qdsp_hip_device_count(C.byref(n)) == 0 and n.value > 0


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly():
    from qdsp_amd import ops

    with pytest.raises(capi.QdspHipError):
        ops.Fir(np.ones(4, np.float32))
    with pytest.raises(capi.QdspHipError):
        ops.Xlator(4.0, 1.0)
    assert b"no usable HIP device" in capi.load().qdsp_hip_error_string(-10004)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under qdsp_amd/ may reference it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "qdsp_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if "import oracle" in txt or "from oracle" in txt or "qdsp_oracle" in txt or "liboracle" in txt:
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_phase_delta_matches_reference_formula():
    from qdsp_amd.ops import phase_delta
    import oracle as O

    for fs, f in ((4.0, 1.0), (2.4e6, 123456.0), (48000.0, -7000.0)):
        d = O.Xlator(fs, f).delta
        assert phase_delta(fs, f) == (float(d[0]), float(d[1]))


def test_bench_starts_its_own_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` with no launcher in the environment starts N ranks through torch.distributed.run as a
    CHILD process (never an exec of itself) and returns their exit code; it does so before importing torch."""
    import importlib
    import subprocess
    import sys

    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index("launch_ranks(args)") < body.index("import torch")     # nothing GPU-related runs first


def test_no_kernel_uses_scratch():
    """profiles/r04_resource_usage.txt (scripts/resource_usage.py: hipcc's kernel-resource-usage remarks for every kernel instantiation
    of the shipped library) was made from the sources as they are now, and no kernel has scratch memory or spilled VGPRs
    (ten instantiations spilled)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("resource_usage", os.path.join(ROOT, "scripts", "resource_usage.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    lines = open(ru.OUT).read().splitlines()
    assert lines[0].split()[2] == ru.source_hash(), "qdsp_amd/csrc changed: run `python scripts/resource_usage.py` and commit the table"
    rows = [ln.split(None, 8) for ln in lines if ln and not ln.startswith("#")]
    assert len(rows) > 300
    # hand-counted `s_waitcnt vmcnt(N)` behind the LDS-DMA prefetches: N matches the stores hipcc emitted
    waits = [ln for ln in lines if ln.startswith("# dma-wait")]
    assert len(waits) >= 4 and all(ln.startswith("# dma-wait ok") for ln in waits), [ln for ln in waits if "BAD" in ln]
    bad = [r[8] for r in rows if int(r[4]) != 0 or int(r[5]) != 0]
    assert not bad, f"kernels with spilled VGPRs / scratch: {bad}"
    # the kernels of the bench legs keep the occupancy DESIGN.md quotes
    occ = {r[8]: int(r[6]) for r in rows}
    assert occ["qk::fir_fft_dma_kernel"] == 4 and occ["qk::pfb_dec8_kernel<true>"] == 2
    # (the channelizer's oversampled forms are held to three workgroups per CU by their LDS, whatever the registers allow; M = 64 needs four)
    chan = {k: v for k, v in occ.items() if k.startswith("qk::chan_uniform_kernel<")}
    assert len(chan) == 32 and all(v >= 3 for v in chan.values()) and all(v == 4 for k, v in chan.items() if ", 64, " in k)


def test_fir_dispatch_table_is_what_the_committed_sweep_gives():
    """qdsp_amd/csrc/dispatch_table.inc (FIR<complex_t>, AUTO: fastest kernel family per (call size, taps) cell) is generated from
    profiles/r04_sweep_fir_table.txt by scripts/gen_dispatch_table.py; an edited table or a new sweep without a regenerated table fails here
    (tests/test_gpu_dispatch.py re-measures a sample of cells on the GPU)."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_dispatch_table.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    inc = open(os.path.join(ROOT, "qdsp_amd", "csrc", "dispatch_table.inc")).read()
    assert "GENERATED" in inc and "kFirPick[kFirPickRows][kFirPickCols]" in inc
    dec = open(os.path.join(ROOT, "qdsp_amd", "csrc", "decim_table.inc")).read()      # the decimators' exception table, same generator, same check
    assert "GENERATED" in dec and "kDecimTab[kDecimTabClasses][kDecimTabMs][kDecimTabRows][kDecimTabCols]" in dec and "exception cells" in dec
