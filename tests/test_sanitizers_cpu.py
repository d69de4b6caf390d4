"""Host-side code under sanitizers, on the CPU (GPU AddressSanitizer / XNACK runs are not available on
the test pool and not wanted).

* the block-graph mirror (qdsp_amd/host/dsp: stream hand-offs, device-resident links, Splitter banks, live retunes) runs its
  graph_check harness against a TEST-ONLY fake libqdsp_hip (tests/fake_hip/fake_qdsp_hip.cpp: same C ABI, every operator a copy on
  the calling thread) built with -fsanitize=thread and with -fsanitize=address,undefined;
* qdsp_amd/csrc/knobs.cpp: readers against concurrent reloads under -fsanitize=thread;
* qdsp_amd/csrc/ring.cpp: the halo ring's buffer rotation against a fake synchronous HIP runtime + one-rank RCCL under
  -fsanitize=address,undefined.
Pass = every program exits 0 and no sanitizer report appears on stderr."""
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, "fake_hip")
HOST = os.path.join(ROOT, "qdsp_amd", "host")
SAN = {"thread": "thread", "address": "address,undefined"}
BASE = ["-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer"]
ENV = {
    "TSAN_OPTIONS": "halt_on_error=1 exitcode=66 second_deadlock_stack=1",
    "ASAN_OPTIONS": "detect_leaks=1 exitcode=67",
    "UBSAN_OPTIONS": "print_stacktrace=1 halt_on_error=1 exitcode=68",
}
REPORT = ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def cxx(args, cwd=FAKE):
    r = subprocess.run(["g++"] + args, cwd=cwd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.fixture(scope="module", params=sorted(SAN))
def graph(request):
    """graph_check + the fake library, both built with one sanitizer."""
    tag = request.param
    out = os.path.join(FAKE, "build", tag)
    os.makedirs(out, exist_ok=True)
    flags = BASE + [f"-fsanitize={SAN[tag]}"]
    cxx(flags + ["-fPIC", "-shared", "-o", os.path.join(out, "libqdsp_hip.so"), "fake_qdsp_hip.cpp"])
    cxx(flags + ["-Wall", "-pthread", f"-I{HOST}", f"-I{os.path.join(ROOT, 'include')}", "-o", os.path.join(out, "graph_check"),
                 os.path.join(HOST, "examples", "graph_check.cpp"), f"-L{out}", "-lqdsp_hip", "-Wl,-rpath,$ORIGIN"])
    return tag, os.path.join(out, "graph_check")


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(120_000) + 1j * rng.standard_normal(120_000)).astype(np.complex64)
    x.tofile(d / "x.cf32")
    np.ascontiguousarray(x.real).tofile(d / "x.f32")
    np.hanning(63).astype(np.float32).tofile(d / "t.f32")
    return d


def run_clean(cmd, cwd, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.update(ENV)
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
    hits = [ln for ln in r.stderr.splitlines() if any(k in ln for k in REPORT)]
    assert r.returncode == 0 and not hits, f"{' '.join(map(str, cmd))}: rc {r.returncode}\n{r.stderr[-4000:]}"
    return r.stdout


GRAPHS = [
    ["stream"],
    ["fir", "x.cf32", "y.cf32", "4096", "t.f32"],
    ["firf", "x.f32", "y.f32", "4096", "t.f32"],
    ["resampst", "x.cf32", "yst.cf32", "10000", "48000", "32000", "12000", "6000"],
    ["chain", "x.cf32", "yc.cf32", "10000", "t.f32", "2400000", "100000", "2400000", "240000"],
    ["math", "x.cf32", "ym.cf32", "10000", "mul", "2400000", "100000"],
    ["vfo", "x.cf32", "yv.cf32", "10000", "300000", "2400000", "240000", "200000"],
    ["split", "x.cf32", "ys", "10000", "4", "2400000", "240000", "200000"],
    ["splitretune", "x.cf32", "yr", "10000", "4", "2400000", "240000", "200000", "5", "300000"],
    ["splitretune", "x.cf32", "yc", "10000", "4", "2400000", "240000", "200000", "5", "0", "reconf"],
    ["splitretune", "x.cf32", "yb", "10000", "4", "2400000", "240000", "200000", "5", "0", "rebind"],
    ["mulsplit", "x.cf32", "ym", "10000", "4", "2400000", "240000", "200000"],
    ["sine", "ysin.cf32", "10000", "20", "2400000", "100000", "t.f32"],
    ["bench", "vfo", "100000", "40", "2400000", "240000"],
    ["bench", "chain", "100000", "40", "2400000", "240000"],
    ["bench", "split4", "100000", "30", "2400000", "240000"],
]


@pytest.mark.parametrize("args", GRAPHS, ids=[" ".join(a[:1] + a[-1:]) if a[0] == "splitretune" else a[0] + ("_" + a[1] if a[0] == "bench" else "") for a in GRAPHS])
def test_block_graph_is_clean_under_sanitizers(graph, files, args):
    tag, exe = graph
    out = run_clean([exe] + args, cwd=files, extra_env={"FAKE_HIP_JITTER": "1"})
    if args[0] not in ("stream", "bench"):
        assert "graph ok" in out or "ok" in out, out[-500:]


def test_knob_table_readers_against_reloads_under_tsan():
    out = os.path.join(FAKE, "build", "thread")
    os.makedirs(out, exist_ok=True)
    cxx(BASE + ["-pthread", "-fsanitize=thread", f"-I{os.path.join(ROOT, 'include')}", "-o", os.path.join(out, "knobs_race"), "knobs_race.cpp",
                os.path.join(ROOT, "qdsp_amd", "csrc", "knobs.cpp")])
    assert "knobs ok" in run_clean([os.path.join(out, "knobs_race")], cwd=out)


def test_halo_ring_bookkeeping_under_asan_ubsan():
    """qdsp_amd/csrc/ring.cpp with a fake HIP runtime and a fake one-rank librccl.so.1 in front of the real ones."""
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime.h")):
        pytest.skip("needs the HIP headers")
    out = os.path.join(FAKE, "build", "address")
    os.makedirs(out, exist_ok=True)
    flags = BASE + ["-fsanitize=address,undefined"]
    cxx(flags + ["-fPIC", "-shared", "-DFAKE_HIP", "-o", os.path.join(out, "libfakehip.so"), "fake_hip_runtime.cpp"])
    cxx(flags + ["-fPIC", "-shared", "-DFAKE_RCCL", "-o", os.path.join(out, "librccl.so.1"), "fake_hip_runtime.cpp"])
    cxx(flags + ["-D__HIP_PLATFORM_AMD__", f"-I{hip_inc}", f"-I{os.path.join(ROOT, 'include')}", "-o", os.path.join(out, "ring_selftest"),
                 "ring_selftest.cpp", os.path.join(ROOT, "qdsp_amd", "csrc", "ring.cpp"), f"-L{out}", "-lfakehip", "-ldl", "-Wl,-rpath,$ORIGIN"])
    env = {"LD_LIBRARY_PATH": out + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")}
    assert "ring ok" in run_clean([os.path.join(out, "ring_selftest")], cwd=out, extra_env=env)
