"""The N > 1 path with the HIP operators (BASELINE configs[3] and the 8-GPU leg of configs[4]) on
the one-GPU test box: 2 and 4 ranks share cuda:0, every rank runs the HIP kernel of its chunk behind
qdsp_amd.sharding.RingStream (the runner bench.py --gpus N uses), halos travel over gloo as CPU tensors
(RCCL refuses several ranks on one device); plus one rank as its own ring neighbour over real RCCL.
The concatenated outputs are compared with the UNSHARDED CPU oracle of the whole stream.
At most 4 worker processes + this one use the GPU at a time (the box allows 6)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from conftest import rel_rms

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_ring_gpu_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(case, backend, world, outdir, steps, n, extra_env=None):
    port = str(_free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, WORKER, case, backend, str(world), str(r), port, str(outdir), str(steps), str(n)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420)[0])
    finally:
        for p in procs:            # the exact processes started here, nothing else
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {case}/{backend}/{world} failed:\n{o[-3000:]}"


def _expected(case, x):
    from bench import lowpass_taps

    if case == "fir256":
        return O.Fir(lowpass_taps(256, 1 / 16), acc=O.ACC_F64).process(x)
    if case == "decim8":
        return O.Resampler(lowpass_taps(256, 1 / 16), 1, 8, acc=O.ACC_F64).process(x)
    xl = O.Xlator(1.0, 0.1234, exact=True, volk_gain=True)
    if case == "xlate_fir_decim8":
        return O.Resampler(lowpass_taps(256, 1 / 16), 1, 8, acc=O.ACC_F64).process(xl.process(x))
    if case == "vfo50":
        return O.Resampler(lowpass_taps(401, 0.4 / 50), 1, 50, acc=O.ACC_F64).process(xl.process(x))
    raise AssertionError(case)


def _collect(case, outdir, world, steps):
    return [np.load(os.path.join(outdir, f"{case}_{s}_{r}.npy")) for s in range(steps) for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case", ["fir256", "xlate_fir_decim8", "decim8"])
def test_ranks_on_one_gpu_match_unsharded_oracle(tmp_path, case, world):
    """configs[3] shape (256-tap FIR, chunk-sharded, 255-sample halo) and configs[2] sharded the same way."""
    steps, n = 2, 1 << 18           # (256 taps x 2^18 samples per chunk: past the small-call direct kernel)
    _run_ranks(case, "gloo", world, tmp_path, steps, n)
    y = np.concatenate(_collect(case, tmp_path, world, steps))
    x = O.synth_iq(0, steps * world * n, seed=4321)
    want = _expected(case, x)
    assert len(y) == len(want)
    assert rel_rms(y, want) < 2e-6
    # a wrong or missing halo is an O(1) error in the first 255 outputs of a chunk: check those windows on their own
    per = len(want) // (steps * world)
    for c in range(1, steps * world):
        w0 = c * per
        assert rel_rms(y[w0:w0 + 64], want[w0:w0 + 64]) < 1e-5, (case, world, c)
    # (chunks of 2^18 samples: the one-wave 1024-point overlap-save form; the 4096-point kernels take over from 3-4 million)
    assert open(os.path.join(tmp_path, f"{case}_kernel_0.txt")).read() == "fir_fft1k_kernel"


def test_large_decimation_vfo_on_two_ranks(tmp_path):
    """The VFO's everyday shape (401 taps, decimate by 50: decim_mfma_kernel) sharded over two ranks; chunks are
    multiples of lcm(50, 512) = 12800 samples (per-call phase restart and the VOLK gain cadence both line up)."""
    from qdsp_amd.sharding import chunk_alignment

    case, world, steps, n = "vfo50", 2, 2, 12800 * 8
    assert chunk_alignment(50, 1, 512) == 12800
    _run_ranks(case, "gloo", world, tmp_path, steps, n)
    y = np.concatenate(_collect(case, tmp_path, world, steps))
    x = O.synth_iq(0, steps * world * n, seed=4321)
    want = _expected(case, x)
    assert len(y) == len(want) and rel_rms(y, want) < 2e-6
    per = len(want) // (steps * world)
    for c in range(1, steps * world):
        assert rel_rms(y[c * per:c * per + 8], want[c * per:c * per + 8]) < 1e-5, c
    assert open(os.path.join(tmp_path, f"{case}_kernel_0.txt")).read() == "decim_mfma_kernel"
    with pytest.raises(AssertionError):
        _run_ranks(case, "gloo", world, tmp_path, steps, 1 << 17)     # 2^17 % 12800 != 0: RingStream refuses, the ranks exit non-zero


def test_channelizer_64_on_two_ranks(tmp_path):
    """The multi-GPU leg of configs[4]: time-sharded, all 64 channels on every rank, 256-sample raw halo."""
    from bench import lowpass_taps

    case, world, steps, n = "chan64", 2, 2, 1 << 16
    _run_ranks(case, "gloo", world, tmp_path, steps, n)
    ys = _collect(case, tmp_path, world, steps)
    y = np.concatenate(ys, axis=1)
    x = O.synth_iq(0, steps * world * n, seed=4321)
    taps = lowpass_taps(256, 1 / 128)
    assert y.shape == (64, len(x) // 64)
    assert open(os.path.join(tmp_path, f"{case}_kernel_0.txt")).read() == "chan_uniform_kernel"
    for c in (0, 7, 31, 32, 63):
        xl = O.Xlator(1.0, -(c - 31.5) / 64.0, exact=True, volk_gain=True)
        want = O.Resampler(taps, 1, 64, acc=O.ACC_F64).process(xl.process(x))
        # (1e-5 = the north_star bar: the uniform kernel applies each channel's NCO deviation from the 1/64 grid -- float
        # rounding of the caller's increments -- at the centre of the tap window; with this narrow 1/128 prototype, whose
        # response spans all 256 taps, the worst channel sits at 7e-6; the 1/16 prototype of test_channelizer_64_channels
        # stays under 4e-6)
        assert rel_rms(y[c], want) < 1e-5, c
        per = len(want) // (steps * world)
        for k in range(1, steps * world):
            assert rel_rms(y[c][k * per:k * per + 8], want[k * per:k * per + 8]) < 2e-5, (c, k)


@pytest.mark.parametrize("transport", ["c", "torch"])
@pytest.mark.parametrize("case", ["fir256", "xlate_fir_decim8"])
def test_rccl_self_ring_matches_unsharded_oracle(tmp_path, case, transport):
    """One rank as its own ring neighbour over real RCCL: the halo of step s+1 is this rank's own tail of step s, sent and
    received by the library's C ring (qdsp_hip_ring_post / _complete: ncclSend + ncclRecv on the ring's own stream) and
    prefetched under the kernel -- the transport the 8-GPU run uses, with world = 1 so the result is the plain unsharded stream.
    "torch": the same exchange through torch.distributed's send / recv (QDSP_RING_TRANSPORT=torch), the path all ranks fall back
    to together when the C ring cannot be had on every one of them."""
    steps, n = 4, 1 << 17
    _run_ranks(case, "nccl", 1, tmp_path, steps, n, {"QDSP_RING_TRANSPORT": transport})
    y = np.concatenate(_collect(case, tmp_path, 1, steps))
    x = O.synth_iq(0, steps * n, seed=4321)
    want = _expected(case, x)
    assert len(y) == len(want) and rel_rms(y, want) < 2e-6
    per = len(want) // steps
    for c in range(1, steps):
        assert rel_rms(y[c * per:c * per + 64], want[c * per:c * per + 64]) < 1e-5, (case, c)


def test_c_ring_entry_points_self_ring():
    """qdsp_hip_ring_* straight through ctypes, one rank as its own neighbour (include/qdsp_hip.h "ring"; what a C++ graph calls):
    what a post sends is what its complete delivers, `d_prev_halo` is the delivery of the post before (zeros before the
    first), a post reads the tail as of what the producer stream had queued, three posts outstanding are refused, and a
    256-tap FIR fed chunk by chunk with the halos from the ring equals the unsharded oracle (src/dsp/filter.h:51-74)."""
    import ctypes as C

    import torch

    from qdsp_amd import capi, ops

    L = capi.load()
    H = 255
    idbuf = (C.c_char * 128)()
    capi.check(L.qdsp_hip_ring_unique_id(idbuf))
    ring = C.c_void_p()
    capi.check(L.qdsp_hip_ring_create(C.byref(ring), 0, 0, 1, idbuf, H * 8))
    try:
        st = torch.cuda.current_stream().cuda_stream
        halo, prev = C.c_void_p(), C.c_void_p()

        def view(p):
            buf = torch.empty(H, dtype=torch.complex64, device="cuda")
            capi.check(L.qdsp_hip_memcpy_d2d(0, buf.data_ptr(), p, H * 8))
            return buf.cpu().numpy()

        tails = [torch.from_numpy(O.synth_iq(1000 * k, H, seed=7)).cuda() for k in range(4)]
        for k, tl in enumerate(tails):
            scratch = torch.zeros_like(tl)
            scratch.copy_(tl, non_blocking=True)          # queued on the producer stream just before the post: the post must see it
            capi.check(L.qdsp_hip_ring_post(ring, scratch.data_ptr(), st))
            capi.check(L.qdsp_hip_ring_complete(ring, st, C.byref(halo), C.byref(prev)))
            torch.cuda.synchronize()
            assert np.array_equal(view(halo.value), tl.cpu().numpy())
            want_prev = tails[k - 1].cpu().numpy() if k else np.zeros(H, np.complex64)
            assert np.array_equal(view(prev.value), want_prev)
        capi.check(L.qdsp_hip_ring_post(ring, tails[0].data_ptr(), st))
        capi.check(L.qdsp_hip_ring_post(ring, tails[1].data_ptr(), st))
        assert L.qdsp_hip_ring_post(ring, tails[2].data_ptr(), st) == -10001        # QDSP_HIP_EINVAL: two outstanding
        capi.check(L.qdsp_hip_ring_drain(ring))
        assert L.qdsp_hip_ring_complete(ring, st, None, None) == -10001             # nothing outstanding after the drain
        # a FIR over four chunks, every chunk a fresh handle whose history is what a NEW ring delivered (d_prev_halo of its first
        # complete is the stream's zero state)
        L.qdsp_hip_ring_destroy(ring)
        ring = C.c_void_p()
        capi.check(L.qdsp_hip_ring_unique_id(idbuf))
        capi.check(L.qdsp_hip_ring_create(C.byref(ring), 0, 0, 1, idbuf, H * 8))
        taps = O.lowpass_taps_f64(256, 1 / 16).astype(np.float32)
        n = 50_000
        x = O.synth_iq(0, 4 * n, seed=11)
        xd = torch.from_numpy(x).cuda()
        ys = []
        for k in range(4):
            chunk = xd[k * n:(k + 1) * n]
            capi.check(L.qdsp_hip_ring_post(ring, chunk.data_ptr() + (n - H) * 8, st))
            capi.check(L.qdsp_hip_ring_complete(ring, st, C.byref(halo), C.byref(prev)))
            f = ops.Fir(taps, max_block=0)
            f.set_history_ptr(prev.value, st)      # rank 0 of a one-rank ring: the tail of the chunk before (zeros at the start)
            ys.append(f.process(chunk).cpu().numpy())
            f.close()
        want = O.Fir(taps, acc=O.ACC_F64).process(x)
        assert rel_rms(np.concatenate(ys), want) < 2e-6
    finally:
        L.qdsp_hip_ring_destroy(ring)


# ---------------------------------------------------------------------------------------------
# bench.py's own multi-rank path, end to end: launcher -> ranks -> halo
# self-check -> max-over-ranks clock -> ONE JSON line that says by itself what carried the halo.
# ---------------------------------------------------------------------------------------------
def _run_bench(argv, extra_env, timeout=900):
    import json

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py")] + argv, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"bench.py {argv} failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal_end_to_end():
    """`python bench.py --gpus 2` with no launcher: it starts torch.distributed.run itself (a child, before touching the GPU),
    both ranks share cuda:0 (QDSP_BENCH_REHEARSE=1: halos over gloo, RCCL refuses two ranks on one device), every leg of the
    default line runs its halo self-check, and the line carries the `rccl` object."""
    line = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--log2n", "20", "--no-cpu-baseline", "--spinup-ms", "20"],
                      {"QDSP_BENCH_REHEARSE": "1"})
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["halo_samples"] == 255 and line["config"]["samples_per_gpu_per_step"] == 1 << 20
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert abs(line["value"] - 2 * (1 << 20) / (line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]
    rc = line["rccl"]
    assert rc["ring_ranks_requested"] == 2 and rc["halo_bytes"] == 2040
    assert rc["ring_transport"].startswith("gloo")                      # the rehearsal says so itself
    assert rc["halo_check_max_rel_err"] is not None and rc["halo_check_max_rel_err"] < 2e-6
    for leg, halo in (("chain", 256), ("channelizer", 256), ("channelizer_m8", 256)):
        assert line[leg]["config"]["halo_samples"] == halo, leg
        assert line[leg]["rccl"]["halo_check_max_rel_err"] < 2e-6, leg
    assert "block_call" not in line and "cpu_baseline" not in line      # N = 1 only


def test_bench_self_ring_reports_the_communicator():
    """One rank as its own ring neighbour over REAL RCCL, through bench.py: the `rccl` object is filled from the communicator
    (ncclCommCount == 1 here; == N on an N-GPU run) and the exchange is timed on the ring's stream."""
    line = _run_bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--log2n", "20", "--no-cpu-baseline", "--no-block-call", "--no-channelizer",
                       "--spinup-ms", "20"], {"QDSP_BENCH_SELF_RING": "1"})
    rc = line["rccl"]
    assert rc["ring_transport"].startswith("qdsp_hip_ring")
    assert rc["ring_comm_ranks"] == 1 and rc["ring_comm_ranks_min"] == 1 and rc["ring_comm_ranks_max"] == 1
    assert rc["ring_comm_device"] == 0 and rc["rccl_version"] > 0
    assert rc["exchanges_timed"] >= 3 and 0 < rc["exchange_us_mean"] <= rc["exchange_us_max"] < 5e4
    assert line["chain"]["rccl"]["ring_comm_ranks"] == 1
