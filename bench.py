#!/usr/bin/env python3
"""bench.py -- Msamples/s of complex IQ through the FIR / decimator hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload fir256]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
(one rank per GPU, backend nccl == RCCL).  Started WITHOUT a launcher (`python bench.py --gpus N`, no WORLD_SIZE in
the environment) it starts those N ranks itself, as a child process and before anything here touches the GPU, and
exits with their code.

A "step" = one pass of the hot path over one batch: every rank runs its operator once over
its 2^LOG2N-sample chunk of a continuous synthetic IQ stream that is already resident in
HBM.  With N > 1 the stream is cut block-cyclically over the ranks and each step begins with
the ring-neighbour halo hand-off (the previous chunk's last H input samples -> this rank's
filter history, RCCL send/recv over xGMI: qdsp_amd/sharding.py RingStream, which the multi-rank tests
drive too); there is no other communication.  Per-GPU work is fixed, so scaling is "weak".

Workloads (BASELINE.json configs):
    fir256            configs[1]: 256-tap complex FIR                      (default; the metric)
    xlate_fir_decim8  configs[2]: fused NCO + 256-tap FIR + decimate-by-8
    decim8            256-tap decimate-by-8 (no NCO)
    xlate             NCO mixer alone
    fir63             63-tap FIR (the reference BlackmanWindow design of configs[0])

Rank 0 prints ONE JSON line (contract in the task statement) extended with
  "roofline":     dominant kernel vs the HBM roofline: algorithmic bytes per launch / mean
                  launch duration (HIP events on the launch stream, measured here)
  "roofline_fp32": the same launch vs the FP32 vector peak (the direct-form 256-tap FIR is
                  compute-bound: 1024 FLOP / 16 B per sample)
  "block_call":   N = 1 only: per-call time of the same operator on ONE reference-sized block (1e6 samples, the most the
                  reference's stream API hands over per call, stream.h:7), 200 back-to-back calls -- latency-bound, not
                  part of `value`.
  "chain":        default run only: the fused xlating-FIR + decimate-by-8 chain (BASELINE configs[2]) timed by the same
                  protocol right behind the FIR, with its own roofline object
  "cpu_baseline": the CPU oracle timed on this host's cores, N = 1 only: scalar VOLK-generic order on all worker
                  threads (`value`), a SIMD-lane FMA variant (`value_simd`), and one block-graph worker thread
                  source -> FIR -> NullSink (`value_1thread_graph`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHAN_ROW_PAD = 32           # samples between the end of a channel row and the start of the next (see run_workload)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
FP32_PEAK_TFLOPS = 157.3   # FP32 vector == FP32 MFMA peak (same file)

# algorithmic bytes / flops per INPUT sample
WORKLOADS = {
    "fir256": dict(ntaps=256, decim=1, rot=False, bytes=16.0, flops=1024.0),
    "fir63": dict(ntaps=63, decim=1, rot=False, bytes=16.0, flops=252.0),
    "decim8": dict(ntaps=256, decim=8, rot=False, bytes=9.0, flops=128.0),
    "xlate_fir_decim8": dict(ntaps=256, decim=8, rot=True, bytes=9.0, flops=134.0),
    "xlate": dict(ntaps=0, decim=1, rot=True, bytes=16.0, flops=6.0),
    # decimate by 4 (round 3: pfb_dec4_kernel): 8 B in + 8/4 B out
    "decim4": dict(ntaps=256, decim=4, rot=False, bytes=10.0, flops=256.0),
    "xlate_fir_decim4": dict(ntaps=256, decim=4, rot=True, bytes=10.0, flops=262.0),
    # short filters (the strided-window direct kernel)
    "decim8_t63": dict(ntaps=63, decim=8, rot=False, bytes=9.0, flops=31.5),
    "xlate_fir_decim8_t63": dict(ntaps=63, decim=8, rot=True, bytes=9.0, flops=37.5),
    # the reference VFO's usual shape (vfo.h: 2.4 Msps -> 48 kHz, ~8 taps per unit of decimation): 8 + 8/50 B
    "vfo50": dict(ntaps=401, decim=50, rot=True, bytes=8.16, flops=38.1, fc=0.4 / 50),
    # 48 kHz -> 44.1 kHz, 16 taps per phase: 8 B in + 8 * 147/160 B out, 4 * 16 * 147/160 FLOP per input sample
    "resamp147_160": dict(ntaps=147 * 16 - 3, decim=160, interp=147, rot=False, bytes=8.0 + 8.0 * 147 / 160, flops=4 * 16 * 147 / 160, fc=0.4 / 160),
    # BASELINE configs[4]: 64 channels at (c - 31.5) fs/64, 256 taps, decimate 64: 8 B in + 64*8/64 B out
    "chan64": dict(ntaps=256, decim=64, rot=True, bytes=16.0, flops=1408.0, nchan=64),
    # its oversampled variant (SURVEY 8d config 5 "and M = 8"): 8 B in + 64*8/8 B out per input sample
    "chan64m8": dict(ntaps=256, decim=8, rot=True, bytes=72.0, flops=16384.0, nchan=64),
    # REAL data (FIR<float>, PolyphaseResampler<float>: SURVEY 8f rank 3), single GPU only: 4 B in + 4 L / M B out per sample.  Round 4.
    "fir256_real": dict(ntaps=256, decim=1, rot=False, bytes=8.0, flops=512.0, real=True),
    "decim8_real": dict(ntaps=256, decim=8, rot=False, bytes=4.5, flops=64.0, real=True),
    "decim4_real": dict(ntaps=256, decim=4, rot=False, bytes=5.0, flops=128.0, real=True),
    "decim50_real": dict(ntaps=401, decim=50, rot=False, bytes=4.08, flops=16.0, real=True, fc=0.4 / 50),
    "resamp147_160_real": dict(ntaps=147 * 16 - 3, decim=160, interp=147, rot=False, bytes=4.0 + 4.0 * 147 / 160, flops=2 * 16 * 147 / 160, real=True, fc=0.4 / 160),
}


def lowpass_taps(ntaps: int, fc: float):
    """Harness taps: Blackman-windowed sinc designed in FP64, unit DC gain.
    The reference's own BlackmanWindow cannot produce an even tap count."""
    import numpy as np

    n = np.arange(ntaps, dtype=np.float64)
    c = (ntaps - 1) / 2.0
    h = 2 * fc * np.sinc(2 * fc * (n - c))
    w = 0.42 - 0.5 * np.cos(2 * np.pi * n / (ntaps - 1)) + 0.08 * np.cos(4 * np.pi * n / (ntaps - 1))
    h = h * w
    return (h / h.sum()).astype(np.float32)


def make_op(ops, name: str, device: int):
    w = WORKLOADS[name]
    if name == "xlate":
        return ops.Xlator(phase_inc=ops.phase_delta(1.0, 0.1234), device=device, max_block=0)
    if name in ("chan64", "chan64m8"):
        incs = [ops.phase_delta(1.0, -(c - 31.5) / 64.0) for c in range(64)]
        return ops.Channelizer(lowpass_taps(256, 1.0 / 128.0), 1, w["decim"], incs, device=device, max_block=0)
    taps = lowpass_taps(w["ntaps"], w.get("fc", 1.0 / 16.0)) if name != "fir63" else None
    if taps is None:
        import numpy as np

        # BlackmanWindow(cutoff=0.1 fs, transWidth=4 fs/63) of SURVEY 8d config 1, as designed by
        # the C++ host mirror; for the bench any fixed 63 taps do: use the FP64 design.
        taps = lowpass_taps(63, 0.1)
    L = w.get("interp", 1)
    if L > 1:
        taps = (taps * L).astype(taps.dtype)       # unit pass-band gain after the zero-stuffing
    if w["rot"]:
        return ops.Vfo(taps, L, w["decim"], ops.phase_delta(1.0, 0.1234), device=device, max_block=0)
    cplx = not w.get("real", False)
    if w["decim"] > 1 or L > 1:
        return ops.Resampler(taps, L, w["decim"], complex_data=cplx, device=device, max_block=0)
    return ops.Fir(taps, complex_data=cplx, device=device, max_block=0)


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            return next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        return ""


def cpu_baseline(name: str, seconds: float):
    """The CPU path beside the GPU number, kind 'port': the repo's restatement of the reference
    algorithm (oracle/qdsp_oracle.c), on this host's cores, three ways:
      value               all worker threads, VOLK-GENERIC accumulation order (sequential `acc += a*b`): every thread
                          filters its own chunk of the same synthetic stream with its own history, as the reference
                          would with one block graph per core;
      value_simd          the same with 16-lane partial sums + fused multiply-adds per output -- the shape of the
                          VOLK SIMD kernels a real build dispatches to (filter.h:63-67 is such a call);
      value_1thread_graph ONE block-graph worker thread: source -> filter block -> NullSink over dsp::stream hand-offs,
                          how the reference itself runs a block (block.h:55-57,83-85), generic order
                          (+ value_1thread_graph_simd).
    Bounded to roughly `seconds` of wall time in total."""
    import subprocess
    import tempfile

    import numpy as np

    import oracle as O

    w = WORKLOADS[name]
    taps = lowpass_taps(w["ntaps"] or 256, w.get("fc", 1.0 / 16.0))

    def make(acc):
        if name == "xlate":
            return O.Xlator(1.0, 0.1234)
        if w["rot"]:
            xl, rs = O.Xlator(1.0, 0.1234), O.Resampler(taps, 1, w["decim"], acc=acc)

            class _V:
                def process(self, x):
                    return rs.process(xl.process(x))
            return _V()
        if w["decim"] > 1:
            return O.Resampler(taps, 1, w["decim"], acc=acc)
        return O.Fir(taps, acc=acc)

    # A 1-GPU box gives this job a 16-core share of the host whatever the affinity mask says
    # (task statement: "size worker pools to the box's CPU share (16 for one GPU)").
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, int(os.environ.get("QDSP_BENCH_CPU_THREADS", "16"))))
    probe_n = 1 << 18
    x = O.synth_iq(0, probe_n, seed=1234)

    def threaded(acc, budget, cores=cores, share=1.0):
        """`cores` threads, each on its own run of blocks; `share` < 1: the threads are expected to share cores (more threads than the job's
        CPU share), the sample per thread shrinks accordingly so that the wall time stays near `budget`."""
        op = make(acc)
        op.process(x[:4096])
        t0 = time.perf_counter()
        op.process(x)
        r1 = probe_n / (time.perf_counter() - t0)            # samples/s, one thread
        per_thread = int(min(max(r1 * budget * 0.6 * share, probe_n), 1 << 27))
        blocks = max(1, per_thread // probe_n)
        per_thread = blocks * probe_n

        def worker(i, res):
            o = make(acc)
            xi = O.synth_iq(i * per_thread, probe_n, seed=1234)
            for _ in range(blocks):                          # blocks of 2^18, history carried
                o.process(xi)
            res[i] = True

        res = [False] * cores
        ths = [threading.Thread(target=worker, args=(i, res)) for i in range(cores)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        assert all(res)
        return cores * per_thread / dt / 1e6, r1 / 1e6, per_thread, blocks, dt

    v_gen, r1_gen, per_thread, blocks, dt_gen = threaded(O.ACC_F32, seconds * 0.35)
    out = {
        "cpu_model": _cpu_model(),
        "host_cores_visible": avail,
        "value": round(v_gen, 3),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "order": "scalar VOLK-generic accumulation order (sequential acc += a*b per component)",
        "value_1core": round(r1_gen, 3),
        "sample": f"{cores} threads x {per_thread} samples ({blocks} blocks of {probe_n}) of the same synthetic IQ, "
                  f"workload {name}, oracle/qdsp_oracle.c (VOLK-generic accumulation order, gcc -O3 target_clones), "
                  f"{dt_gen:.1f} s wall",
    }
    if avail > cores:
        # "all nproc cores".  The affinity mask of a 1-GPU box shows every core of the host (256) while the job's CPU share is
        # 16, so this figure says what the mask's cores deliver to THIS job, bounded to the same wall budget -- beside `value`, not instead.
        # (the sample per thread is sized for `cores` cores' worth of time slices: round 4's first version sized it as if every thread had a
        # core of its own and spent 46 s of wall time on a 2.8 s budget)
        v_all, _, pt_all, bl_all, dt_all = threaded(O.ACC_F32, seconds * 0.2, avail, share=cores / avail)
        out["value_all_cores"] = round(v_all, 3)
        out["cores_all"] = avail
        out["sample_all_cores"] = f"{avail} threads x {pt_all} samples ({bl_all} blocks of {probe_n}), generic order, {dt_all:.1f} s wall"
    if name != "xlate":
        v_simd, r1_simd, pt2, bl2, dt_simd = threaded(O.ACC_SIMD, seconds * 0.2)
        out["value_simd"] = round(v_simd, 3)
        out["value_simd_1core"] = round(r1_simd, 3)
        out["sample_simd"] = (f"{cores} threads x {pt2} samples, one output at a time with 64 float lanes of partial sums + FMA "
                              f"(the shape of VOLK's AVX-512 dot-product kernels), {dt_simd:.1f} s wall")
    # one worker thread per block through the block graph (source -> filter -> NullSink)
    exe = os.path.join(ROOT, "oracle", "cpu_graph_bench")
    if name != "xlate" and not w["rot"] and os.path.exists(exe):
        with tempfile.NamedTemporaryFile(suffix=".f32") as tf:
            np.asarray(taps, np.float32).tofile(tf.name)
            kind = "fir" if w["decim"] == 1 else "decim"
            for acc, key in ((0, "value_1thread_graph"), (3, "value_1thread_graph_simd")):
                try:
                    r = subprocess.run([exe, kind, tf.name, str(w["decim"]), str(acc), "500000", f"{max(1.0, seconds * 0.12):.2f}"],
                                       capture_output=True, text=True, timeout=60, check=True)
                    out[key] = round(float(r.stdout.strip().split("msps=")[1]), 3)
                except Exception as e:  # noqa: BLE001  (a baseline that cannot be produced is reported as such)
                    out[key] = None
                    out[key + "_error"] = str(e)[:200]
        out["sample_1thread_graph"] = ("oracle/cpu_graph_bench: HandlerSource -> CPU filter block (oracle arithmetic) -> NullSink on the "
                                       "block-runtime mirror, one std::thread per block, 500000-sample blocks")
    return out


WORKLOAD_TEXT = {
    "fir256": "256-tap complex FIR (real taps), synthetic IQ resident in HBM (BASELINE configs[1])",
    "fir63": "63-tap complex FIR, synthetic IQ (BASELINE configs[0] shape)",
    "decim8": "256-tap polyphase decimate-by-8, synthetic IQ",
    "xlate_fir_decim8": "fused NCO + 256-tap FIR + decimate-by-8 (BASELINE configs[2])",
    "xlate": "NCO frequency translator alone",
    "decim4": "256-tap polyphase decimate-by-4, synthetic IQ",
    "xlate_fir_decim4": "fused NCO + 256-tap FIR + decimate-by-4",
    "decim8_t63": "63-tap polyphase decimate-by-8",
    "xlate_fir_decim8_t63": "fused NCO + 63-tap FIR + decimate-by-8",
    "vfo50": "fused NCO + 401-tap FIR + decimate-by-50 (the reference VFO's 2.4 Msps -> 48 kHz shape)",
    "resamp147_160": "rational resampler 147/160 (48 kHz -> 44.1 kHz), 16 taps per phase",
    "chan64": "64-channel polyphase channelizer, 256 taps, decimate 64 (BASELINE configs[4])",
    "chan64m8": "64-channel polyphase channelizer oversampled by 8: 256 taps, decimate 8 (BASELINE configs[4], M = 8 variant)",
    "fir256_real": "256-tap FIR<float> on real samples",
    "decim8_real": "256-tap polyphase decimate-by-8 on real samples",
    "decim4_real": "256-tap polyphase decimate-by-4 on real samples",
    "decim50_real": "401-tap polyphase decimate-by-50 on real samples",
    "resamp147_160_real": "rational resampler 147/160 on real samples, 16 taps per phase",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="fir256", choices=sorted(WORKLOADS))
    ap.add_argument("--log2n", type=int, default=27, help="input samples per GPU per step = 2^log2n")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-block-call", action="store_true",
                    help="skip the `block_call` measurement (profiling runs: its 600 small launches would share kernel names with the "
                         "timed launches in rocprofv3's per-kernel statistics)")
    ap.add_argument("--no-chain", action="store_true",
                    help="default run (fir256) only: skip the second driver-timed leg, the fused xlating-FIR + decimate-by-8 chain "
                         "(BASELINE configs[2]) reported as the `chain` object of the same JSON line")
    ap.add_argument("--no-channelizer", action="store_true",
                    help="default run (fir256) only: skip the `channelizer` / `channelizer_m8` legs (BASELINE configs[4]: 64 channels, "
                         "decimation 64 and 8)")
    ap.add_argument("--kernel-iters", type=int, default=10)
    ap.add_argument("--spinup-ms", type=float, default=200.0,
                    help="untimed device spin-up before the W warmup steps: the GPU needs ~50 launches (~25 ms) "
                         "for its clocks to settle under this load; the timed region is still exactly K steps")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), as a CHILD process and BEFORE anything in this
    process has touched the GPU (no torch import yet) -- a GPU-initialised process must never exec.  Returns the
    children's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def run_workload(name: str, args, ctx) -> dict:
    """W warmup + exactly K timed steps of one workload on this rank's GPU (all ranks together when world > 1),
    then the dominant kernel timed with HIP events.  Returns the measured fields."""
    import torch
    import torch.distributed as dist

    from qdsp_amd import ops
    from qdsp_amd.sharding import RingStream, chunk_alignment

    world, rank, local_rank, dev = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["dev"]
    rehearse, self_ring = ctx["rehearse"], ctx["self_ring"]
    ctrl, ctrl_dev = ctx.get("ctrl"), ctx.get("ctrl_dev", "cpu")
    ring = None

    def ctrl_reduce(v: float, how=None) -> float:
        """Control-plane reduction over the ranks.  The ring's own RCCL communicator must be idle first: two communicators of
        one process with operations in flight on different streams may wait for each other (so: drain, then the collective --
        which runs on gloo whenever that group could be had, main())."""
        if world == 1:
            return v
        if ring is not None:
            ring.drain()
        t = torch.tensor([v], dtype=torch.float64, device=ctrl_dev)
        dist.all_reduce(t, op=how or dist.ReduceOp.MAX, group=ctrl)
        return float(t.item())

    def ctrl_barrier():
        ctrl_reduce(0.0)

    def dbg(msg):
        if os.environ.get("QDSP_BENCH_DEBUG"):
            print(f"[bench rank {rank}] {name}: {msg}", file=sys.stderr, flush=True)

    w = WORKLOADS[name]
    n = 1 << args.log2n
    # chunk starts must be multiples of lcm(decim, 512) once the stream is cut over ranks (per-call phase restart of the
    # resampler, VOLK gain cadence: qdsp_amd/sharding.py); a single rank runs one continuous stream and needs none
    align = chunk_alignment(w["decim"], w.get("interp", 1), 512 if w["rot"] else 0) if (world > 1 or self_ring) else 1
    n -= n % align
    if w.get("interp", 1) > 1:
        n -= n % w["decim"]
    op = make_op(ops, name, local_rank)
    is_chan = name in ("chan64", "chan64m8")
    has_hist = name != "xlate"
    has_nco = w["rot"]
    H = op.history_len if has_hist else 0

    # Block-cyclic cut of one continuous stream (qdsp_amd/sharding.py RingStream): step s, rank r owns samples
    # [(s*world + r)*n, +n).  The synthetic block content repeats every step (same buffer), the halo hand-off is
    # the real one: every step each rank's tail goes to its ring successor over RCCL, prefetched under the kernel.
    x = ops.synth_iq(n, first_sample=rank * n, seed=1234, device=local_rank)
    is_real = w.get("real", False)
    if is_real:
        if world > 1 or self_ring:
            raise SystemExit(f"workload {name}: the real-data workloads run on one GPU (the ring halo carries complex samples)")
        x = torch.view_as_real(x)[:, 0].contiguous()          # the real parts: uniform [-1, 1)
    nout = n // w["decim"] * w.get("interp", 1)
    # channel rows 32 samples (256 bytes) longer than the data: with a power-of-two row stride (2^24 samples at M = 8) the 64 lines a
    # tile writes -- one per channel -- fall on the same memory channel (chan64m8: 2.85 ms, 2.46 with the pad; EXPERIMENTS.md section 4).  The
    # stride is the caller's to choose (qdsp_hip_chan_cf32_process_dev's out_stride argument); reported in config.out_row_stride.
    out = torch.empty((w["nchan"], nout + CHAN_ROW_PAD) if is_chan else nout, dtype=torch.float32 if is_real else torch.complex64, device=dev)
    ring = RingStream(op, n, rank, world, transport="host" if rehearse else "device", align=align,
                      exchange=(world > 1 or self_ring), ctrl_group=ctrl,
                      prefetch=os.environ.get("QDSP_BENCH_NO_PREFETCH", "0") != "1")
    c_ring = getattr(ring, "_ring", None)
    if c_ring is not None:
        c_ring.set_timing(True)

    def step():
        ring.step(x, out, next_x=x)

    def restart():
        """Operator and ring back to the start of the stream (after the communication-free spin-up)."""
        if has_hist:
            op.reset()                     # zero history, NCO phase 0
        else:
            op.set_phase(1.0, 0.0)
        if has_nco and world > 1:
            op.advance(rank * n)

    if args.spinup_ms > 0:
        # Time-based, so it must be communication-free (ranks run different counts): the bare kernel only.
        t_end = time.perf_counter() + args.spinup_ms * 1e-3
        while time.perf_counter() < t_end:
            op.process(x, out)
            torch.cuda.synchronize()
        restart()
        torch.cuda.synchronize()
        ctrl_barrier()
    dbg("spinup done")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # Halo check (outside the timed region, no oracle): a fresh operator fed the predecessor's
    # regenerated tail + my head must reproduce my head.
    halo_err = None
    if world > 1 and H:
        pos_now = ring.stream_position()
        step()
        torch.cuda.synchronize()
        m = 1 << 16
        prev_tail = ops.synth_iq(H, first_sample=((rank - 1) % world) * n + n - H, seed=1234, device=local_rank)
        chk = make_op(ops, name, local_rank)
        if has_nco:
            chk.advance(pos_now)
        chk.set_history_dev(prev_tail)
        ref = chk.process(x[:m])
        torch.cuda.synchronize()
        got = out[:, : ref.shape[1]] if is_chan else out[: ref.numel()]
        # (not bit-equal by construction: the short reference call ends in a zero-padded FFT
        # segment where the full chunk has real samples; a wrong halo is an O(1) error)
        err = (ref - got).abs().max().item()
        halo_err = err / max(ref.abs().max().item(), 1e-30)
        if not halo_err < 2e-6:
            # (the other ranks are released by the launcher: torch.distributed.run ends every rank when one exits non-zero)
            raise SystemExit(f"rank {rank}: {name}: halo exchange produced different outputs than the unsharded filter (max err {err:.3e})")
        chk.close()
        halo_err = ctrl_reduce(halo_err)
    dbg("halo check done")

    if c_ring is not None:
        ring.drain()
        c_ring.set_timing(True)        # (restarts the statistics: only the timed region's exchanges are reported)
    ctrl_barrier()
    torch.cuda.synchronize()
    # HIP events on the stream the operator launches on (torch's current stream: ops.* pass it to *_process_dev), around the SAME
    # K steps the wall clock brackets: the dominant kernel's mean launch duration, <= ms_per_step by construction
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ctrl_barrier()                     # (drains the exchange prefetched for a step that does not run, then the barrier)
    dt = time.perf_counter() - t0
    kms_timed = ev0.elapsed_time(ev1) / args.steps
    dt = ctrl_reduce(dt)
    dbg("timed loop done")
    ring.drain()
    rccl = None
    if world > 1 or self_ring:
        info = c_ring.info() if c_ring is not None else {}
        xus = c_ring.exchange_us() if c_ring is not None else {}
        rccl = {
            "ring_transport": ring.transport_name(),
            "ring_ranks_requested": world,
            # what ncclCommCount says on rank 0, and the smallest / largest answer over all ranks
            "ring_comm_ranks": info.get("comm_ranks"),
            # (no C ring -- rehearsal over gloo, or the torch p2p fallback -- : None; the reductions still run so that every rank issues the same collectives)
            "ring_comm_ranks_min": (lambda v: int(v) if c_ring is not None else None)(-ctrl_reduce(-float(info.get("comm_ranks", -1)))) if world > 1 else info.get("comm_ranks"),
            "ring_comm_ranks_max": (lambda v: int(v) if c_ring is not None else None)(ctrl_reduce(float(info.get("comm_ranks", -1)))) if world > 1 else info.get("comm_ranks"),
            "ring_comm_device": info.get("comm_device"),
            "rccl_version": info.get("rccl_version"),
            "halo_bytes": H * 8,
            "halo_check_max_rel_err": halo_err,
            "halo_check": ("fresh operator + regenerated predecessor tail vs this rank's output head, max over ranks" if halo_err is not None
                           else "not run (one rank: its own tail is its halo; tests/test_gpu_ring.py compares that stream with the oracle)"),
            "exchange_us_mean": round(ctrl_reduce(xus.get("mean_us", 0.0)), 2),
            "exchange_us_max": round(ctrl_reduce(xus.get("max_us", 0.0)), 2),
            "exchanges_timed": xus.get("exchanges"),
            "exchange_us_source": "HIP events around each ncclSend/ncclRecv group on the ring stream, timed region only, max over ranks "
                                  "(includes waiting for the neighbour's end of the pair)" if c_ring is not None else None,
            "control_plane": ctx.get("ctrl_name"),
        }
    ms_per_step = dt / args.steps * 1e3
    value = world * n / (dt / args.steps) / 1e6

    # The same kernel once more on its own (args.kernel_iters back-to-back launches, no ring bookkeeping between them): reported as
    # kernel_ms_isolated; `roofline` is computed from the launches of the timed region.
    kms_iso = op.time_dev(x, out, args.kernel_iters)
    kinfo = op.last_kernel()
    kms = kms_timed
    torch.cuda.synchronize()
    dbg("kernel timing done")
    achieved_gbs = w["bytes"] * n / (kms * 1e-3) / 1e9
    # FLOPs actually executed per input sample: direct form = 4*ntaps/decim (+6 for the NCO);
    # the overlap-save kernel: 6 radix-16 passes + 4 twiddle passes + spectrum product per
    # 4096-point block = 1524 FLOP/lane x 256 lanes / (4097 - ntaps) valid outputs.
    flops = w["flops"]
    if kinfo["name"] in ("fir_fft_kernel", "fir_fft_dma_kernel", "fir_fft_dmapk_kernel"):
        flops = 1524.0 * 256 / (4097 - w["ntaps"])
    elif kinfo["name"] == "pfb_dec8_kernel":
        # polyphase overlap-save, one wave per segment of 4096 inputs: ~2860 VALU instructions per wave (PMC), of which
        # ~950 are fused multiply-adds: ~3800 FLOP per lane x 64 lanes / (8 * (513 - ntaps/8)) new input samples
        flops = 3800.0 * 64 / (8 * (513 - (w["ntaps"] + 7) // 8))
    elif kinfo["name"] == "pfb_dec4_kernel":
        # the same eight column transforms against two spectrum sets + two inverses: ~3850 VALU instructions per wave and segment
        flops = 5100.0 * 64 / (8 * (513 - (w["ntaps"] + 4 + 7) // 8))
    elif kinfo["name"] == "chan_uniform_kernel":
        # per lane and wave tile (16 output times x 64 channels = 1024 input samples per wave): 64 complex
        # MACs (512), radix-16 (~200), twiddles (~90), radix-4 across the quad (~220), per-channel
        # rotation (~320): ~1350 FLOP x 64 lanes / 1024 input samples
        flops = 84.0 * (64 // w["decim"])
    elif kinfo["name"] == "resamp_mfma_kernel":
        # 16 blocks x 4 x 4 MACs x 2 components per step; per 4 periods (4 * decim inputs): groups * band columns steps
        L, M = w.get("interp", 1), w["decim"]
        flops = 2.0 * 512 * ((L + 63) // 64) * (3 * M // L + (w["ntaps"] + L - 1) // L + 1) / (4.0 * M)
    elif kinfo["name"] == "decim_mfma_kernel":
        # FP32 MFMA: per tile of 16 rows (16 * decim samples) 4 * ceil(decim / 8) v_mfma_f32_16x16x4_f32 of 2048 FLOP (the
        # A operand carries 16 tap rows whatever the tap count), one tile in 17 read twice; + ~14 VALU FLOP per sample (NCO)
        flops = 2048.0 * 4 * ((w["decim"] + 7) // 8) / (16 * w["decim"]) * 17 / 16 + (14.0 if w["rot"] else 0.0)
    achieved_tf = flops * n / (kms * 1e-3) / 1e12
    res = {
        "value": round(value, 1),
        "ms_per_step": round(ms_per_step, 4),
        "config": {
            "workload": WORKLOAD_TEXT[name],
            "name": name,
            "samples_per_gpu_per_step": n,
            "ntaps": w["ntaps"],
            "decim": w["decim"],
            "halo_samples": H if world > 1 else 0,
            "partition": ("single stream" + (" (self-ring RCCL exchange every step)" if self_ring else "")) if world == 1
                         else f"block-cyclic time chunks over {world} ranks, ring halo over RCCL "
                              + ("(qdsp_hip_ring_*, the C ABI)" if getattr(ring, "_ring", None) is not None else
                                 "(gloo, ranks sharing one GPU: rehearsal)" if rehearse else "(torch.distributed p2p)"),
            **({"channels": w["nchan"], "out_row_stride": nout + CHAN_ROW_PAD} if is_chan else {}),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": kinfo["name"],
            "kernel_ms": round(kms, 4),
            "kernel_ms_source": "HIP events around the K timed steps (one launch per step per GPU)",
            "kernel_ms_isolated": round(kms_iso, 4),
            "algorithmic_bytes_per_sample": w["bytes"],
            "launch": {"grid": kinfo["grid"], "block": kinfo["block"], "lds_bytes": kinfo["lds_bytes"]},
        },
        "roofline_fp32": {
            "bound": "mfma" if kinfo["name"] in ("decim_mfma_kernel", "resamp_mfma_kernel") else "valu",
            "achieved": round(achieved_tf, 2),
            "peak": FP32_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / FP32_PEAK_TFLOPS, 4),
            "flops_per_sample": round(flops, 1),
            "direct_form_flops_per_sample": w["flops"],
        },
        "hbm_roofline_msps": round(HBM_PEAK_GBS * 1e9 / w["bytes"] / 1e6, 1),
        "frac_of_hbm_roofline_msps": round(value / world / (HBM_PEAK_GBS * 1e9 / w["bytes"] / 1e6), 4),
    }
    if rccl is not None:
        res["rccl"] = rccl
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        try:
            t = json.load(open(tj)).get(name)
            if t and t.get("samples") == n:
                res["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                res["roofline"]["traffic_measured_in_run"] = False     # PMC passes cannot share a run with the timing (MI355X_MICROARCH.md)
                res["roofline"]["traffic_source"] = t.get("source")
        except Exception:
            pass
    op.close()
    del x, out
    torch.cuda.empty_cache()
    return res


def block_call(name: str, ctx) -> dict:
    """One reference-sized call: the block API the operators sit behind hands over at most 1e6 samples per call
    (src/dsp/stream.h:7).  200 back-to-back calls on device buffers, HIP events on the operator's stream."""
    import torch

    from qdsp_amd import ops

    w = WORKLOADS[name]
    n = 1_000_000
    n -= n % (w["decim"] if w.get("interp", 1) > 1 else 1)
    op = make_op(ops, name, ctx["local_rank"])
    x = ops.synth_iq(n, seed=4321, device=ctx["local_rank"])
    if w.get("real", False):
        x = torch.view_as_real(x)[:, 0].contiguous()
    nout = n // w["decim"] * w.get("interp", 1)
    out = torch.empty((w["nchan"], nout + CHAN_ROW_PAD) if name in ("chan64", "chan64m8") else nout + 8,
                      dtype=torch.float32 if w.get("real", False) else torch.complex64, device=ctx["dev"])
    op.process(x, out)
    torch.cuda.synchronize()
    ms = min(op.time_dev(x, out, 200) for _ in range(3))
    kernel = op.last_kernel()["name"]
    op.close()
    return {"samples": n, "us_per_call": round(ms * 1e3, 2), "msps": round(n / ms / 1e3, 1), "kernel": kernel}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    # RCCL between processes needs dmabuf IPC on this platform (the host driver supports nothing else): without this setting the
    # first send / recv between two ranks fails with `hipIpcGetMemHandle: invalid argument`.  It is exported by the image already;
    # set here as well -- before anything initialises HIP -- so that a launcher with a scrubbed environment still works.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: never depend on the hostname resolving

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU fallback for the hot path")
    # Rehearsal on a 1-GPU box: QDSP_BENCH_REHEARSE=1 puts every rank on cuda:0 and moves the
    # halo through gloo/CPU tensors (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearse = os.environ.get("QDSP_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # QDSP_BENCH_SELF_RING=1 (1 GPU only, never set by the driver): run the RCCL ring exchange with this rank
    # as its own neighbour, to exercise the real send/recv path and see its per-step cost on a 1-GPU box.
    self_ring = world == 1 and os.environ.get("QDSP_BENCH_SELF_RING", "0") == "1"
    if world > 1 or self_ring:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctx = dict(world=world, rank=rank, local_rank=local_rank, dev=dev, rehearse=rehearse, self_ring=self_ring)
    if world > 1:
        # Control plane (barriers, the max-over-ranks clock, the ring's id broadcast) on gloo: the only RCCL communicator with
        # traffic in this process is then the halo ring's.  If gloo cannot be had (every rank fails alike: one host), the default
        # RCCL group carries it, always behind a drained ring (run_workload.ctrl_reduce).
        if rehearse:
            ctx.update(ctrl=None, ctrl_dev="cpu", ctrl_name="gloo (default group, rehearsal)")
        else:
            try:
                ctx.update(ctrl=dist.new_group(backend="gloo"), ctrl_dev="cpu", ctrl_name="gloo group beside the default nccl group")
            except Exception as e:  # noqa: BLE001
                print(f"bench.py rank {rank}: no gloo control group ({e!r}); control collectives on the default RCCL group", file=sys.stderr, flush=True)
                ctx.update(ctrl=None, ctrl_dev=dev, ctrl_name="default nccl group (no gloo)")

    res = run_workload(args.workload, args, ctx)
    chain = None
    if args.workload == "fir256" and not args.no_chain:
        # the metric's name says "FIR+decimate chain": the fused xlating FIR + decimate-by-8 (BASELINE configs[2]) is
        # timed by the same protocol (W warmup, K steps, barrier + synchronize on both sides, max over ranks) right
        # behind the FIR and reported in the same line
        chain = run_workload("xlate_fir_decim8", args, ctx)

    chan = chan8 = None
    if args.workload == "fir256" and not args.no_chain and not args.no_channelizer:
        # BASELINE configs[4]: the 64-channel channelizer on one stream, critically sampled (decimation 64) and its M = 8 variant
        # (SURVEY 8d config 5), by the same protocol
        chan = run_workload("chan64", args, ctx)
        chan8 = run_workload("chan64m8", args, ctx)

    if rank == 0:
        line = {
            "metric": "Msamples/s complex IQ through 256-tap FIR+decimate chain, 1/2/4/8 GPU",
            "value": res["value"],
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "spinup_ms": args.spinup_ms,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": res["config"],
            "roofline": res["roofline"],
            "roofline_fp32": res["roofline_fp32"],
            "hbm_roofline_msps": res["hbm_roofline_msps"],
            "frac_of_hbm_roofline_msps": res["frac_of_hbm_roofline_msps"],
        }
        if "rccl" in res:
            line["rccl"] = res["rccl"]
        if chain is not None:
            line["chain"] = {"value": chain["value"], "unit": "Msamples/s", "ms_per_step": chain["ms_per_step"],
                             "steps": args.steps, "warmup": args.warmup, "config": chain["config"],
                             "roofline": chain["roofline"], "hbm_roofline_msps": chain["hbm_roofline_msps"],
                             "frac_of_hbm_roofline_msps": chain["frac_of_hbm_roofline_msps"],
                             **({"rccl": chain["rccl"]} if "rccl" in chain else {})}
        for key, leg in (("channelizer", chan), ("channelizer_m8", chan8)):
            if leg is not None:
                line[key] = {"value": leg["value"], "unit": "Msamples/s (input rate, all 64 channels)", "ms_per_step": leg["ms_per_step"],
                             "steps": args.steps, "warmup": args.warmup, "config": leg["config"], "roofline": leg["roofline"],
                             "hbm_roofline_msps": leg["hbm_roofline_msps"], "frac_of_hbm_roofline_msps": leg["frac_of_hbm_roofline_msps"],
                             **({"rccl": leg["rccl"]} if "rccl" in leg else {})}
        if world == 1 and not args.no_block_call:
            # what a block of the reference's graph gets per call (latency-bound: DESIGN.md section 5, EXPERIMENTS.md "Reference-sized calls")
            line["block_call"] = block_call(args.workload, ctx)
            if chain is not None:
                line["chain"]["block_call"] = block_call("xlate_fir_decim8", ctx)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1 or self_ring:
        torch.cuda.synchronize()
        if world > 1:
            t = torch.zeros(1, device=ctx["ctrl_dev"])
            dist.all_reduce(t, group=ctx["ctrl"])          # (every ring is drained and closed by now)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
