#!/usr/bin/env python3
"""bench.py -- Msamples/s of complex IQ through the FIR / decimator hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload fir256]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
(one rank per GPU, backend nccl == RCCL).

A "step" = one pass of the hot path over one batch: every rank runs its operator once over
its 2^LOG2N-sample chunk of a continuous synthetic IQ stream that is already resident in
HBM.  With N > 1 the stream is cut block-cyclically over the ranks and each step begins with
the ring-neighbour halo hand-off (the previous chunk's last H input samples -> this rank's
filter history, RCCL send/recv over xGMI, qdsp_amd/sharding.py); there is no other
communication.  Per-GPU work is fixed, so scaling is "weak".

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
  "cpu_baseline": the CPU oracle (VOLK-generic order) timed on this host's cores, N = 1 only.
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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
FP32_PEAK_TFLOPS = 157.3   # FP32 vector == FP32 MFMA peak (same file)

# algorithmic bytes / flops per INPUT sample
WORKLOADS = {
    "fir256": dict(ntaps=256, decim=1, rot=False, bytes=16.0, flops=1024.0),
    "fir63": dict(ntaps=63, decim=1, rot=False, bytes=16.0, flops=252.0),
    "decim8": dict(ntaps=256, decim=8, rot=False, bytes=9.0, flops=128.0),
    "xlate_fir_decim8": dict(ntaps=256, decim=8, rot=True, bytes=9.0, flops=134.0),
    "xlate": dict(ntaps=0, decim=1, rot=True, bytes=16.0, flops=6.0),
    # short filters (the strided-window direct kernel)
    "decim8_t63": dict(ntaps=63, decim=8, rot=False, bytes=9.0, flops=31.5),
    "xlate_fir_decim8_t63": dict(ntaps=63, decim=8, rot=True, bytes=9.0, flops=37.5),
    # the reference VFO's usual shape (vfo.h: 2.4 Msps -> 48 kHz, ~8 taps per unit of decimation): 8 + 8/50 B
    "vfo50": dict(ntaps=401, decim=50, rot=True, bytes=8.16, flops=38.1, fc=0.4 / 50),
    # BASELINE configs[4]: 64 channels at (c - 31.5) fs/64, 256 taps, decimate 64: 8 B in + 64*8/64 B out
    "chan64": dict(ntaps=256, decim=64, rot=True, bytes=16.0, flops=1408.0, nchan=64),
    # its oversampled variant (SURVEY 8d config 5 "and M = 8"): 8 B in + 64*8/8 B out per input sample
    "chan64m8": dict(ntaps=256, decim=8, rot=True, bytes=72.0, flops=16384.0, nchan=64),
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
    if w["rot"]:
        return ops.Vfo(taps, 1, w["decim"], ops.phase_delta(1.0, 0.1234), device=device, max_block=0)
    if w["decim"] > 1:
        return ops.Resampler(taps, 1, w["decim"], device=device, max_block=0)
    return ops.Fir(taps, device=device, max_block=0)


def cpu_baseline(name: str, seconds: float):
    """The CPU oracle (kind 'port': the repo's restatement of the reference algorithm with
    VOLK-generic accumulation order) on this host's cores: every thread filters its own
    chunk of the same synthetic stream (own history), as the reference would with one block
    graph per core.  Bounded to roughly `seconds` of wall time."""
    import numpy as np

    import oracle as O

    w = WORKLOADS[name]
    taps = lowpass_taps(w["ntaps"] or 256, w.get("fc", 1.0 / 16.0))

    def make():
        if name == "xlate":
            return O.Xlator(1.0, 0.1234)
        if w["rot"]:
            xl, rs = O.Xlator(1.0, 0.1234), O.Resampler(taps, 1, w["decim"])

            class _V:
                def process(self, x):
                    return rs.process(xl.process(x))
            return _V()
        if w["decim"] > 1:
            return O.Resampler(taps, 1, w["decim"])
        return O.Fir(taps)

    # A 1-GPU box gives this job a 16-core share of the host whatever the affinity mask says
    # (task statement: "size worker pools to the box's CPU share (16 for one GPU)").
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, int(os.environ.get("QDSP_BENCH_CPU_THREADS", "16"))))
    probe_n = 1 << 18
    x = O.synth_iq(0, probe_n, seed=1234)
    op = make()
    op.process(x[:4096])
    t0 = time.perf_counter()
    op.process(x)
    r1 = probe_n / (time.perf_counter() - t0)            # samples/s, one thread
    per_thread = int(min(max(r1 * seconds * 0.6, probe_n), 1 << 27))
    blocks = max(1, per_thread // probe_n)
    per_thread = blocks * probe_n

    def worker(i, res):
        o = make()
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
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {
        "cpu_model": model,
        "host_cores_visible": avail,
        "value": round(cores * per_thread / dt / 1e6, 3),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "value_1core": round(r1 / 1e6, 3),
        "sample": f"{cores} threads x {per_thread} samples ({blocks} blocks of {probe_n}) of the same synthetic IQ, "
                  f"workload {name}, oracle/qdsp_oracle.c (VOLK-generic accumulation order, gcc -O3 target_clones), "
                  f"{dt:.1f} s wall",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="fir256", choices=sorted(WORKLOADS))
    ap.add_argument("--log2n", type=int, default=27, help="input samples per GPU per step = 2^log2n")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=10)
    ap.add_argument("--spinup-ms", type=float, default=200.0,
                    help="untimed device spin-up before the W warmup steps: the GPU needs ~50 launches (~25 ms) "
                         "for its clocks to settle under this load; the timed region is still exactly K steps")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from qdsp_amd import capi as capi_mod
    from qdsp_amd import ops, sharding  # noqa: F401  (sharding documents the partition this file uses)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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
    ring = world > 1 or self_ring

    def dbg(msg):
        if os.environ.get("QDSP_BENCH_DEBUG"):
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    w = WORKLOADS[args.workload]
    n = 1 << args.log2n
    op = make_op(ops, args.workload, local_rank)
    is_chan = args.workload in ("chan64", "chan64m8")
    has_hist = args.workload != "xlate"
    has_nco = w["rot"]
    H = op.history_len if has_hist else 0

    # Block-cyclic cut of one continuous stream: step s, rank r owns samples
    # [(s*world + r)*n, +n).  The synthetic block content repeats every step (same buffer),
    # so the halo a rank needs is always its ring predecessor's current tail.
    x = ops.synth_iq(n, first_sample=rank * n, seed=1234, device=local_rank)
    nout = n // w["decim"]
    out = torch.empty((w["nchan"], nout) if is_chan else nout, dtype=torch.complex64, device=dev)
    tail = x[n - H:] if H else None
    halo = torch.zeros(max(H, 1), dtype=torch.complex64, device=dev)   # RCCL recv lands here ...
    set_hist = getattr(capi_mod.load(), op._prefix + "_set_history_dev") if H else None
    tail_f = torch.view_as_real(tail).contiguous() if H else None   # (H, 2) float32, its own buffer
    halo_f = torch.view_as_real(halo)                               # view: writes land in `halo`

    # the timed loop talks to the C ABI directly (no per-step Python checks / tensor slicing)
    import ctypes as C

    capi = capi_mod

    h, xin, yout = op._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if is_chan:
        _chan_fn = capi.load().qdsp_hip_chan_cf32_process_dev

        def fn(h_, xin_, n_, yout_, stream_):
            return _chan_fn(h_, xin_, n_, yout_, nout, stream_)
    else:
        fn = getattr(capi.load(), op._prefix + "_process_dev")

    # The ring exchange of every step, built once (float32 views: a plain dtype for RCCL).  The halo of step
    # s+1 is the predecessor's INPUT tail, known before step s is computed: it is requested right after
    # step s's history is installed and BEFORE step s's kernel is launched, into the other of two halo
    # buffers, so the RCCL send/recv (its own stream, ordered after everything already queued here) runs
    # under the kernel instead of in front of it (measured on one GPU as its own ring neighbour:
    # +0.037 ms per step when serialised).  QDSP_BENCH_NO_PREFETCH=1 restores the serial order.
    p2p_ops = None
    halo2 = [halo, torch.zeros_like(halo)]
    prefetch = ring and H and not rehearse and os.environ.get("QDSP_BENCH_NO_PREFETCH", "0") != "1"
    if ring and H and not rehearse:
        p2p_ops = [[dist.P2POp(dist.isend, tail_f, (rank + 1) % world), dist.P2POp(dist.irecv, torch.view_as_real(hb), (rank - 1) % world)]
                   for hb in halo2]
    pending = {"reqs": None, "par": 0}

    def step():
        if prefetch:
            if pending["reqs"] is None:                      # first step (or after drain()): nothing in flight yet
                pending["reqs"] = dist.batch_isend_irecv(p2p_ops[pending["par"]])
            for r in pending["reqs"]:
                r.wait()
            rc = set_hist(h, C.c_void_p(halo2[pending["par"]].data_ptr()), stream)
            if rc < 0:
                capi.check(int(rc), "set_history_dev")
            pending["par"] ^= 1
            pending["reqs"] = dist.batch_isend_irecv(p2p_ops[pending["par"]])   # the next step's halo
        elif ring and H:
            # ring halo: my tail -> next rank's history; previous rank's tail -> mine
            if rehearse:
                tail_h, halo_h = tail.cpu(), torch.empty(H, dtype=torch.complex64)
                reqs = dist.batch_isend_irecv([
                    dist.P2POp(dist.isend, torch.view_as_real(tail_h), (rank + 1) % world),
                    dist.P2POp(dist.irecv, torch.view_as_real(halo_h), (rank - 1) % world),
                ])
                for r in reqs:
                    r.wait()
                halo.copy_(halo_h)
            else:
                reqs = dist.batch_isend_irecv(p2p_ops[0])
                for r in reqs:
                    r.wait()
            # ... and is copied (2 KB, device to device, same stream) into the filter's history
            rc = set_hist(h, C.c_void_p(halo.data_ptr()), stream)
            if rc < 0:
                capi.check(int(rc), "set_history_dev")
        rc = fn(h, xin, n, yout, stream)
        if rc < 0:
            capi.check(int(rc), "process_dev")

    def drain():
        """Complete the exchange requested for a step that will not run (every rank has one in flight)."""
        if pending["reqs"] is not None:
            for r in pending["reqs"]:
                r.wait()
            pending["reqs"] = None

    # NCO bookkeeping without communication: `pos` = stream position of this rank's next
    # chunk; phases are exact multiples of the fixed-point increment, so advance() is exact.
    pos = rank * n
    if has_nco and world > 1:
        op.advance(pos)

    def stepped():
        nonlocal pos
        pos += world * n
        if has_nco and world > 1:
            op.advance((world - 1) * n)   # the call itself advanced by n

    if args.spinup_ms > 0:
        # Time-based, so it must be communication-free (ranks run different counts): the bare
        # kernel only, then the operator state is put back to the start of the stream.
        t_end = time.perf_counter() + args.spinup_ms * 1e-3
        while time.perf_counter() < t_end:
            rc = fn(h, xin, n, yout, stream)
            if rc < 0:
                capi.check(int(rc), "process_dev")
            torch.cuda.synchronize()
        if has_hist:
            op.reset()                     # zero history, NCO phase 0
        else:
            op.set_phase(1.0, 0.0)
        if has_nco and world > 1:
            op.advance(pos)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    dbg("spinup done")
    for _ in range(args.warmup):
        step()
        stepped()
    torch.cuda.synchronize()

    # Halo check (outside the timed region, no oracle): a fresh operator fed the predecessor's
    # regenerated tail + my head must reproduce my head.
    if world > 1 and H:
        pos_now = pos
        step()
        stepped()
        torch.cuda.synchronize()
        m = 1 << 16
        prev_tail = ops.synth_iq(H, first_sample=((rank - 1) % world) * n + n - H, seed=1234, device=local_rank)
        chk = make_op(ops, args.workload, local_rank)
        if has_nco:
            chk.advance(pos_now)
        chk.set_history_dev(prev_tail)
        ref = chk.process(x[:m])
        torch.cuda.synchronize()
        got = out[:, : ref.shape[1]] if is_chan else out[: ref.numel()]
        # (not bit-equal by construction: the short reference call ends in a zero-padded FFT
        # segment where the full chunk has real samples; a wrong halo is an O(1) error)
        err = (ref - got).abs().max().item()
        if not err < 2e-6 * max(ref.abs().max().item(), 1e-30):
            raise SystemExit(f"rank {rank}: halo exchange produced different outputs than the unsharded filter (max err {err:.3e})")
        chk.close()

    dbg("halo check done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        stepped()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    dbg("timed loop done")
    ms_per_step = dt / args.steps * 1e3
    value = world * n / (dt / args.steps) / 1e6

    # Dominant kernel, timed with HIP events on the stream it is launched on.
    kms = op.time_dev(x, out, args.kernel_iters)
    dbg("kernel timing done")
    kinfo = op.last_kernel()
    torch.cuda.synchronize()
    achieved_gbs = w["bytes"] * n / (kms * 1e-3) / 1e9
    # FLOPs actually executed per input sample: direct form = 4*ntaps/decim (+6 for the NCO);
    # the overlap-save kernel: 6 radix-16 passes + 4 twiddle passes + spectrum product per
    # 4096-point block = 1524 FLOP/lane x 256 lanes / (4097 - ntaps) valid outputs.
    flops = w["flops"]
    if kinfo["name"] == "fir_fft_kernel":
        flops = 1524.0 * 256 / (4097 - w["ntaps"])
    elif kinfo["name"] == "chan_uniform_kernel":
        # per lane and wave tile (16 output times x 64 channels = 1024 input samples per wave): 64 complex
        # MACs (512), radix-16 (~200), twiddles (~90), radix-4 across the quad (~220), per-channel
        # rotation (~320): ~1350 FLOP x 64 lanes / 1024 input samples
        flops = 84.0 * (64 // w["decim"])
    achieved_tf = flops * n / (kms * 1e-3) / 1e12

    if rank == 0:
        line = {
            "metric": "Msamples/s complex IQ through 256-tap FIR+decimate chain, 1/2/4/8 GPU",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "spinup_ms": args.spinup_ms,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": {
                    "fir256": "256-tap complex FIR (real taps), synthetic IQ resident in HBM (BASELINE configs[1])",
                    "fir63": "63-tap complex FIR, synthetic IQ (BASELINE configs[0] shape)",
                    "decim8": "256-tap polyphase decimate-by-8, synthetic IQ",
                    "xlate_fir_decim8": "fused NCO + 256-tap FIR + decimate-by-8 (BASELINE configs[2])",
                    "xlate": "NCO frequency translator alone",
                    "decim8_t63": "63-tap polyphase decimate-by-8",
                    "xlate_fir_decim8_t63": "fused NCO + 63-tap FIR + decimate-by-8",
                    "vfo50": "fused NCO + 401-tap FIR + decimate-by-50 (the reference VFO's 2.4 Msps -> 48 kHz shape)",
                    "chan64": "64-channel polyphase channelizer, 256 taps, decimate 64 (BASELINE configs[4])",
                    "chan64m8": "64-channel polyphase channelizer oversampled by 8: 256 taps, decimate 8 (BASELINE configs[4], M = 8 variant)",
                }[args.workload],
                "name": args.workload,
                "samples_per_gpu_per_step": n,
                "ntaps": w["ntaps"],
                "decim": w["decim"],
                "halo_samples": H if world > 1 else 0,
                "partition": ("single stream" + (" (self-ring RCCL exchange every step)" if self_ring else "")) if world == 1 else f"block-cyclic time chunks over {world} ranks, ring halo over RCCL",
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
                "algorithmic_bytes_per_sample": w["bytes"],
                "launch": {"grid": kinfo["grid"], "block": kinfo["block"], "lds_bytes": kinfo["lds_bytes"]},
            },
            "roofline_fp32": {
                "bound": "valu",
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
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                t = json.load(open(tj)).get(args.workload)
                if t and t.get("samples") == n:
                    line["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = t.get("source")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1 or self_ring:
        drain()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
