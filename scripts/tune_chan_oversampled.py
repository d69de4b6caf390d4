import torch, sys, os
sys.path.insert(0, os.getcwd())
from qdsp_amd import ops
from bench import lowpass_taps
n = 1 << 26
x = ops.synth_iq(n, first_sample=0, seed=1, device=0)
incs = [ops.phase_delta(1.0, -(c - 31.5) / 64.0) for c in range(64)]
for d in (8, 16, 32):
    ch = ops.Channelizer(lowpass_taps(256, 1 / 128), 1, d, incs, max_block=0)
    out = torch.empty((64, n // d + 32), dtype=torch.complex64, device="cuda")   # (row stride not a power of two: bench.CHAN_ROW_PAD)
    ch.process(x, out); torch.cuda.synchronize()
    print("decim", d, "%.3f ms per 2^26 samples" % min(ch.time_dev(x, out, 5) for _ in range(3)))
