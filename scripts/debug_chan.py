"""Debug helper: uniform channelizer kernel vs the per-channel kernels on the same input."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qdsp_amd import ops
from bench import lowpass_taps
n = 1 << 16
x = ops.synth_iq(n, first_sample=0, seed=3, device=0)
incs = [ops.phase_delta(1.0, -(c - 31.5) / 64.0) for c in range(64)]
taps = lowpass_taps(256, 1 / 128)
a = ops.Channelizer(taps, 1, 64, incs, max_block=0); a.set_volk_gain(False)
b = ops.Channelizer(taps, 1, 64, incs, max_block=0); b.set_volk_gain(False); b.set_mode(b.DIRECT)
ya = a.process(x).cpu().numpy(); yb = b.process(x).cpu().numpy()
print(a.last_kernel(), b.last_kernel()["name"])
for c in (0, 1, 2, 5, 16, 17, 33, 63):
    r = ya[c] / yb[c]
    e = np.abs(ya[c] - yb[c]).max() / np.abs(yb[c]).max()
    print(c, "err %.2e" % e, "ratio[8:12]", np.round(r[8:12], 3), "ratio[100:102]", np.round(r[100:102], 3))
