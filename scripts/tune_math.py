import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qdsp_amd import ops
n = 1 << 26
a = ops.synth_iq(n, seed=1); b = ops.synth_iq(n, seed=2)
out = torch.empty_like(a)
for op in (0, 2):
    m = ops.Math(op, complex_data=True, max_block=0)
    m.process(a, b, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): m.process(a, b, out)
    e0.record()
    for _ in range(20): m.process(a, b, out)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("math op", op, "%.3f ms per 2^26 complex (24 B/sample): %.0f GB/s" % (ms, n * 24 / ms / 1e6))
