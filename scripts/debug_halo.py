import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from qdsp_amd import ops
n = 1 << 22
for wl in ("fir256", "decim8"):
    x0 = ops.synth_iq(n, first_sample=0)
    x1 = ops.synth_iq(n, first_sample=n)
    op = bench.make_op(ops, wl, 0)
    H = op.history_len
    tail = x0[n - H:]
    halo = torch.zeros(H, dtype=torch.complex64, device="cuda")
    halo.copy_(tail.cpu())
    op.set_history_dev(halo)
    y = op.process(x1).clone()
    torch.cuda.synchronize()
    # reference 1: fresh op, host set_history
    chk = bench.make_op(ops, wl, 0)
    chk.set_history(tail.cpu().numpy())
    r1 = chk.process(x1[: 1 << 16]).clone()
    # reference 2: one op processing x0 then x1 (natural carry)
    nat = bench.make_op(ops, wl, 0)
    nat.process(x0)
    r2 = nat.process(x1).clone()
    torch.cuda.synchronize()
    pt = ops.synth_iq(H, first_sample=n - H)
    print(wl, "H", H, "regen tail equal:", torch.equal(pt, tail), "| y==natural:", torch.equal(y, r2), "| y[:m]==fresh(host hist):", torch.equal(y[: r1.numel()], r1),
          "| maxdiff vs fresh", (y[: r1.numel()] - r1).abs().max().item(), "| kernel", op.last_kernel()["name"], chk.last_kernel()["name"])
