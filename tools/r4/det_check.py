#!/usr/bin/env python3
"""Diagnostic: run the same native iteration twice and report which parameter segments of the flat gradient differ bitwise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
res = []
for rep in range(3):
    prob = psp.LLGC(d=d, off_diag=0.1 / d ** 0.5, T=0.1, seed=42, device=dev)
    m = psp.Solver("det", prob, lr=1e-3, L=1, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                   adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                   backend="native", noise="philox", widths=(64, 64))
    m.train()
    res.append(m._native_plan.grad.cpu().clone())
H = 64
segs = [("W1", (d + 1) * H), ("b1", H), ("W2", H * H), ("b2", H), ("W3", H * d), ("b3", d)]
for a, b in ((0, 1), (0, 2)):
    o = 0
    out = []
    for name, n in segs:
        ga, gb = res[a][o:o + n], res[b][o:o + n]
        nd = int((ga != gb).sum())
        out.append("%s %d/%d (max rel %.2e)" % (name, nd, n, float(((ga - gb).abs().max() / ga.abs().max()))))
        if name in ("b2", "b1") and nd:
            out.append("%s idx %s" % (name, (ga != gb).nonzero().flatten().tolist()[:20]))
        o += n
    print("run %d vs %d: " % (a, b) + "; ".join(out))
