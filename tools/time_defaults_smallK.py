#!/usr/bin/env python3
"""Diagnostic: the reference's constructor defaults (time_approx='outer', detach_forward=False, log-variance) at notebook batch
sizes: ms per iteration of the native plan."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

dev = torch.device("cuda:0")
for (d, K, T) in ((2, 50, 1.0), (20, 256, 0.5), (100, 1024, 0.5)):
    for inner in (False, True):
        prob = psp.LLGC(d=d, off_diag=0.05, T=T, seed=42, device=dev)
        kw = dict(time_approx="inner") if inner else {}
        m = psp.Solver("t", prob, lr=1e-3, L=40, K=K, delta_t=0.01, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                       backend="native", noise="philox", **kw)
        plan = m._choose_plan()
        losses = torch.zeros(64, device=dev)
        for l in range(4):                                  # (Solver.times averages a whole print block, first launches included)
            plan.iteration(l, losses)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for l in range(4, 24):
            plan.iteration(l, losses)
        torch.cuda.synchronize()
        print("d=%d K=%d N=%d %s (default flags): %s, %.3f ms per iteration" % (
            d, K, m.N, "inner 30-30 MLP" if inner else "outer DenseNets", type(plan).__name__, 1e3 * (time.perf_counter() - t0) / 20))
