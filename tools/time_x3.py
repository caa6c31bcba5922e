#!/usr/bin/env python3
"""Diagnostic: headline shape (LLGC d=100, K=65536, N=100, 2x64 MLP) -- ms per iteration of the fp32-MFMA and the split-product
(f16x3) forward, and the first losses of both."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(100, 64, 65536, 1.0)] if len(sys.argv) < 2 else [tuple(float(x) if i == 3 else int(x) for i, x in enumerate(a.split(","))) for a in sys.argv[1:]]
for (d, H, K, T) in shapes:
    for mode in ("fp32", "f16x3"):
        prob = psp.LLGC(d=d, off_diag=0.01, T=T, seed=42, device=dev)
        m = psp.Solver("t", prob, lr=1e-3, L=40, K=K, delta_t=0.01, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                       backend="native", noise="philox", time_approx="inner", adaptive_forward_process=True, detach_forward=True,
                       widths=(H, H), mlp_dtype=mode)
        plan = m._choose_plan()
        losses = torch.zeros(64, device=dev)
        for l in range(4):
            plan.iteration(l, losses)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for l in range(4, 24):
            plan.iteration(l, losses)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        print("d=%d H=%d K=%d N=%d %s: %.3f ms per iteration = %.3g units/s; losses %s" % (
            d, H, K, m.N, mode, ms, K * m.N / ms * 1e3, [round(float(x), 6) for x in losses[:4].tolist()]), flush=True)
