#!/usr/bin/env python3
"""Diagnostic: where the GeneralSolver iteration's time goes outside its two rollout kernels (HIP events of plan.events:
before / after the forward kernel, before / after the backward kernel) -- sampling + boundary terms, loss weights, gradient
assembly + Adam.  Usage: python tools/time_general_segments.py [bf16|fp32]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
prob = psp.DoubleWell_multidim_for_general_solver(d=100, d_1=50, d_2=50, T=0.3, eta=1.0, kappa=1.0, modus="HJB", device=dev)
m = psp.GeneralSolver(prob, "seg", seed=42, delta_t=0.001, N=100, lr=1e-3, L=30, K=65536, K_boundary=50, loss_method="diffusion",
                      verbose=False, device=dev, backend="native", noise="philox", mlp_dtype=mode)
m.V = psp.DenseNet(d_in=101, d_out=1, lr=1e-3, arch=[64, 64], seed=42).to(dev)
plan = m._choose_plan()
for l in range(5):
    plan.iteration(l)
torch.cuda.synchronize()
plan.events = []
t0 = time.perf_counter()
n = 20
for l in range(5, 5 + n):
    plan.iteration(l)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
ev = plan.events
fwd = sum(e[0].elapsed_time(e[1]) for e in ev) / n
mid = sum(e[1].elapsed_time(e[2]) for e in ev) / n
bwd = sum(e[2].elapsed_time(e[3]) for e in ev) / n
tail_head = sum(a[3].elapsed_time(b[0]) for a, b in zip(ev, ev[1:])) / (n - 1)
print("%s: iteration %.3f ms = forward %.3f + loss weights %.3f + backward %.3f + (gradient, Adam, next sample, boundary terms) %.3f"
      % (mode, wall, fwd, mid, bwd, tail_head))
