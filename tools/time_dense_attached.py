#!/usr/bin/env python3
"""Diagnostic: DenseNet controls with the reference's default flags (time_approx='outer', detach_forward=False) at the
size of the committed 'outer' bench line: ms per iteration, native plan (rollout + adjoint sweep + backward kernel)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

dev = torch.device("cuda:0")
for detach in (True, False):
    prob = psp.LLGC(d=100, off_diag=0.01, T=0.5, seed=42, device=dev)
    L = 12
    m = psp.Solver("t", prob, lr=1e-3, L=L, K=65536, delta_t=0.01, loss_method="log-variance", time_approx="outer",
                   adaptive_forward_process=True, detach_forward=detach, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                   backend="native", noise="philox")
    m.train()
    assert m.plan_name == "native"
    torch.cuda.synchronize()
    ts = m.times[2:]
    print("outer d=100 H=30 K=65536 N=%d detach_forward=%s: %.2f ms per iteration = %.3e units/s, loss %.3f -> %.3f" % (
        m.N, detach, 1e3 * sum(ts) / len(ts), 65536 * m.N / (sum(ts) / len(ts)), m.loss_log[0], m.loss_log[-1]))
