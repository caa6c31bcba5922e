#!/usr/bin/env python3
"""Diagnostic: forward kernel time at small K for N = 25 / 50 / 100 (per-step slope and prologue intercept), per forward variant."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

nat = psp.native
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for variant in ("2", "3"):
    os.environ["PSP_FWD_VARIANT"] = variant
    out = []
    for T in (0.25, 0.5, 1.0):
        prob = psp.LLGC(d=100, off_diag=0.01, T=T, seed=42, device=dev)
        m = psp.Solver("t", prob, lr=1e-3, L=4, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                       adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42,
                       device=dev, backend="native", noise="philox", widths=(64, 64), use_graph=False)
        plan = m._choose_plan()
        lib = nat.load()
        ts = []
        for it in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nat.check(lib.psp_hjb_rollout_fwd(C.byref(plan.cfg), nat.ptr(plan.flat_k), nat.ptr(plan.x0_vec), 0, None, None, 42, it,
                                              nat.ptr(plan.path), nat.ptr(plan.D), None, None, nat.ptr(plan.fwd_partial),
                                              nat.stream_ptr(dev)), "fwd")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append((m.N, min(ts[2:]) * 1e3))
    (n0, t0), (n1, t1), (n2, t2) = out
    slope = (t2 - t1) / (n2 - n1)
    print("variant %s K=%d: %s us  -> %.2f us per step, intercept %.1f us" % (variant, K, ["%d: %.1f" % o for o in out], slope, t1 - slope * n1))
