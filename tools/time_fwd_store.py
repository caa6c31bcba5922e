#!/usr/bin/env python3
"""Diagnostic: forward rollout kernel time with and without the path store (store_path = 1 / 0), HIP events, per workload."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

nat = psp.native
dev = torch.device("cuda:0")
for (d, K, T, off) in ((100, 65536, 1.0, 0.01), (200, 32768, 1.0, 0.1 / 200 ** 0.5), (500, 16384, 2.0, 0.1 / 500 ** 0.5)):
    prob = psp.LLGC(d=d, off_diag=off, T=T, seed=42, device=dev)
    m = psp.Solver("t", prob, lr=1e-3, L=4, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                   adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                   backend="native", noise="philox", widths=(64, 64))
    plan = m._choose_plan()
    lib = nat.load()
    res = {}
    for store in (1, 0):
        cfg = nat.HjbConfig.from_buffer_copy(plan.cfg)
        cfg.store_path = store
        ts = []
        for it in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nat.check(lib.psp_hjb_rollout_fwd(C.byref(cfg), nat.ptr(plan.flat_k), nat.ptr(plan.x0_vec), 0, None, None, 42, it,
                                              nat.ptr(plan.path) if store else None, nat.ptr(plan.D), None, None,
                                              nat.ptr(plan.fwd_partial), nat.stream_ptr(dev)), "fwd")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[store] = sum(ts[2:]) / len(ts[2:])
    print("d=%d K=%d: forward with path store %.3f ms, without %.3f ms (%.1f %%)" % (d, K, res[1], res[0], 100 * (res[1] - res[0]) / res[1]))
