#!/usr/bin/env python3
"""Diagnostic: phase cycles of hjbw_bwd_x3_kernel (wide backward, d > 256; four waves, one per SIMD): phase A (own block: dz2 against
the streamed W3^T table), B1 (dz1, dW2), B2 (dW3 / dW1 over the wave's state blocks), barrier wait -- per round of four sample blocks.
usage: python tools/r4/wbx3_stamps.py [d] [K] [N]   (library built with -DPSP_STAMPS on first use)"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "path-space-pde-solver_amd", "csrc")
LIB = os.path.join(CSRC, "libpsp_hip_stamps.so")
if not os.path.exists(LIB):
    spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "path-space-pde-solver_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(extra_flags=["-DPSP_STAMPS"], lib_path=LIB, obj_dir=os.path.join(CSRC, "build_stamps"))
os.environ["PSP_LIB_PATH"] = LIB

import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

dev = torch.device("cuda:0")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 500
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
prob = psp.LLGC(d=d, off_diag=0.01, T=N * 0.005, seed=42, device=dev)
model = psp.Solver("wx", prob, lr=1e-3, L=4, K=K, delta_t=0.005, loss_method="log-variance", time_approx="inner",
                   adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                   backend="native", noise="philox", widths=(64, 64))
plan = model._choose_plan()
nat = psp.native
fwg, nwg = plan.sizes.fwd_workgroups, plan.sizes.bwd_workgroups
buf = torch.zeros((fwg * 8 + nwg * 8) * 8, dtype=torch.int64, device=dev)
assert nat.load().psp_debug_set_stamp_buffer(nat.ptr(buf), buf.numel()) == 1, "library lacks -DPSP_STAMPS"
losses = torch.zeros(4, device=dev)
for l in range(2):
    plan.iteration(l, losses)
torch.cuda.synchronize()
f = buf.cpu().double()[fwg * 64:].reshape(nwg, 8, 8)[:, :4, :]
rounds = f[:, :, 7].clamp(min=1)
tot = (f[:, :, 6] / rounds).mean()
print("d %d  K %d  N %d: %d backward workgroups, %.1f rounds each, %.0f cycles per round" % (d, K, N, nwg, float(rounds.mean()), tot))
for i, n in enumerate(["phase A (dz2 of the own block)", "phase B1 (dz1, dW2)", "phase B2 (dW3, dW1)", "barrier wait"]):
    v = f[:, :, i] / rounds
    print("  %-32s mean %8.0f (%5.1f%%)  min %8.0f  max %8.0f" % (n, v.mean(), 100 * v.mean() / tot, v.min(), v.max()))
