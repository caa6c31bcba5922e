#!/usr/bin/env python3
"""Diagnostic: per-phase tick breakdown of hjbq_fwd_kernel (four trajectories per workgroup) (needs the -DPSP_STAMPS library, see phase_stamps.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("PSP_STAMPS_LIB") or os.path.join(ROOT, "path-space-pde-solver_amd", "csrc", "libpsp_hip_stamps.so")
os.environ["PSP_LIB_PATH"] = LIB

os.environ["PSP_FWD_VARIANT"] = "3"
import torch  # noqa: E402
import path_space_pde_solver_amd as psp  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
prob = psp.LLGC(d=100, off_diag=0.01, T=0.5, seed=42, device=dev)
model = psp.Solver("diag", prob, lr=1e-3, L=3, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                   adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42,
                   device=dev, backend="native", noise="philox", widths=(64, 64))
plan = model._choose_plan()
nat = psp.native
fwg, nwg = plan.sizes.fwd_workgroups, plan.sizes.bwd_workgroups
buf = torch.zeros((fwg * 8 + nwg * 8) * 8, dtype=torch.int64, device=dev)
assert nat.load().psp_debug_set_stamp_buffer(nat.ptr(buf), buf.numel()) == 1
losses = torch.zeros(4, device=dev)
plan.events = []            # eager launches (the stamp buffer is per launch)
for l in range(2):
    plan.iteration(l, losses)
torch.cuda.synchronize()
f = buf.cpu().double()[:fwg * 64].reshape(fwg, 8, 8)
steps = f[:, :, 7].clamp(min=1)
names = ["A X store, W1 + drift products", "B h1, W2 product", "C h2, W3 product, Philox", "D Z, v, sigma product",
         "waits at barriers 1-3", "wait at barrier 4", "whole step"]
tot = (f[:, :, 6] / steps).mean()
print("quad forward K=%d: workgroups %d, ticks per time step %.0f" % (K, fwg, tot))
for i, nme in enumerate(names):
    if nme == "-":  # unused slot
        continue
    v = f[:, :, i] / steps
    print("  %-42s mean %8.0f  (%5.1f%%)  per wave %s" % (nme, v.mean(), 100 * v.mean() / tot, [int(x) for x in v.mean(0).tolist()]))
