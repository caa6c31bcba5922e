#!/usr/bin/env python3
"""Diagnostic: per-phase cycle breakdown of hjbw_fwd_kernel (needs the -DPSP_STAMPS library)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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

d = int(sys.argv[1]) if len(sys.argv) > 1 else 500
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
off = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1 / d ** 0.5
dev = torch.device("cuda:0")
prob = psp.LLGC(d=d, off_diag=off, T=0.2, seed=42, device=dev)
model = psp.Solver("diag", prob, lr=1e-3, L=3, K=K, delta_t=0.01, loss_method="log-variance",
                   time_approx="inner", adaptive_forward_process=True, detach_forward=True,
                   u_l2_error_flag=False, verbose=False, seed=42, device=dev, backend="native",
                   noise="philox", widths=(64, 64))
plan = model._choose_plan()
nat = psp.native
fwg, nwg = plan.sizes.fwd_workgroups, plan.sizes.bwd_workgroups
buf = torch.zeros((fwg * 8 + nwg * 8) * 8, dtype=torch.int64, device=dev)
assert nat.load().psp_debug_set_stamp_buffer(nat.ptr(buf), buf.numel()) == 1, "library lacks -DPSP_STAMPS"
losses = torch.zeros(4, device=dev)
nostore = len(sys.argv) > 4 and sys.argv[4] == "nostore"
for l in range(2):
    if nostore:
        plan.forward_only(l)
    else:
        plan.iteration(l, losses)
torch.cuda.synchronize()
f = buf.cpu().double()[:fwg * 64].reshape(fwg, 8, 8)[:, :4, :]
steps = f[:, :, 7].clamp(min=1)
names = ["X image + path store", "W1 product", "drift product", "tanh, W2, tanh, h stores",
         "Z groups: W3, Philox, xi store, v", "sigma product, cost, Y", "whole step"]
tot = (f[:, :, 6] / steps).mean()
print("wide forward d=%d K=%d: workgroups %d, ticks per time step %.0f" % (d, K, fwg, tot))
for i, nme in enumerate(names):
    v = f[:, :, i] / steps
    print("  %-36s mean %9.0f  (%5.1f%%)" % (nme, v.mean(), 100 * v.mean() / tot))
