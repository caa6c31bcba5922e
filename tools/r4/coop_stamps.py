#!/usr/bin/env python3
"""Diagnostic: per-phase cycle breakdown of hjbc_fwd_kernel (default d = 500; usage: coop_stamps.py [d] [K] [N]) from in-kernel s_memtime stamps.
Needs csrc/libpsp_hip_stamps.so (python tools/r4/coop_stamps.py builds the wide instance with -DPSP_STAMPS if missing)."""
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
model = psp.Solver("coop", prob, lr=1e-3, L=4, K=K, delta_t=0.005, loss_method="log-variance", time_approx="inner",
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
f = buf.cpu().double()[:fwg * 64].reshape(fwg, 8, 8)
steps = f[:, :, 7].clamp(min=1)
names = ["X store + P2 (drift product [+ Philox])", "P3 (h2)", "P4 (Z, [Philox,] sums, v image)", "Y + P5 (sigma product) [+ E] + image",
         "barriers", "P1 (h1 [+ Philox]) + W2/W3 prefetch", "whole step"]
tot = (f[:, :, 6] / steps).mean()
print("workgroups %d; cycles per step (mean over waves) %.0f; MFMA floor per SIMD and step: 2 waves x 870 x 16 = %d" % (fwg, tot, 2 * 870 * 16))
for i, n in enumerate(names):
    v = f[:, :, i] / steps
    print("  %-40s mean %8.0f (%5.1f%%)  min %8.0f  max %8.0f" % (n, v.mean(), 100 * v.mean() / tot, v.min(), v.max()))
