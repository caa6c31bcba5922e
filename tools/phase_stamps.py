#!/usr/bin/env python3
"""Diagnostic: per-phase cycle breakdown of hjb_bwd_kernel from in-kernel s_memtime stamps.

Needs the diagnostic library (built with -DPSP_STAMPS):
    python tools/phase_stamps.py            # builds csrc/libpsp_hip_stamps.so if missing, then runs
The stamped build is never the shipped one; read SHARES, not absolute times (stamps add fences).
"""
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

dev = torch.device("cuda:0")
prob = psp.LLGC(d=100, off_diag=0.01, T=1.0, seed=42, device=dev)
model = psp.Solver("diag", prob, lr=1e-3, L=4, K=65536, delta_t=0.01, loss_method="log-variance",
                   time_approx="inner", adaptive_forward_process=True, detach_forward=True,
                   u_l2_error_flag=False, verbose=False, seed=42, device=dev, backend="native",
                   noise="philox", widths=(64, 64), path_noise=os.environ.get("PSP_PATH_NOISE", "auto"))
plan = model._choose_plan()
nat = psp.native
nwg = plan.sizes.bwd_workgroups
fwg = plan.sizes.fwd_workgroups
buf = torch.zeros((fwg * 8 + nwg * 8) * 8, dtype=torch.int64, device=dev)
assert nat.load().psp_debug_set_stamp_buffer(nat.ptr(buf), buf.numel()) == 1, "library lacks -DPSP_STAMPS"
losses = torch.zeros(4, device=dev)
for l in range(3):
    plan.iteration(l, losses)
torch.cuda.synchronize()
allb = buf.cpu().double()
f = allb[:fwg * 64].reshape(fwg, 8, 8)
steps = f[:, :, 7].clamp(min=1)
fn = ["X store + L1 GEMM (100 MFMA)", "tanh 1", "L2 + tanh 2 + h store + L3 (176 MFMA)", "Philox + row sums",
      "SDE GEMMs (350 MFMA)", "running cost + Y", "whole step"]
ftot = (f[:, :, 6] / steps).mean()
print("forward: workgroups %d, cycles per time step (mean over waves): %.0f; MFMA floor 626 x 32 = %d (x2 waves/SIMD: %d)"
      % (fwg, ftot, 626 * 32, 626 * 64))
for i, n in enumerate(fn):
    v = f[:, :, i] / steps
    print("  %-40s mean %8.0f  (%5.1f%%)" % (n, v.mean(), 100 * v.mean() / ftot))
if os.environ.get("PSP_BWD_VARIANT", "") != "1":
    # role-specialised backward: 8 waves per workgroup, waves 0-3 producers, 4-7 consumers
    s = allb[fwg * 64:fwg * 64 + nwg * 64].reshape(nwg, 8, 8)
    pr, co = s[:, :4, :], s[:, 4:, :]
    R = pr[:, :, 7].clamp(min=1)
    print("backward (role-specialised): workgroups %d, rounds per workgroup %.1f" % (nwg, R.mean()))
    x3 = model._native_plan.matrix_mode == 'f16x3' if hasattr(model, '_native_plan') and model._native_plan is not None else plan.matrix_mode == 'f16x3'
    prod = ([("h2 loads issued, weights -> G", 0), ("G split + image write", 1), ("W3^T G, tanh', h1 loads, next xi", 2),
             ("whole produce phase", 3), ("barrier wait", 4), ("whole round", 6)] if x3 else
            [("xi -> G, issue h2 / next-xi loads", 0), ("G store, GEMM W3^T G, tanh'", 1), ("dz2 store", 3), ("barrier wait", 4), ("whole round", 6)])
    for nm, i in prod:
        v = pr[:, :, i] / R
        print("  producer  %-40s mean %8.0f   min %8.0f  max %8.0f" % (nm, v.mean(), v.min(), v.max()))
    Rc = co[:, :, 7].clamp(min=1)
    for nm, i in ([("both pairs", 0), ("first pair", 1), ("barrier wait", 4), ("whole round", 6)] if x3 else
                  [("eight phases (352 MFMA)", 0), ("barrier wait", 4), ("whole round", 6)]):
        v = co[:, :, i] / Rc
        print("  consumer  %-40s mean %8.0f   min %8.0f  max %8.0f" % (nm, v.mean(), v.min(), v.max()))
    sys.exit(0)
s = allb[fwg * 64:].reshape(nwg, 4, 8)
rounds = s[:, :, 7]
names = ["P1 compute", "barrier A", "P2 compute", "barriers B+C + dz exchange", "P3 compute", "barrier D", "whole round"]
tot = (s[:, :, 6] / rounds).mean()
print("workgroups %d, rounds per workgroup %.1f, cycles per round (mean over waves): %.0f" % (nwg, rounds.mean(), tot))
for i, n in enumerate(names):
    v = (s[:, :, i] / rounds)
    print("  %-28s mean %8.0f  (%5.1f%%)   min %8.0f  max %8.0f" % (n, v.mean(), 100 * v.mean() / tot, v.min(), v.max()))
print("MFMA floor per round per wave (452 MFMA x 32 cyc): %d; x2 waves/SIMD: %d" % (452 * 32, 452 * 64))
