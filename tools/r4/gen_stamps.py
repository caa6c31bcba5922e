#!/usr/bin/env python3
"""Diagnostic: who waits for whom in gen_bwd2_kernel<.., X3> (diffusion-loss backward, value net [H, H]): cycles of work and of
barrier wait per round, producers (waves 0-3) and consumers (waves 4-7), from in-kernel s_memtime stamps (-DPSP_STAMPS build).
usage: python tools/r4/gen_stamps.py [K] [N]"""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = 100
prob = psp.DoubleWell_multidim_for_general_solver(d=d, d_1=d // 2, d_2=d - d // 2, T=0.3, eta=1, kappa=1, modus="HJB", device=dev)
model = psp.GeneralSolver(problem=prob, name="stamps", seed=42, delta_t=0.001, N=N, lr=1e-3, L=3, K=K, K_boundary=50,
                          alpha=[1.0, 1.0, 1.0], loss_method="diffusion", verbose=False, device=dev, backend="native",
                          noise="philox", mlp_dtype="auto")
model.V = psp.DenseNet(d_in=d + 1, d_out=1, lr=1e-3, arch=[64, 64], seed=42).to(dev)
nat = psp.native
nwg = 256
buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
assert nat.load().psp_debug_set_stamp_buffer(nat.ptr(buf), buf.numel()) == 1, "library lacks -DPSP_STAMPS"
model.train()
torch.cuda.synchronize()
assert model.plan_name == "native", model.plan_name
f = buf.cpu().double().reshape(nwg, 8, 8)
rounds = f[:, :, 7].clamp(min=1)
print("diffusion loss d %d  K %d  N %d: %.1f rounds per workgroup (last iteration)" % (d, K, N, float(rounds.mean())))
for name, sl in (("producers (waves 0-3)", slice(0, 4)), ("consumers (waves 4-7)", slice(4, 8))):
    w = (f[:, sl, 0] / rounds[:, sl])
    b = (f[:, sl, 1] / rounds[:, sl])
    print("  %-24s work %8.0f cycles/round (min %6.0f max %6.0f)   barrier wait %8.0f (min %6.0f max %6.0f)"
          % (name, w.mean(), w.min(), w.max(), b.mean(), b.min(), b.max()))
