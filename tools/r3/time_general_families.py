"""Small-K GeneralSolver: the (d, H)-templated kernels (gen_*: one wave per 16-trajectory tile, tables in LDS) against the
run-time-shaped ones (genl_*: four waves per tile, tables in L2) on the SAME two-hidden-layer net.  Run on an MI355X."""
import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
import torch
import path_space_pde_solver_amd as psp
from path_space_pde_solver_amd import plan_general_deep as pgd, plan_general_native as pgn
dev = torch.device("cuda:0")
for H in (30, 64):
    for K in (200, 1024, 4096, 16384):
        row = []
        for fam in ("gen", "genl"):
            prob = psp.DoubleWell_multidim_for_general_solver(d=100, d_1=50, d_2=50, T=0.3, eta=1.0, kappa=1.0, modus="HJB", device=dev)
            m = psp.GeneralSolver(prob, "seg", seed=42, delta_t=0.001, N=50, lr=1e-3, L=40, K=K, K_boundary=50, loss_method="diffusion",
                                  verbose=False, device=dev, backend="native", noise="philox", mlp_dtype="auto")
            m.V = psp.DenseNet(d_in=101, d_out=1, lr=1e-3, arch=[H, H], seed=42).to(dev)
            plan = pgd.GeneralDeepPlan(m) if fam == "genl" else pgn.GeneralNativePlan(m)
            for l in range(5):
                plan.iteration(l)
            torch.cuda.synchronize()
            plan.events = []
            t0 = time.perf_counter(); n = 30
            for l in range(5, 5 + n):
                plan.iteration(l)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e3
            ev = plan.events
            fwd = sum(e[0].elapsed_time(e[1]) for e in ev) / n
            bwd = sum(e[2].elapsed_time(e[3]) for e in ev) / n
            row.append("%s %.3f ms (fwd %.3f, bwd %.3f)" % (fam, wall, fwd, bwd))
        print("H=%d K=%5d N=50: %s" % (H, K, "   ".join(row)), flush=True)
